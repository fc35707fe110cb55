mkdir -p gpurun_out/r5k
FCL_PRECISION=0 python3 -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r5k/test_exact.log 2>&1
FCL_PRECISION=0 python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/r5k/bench_fp32_exact.json 2> gpurun_out/r5k/bench_fp32_exact.err
FCL_PRECISION=0 FCL_PCONV=0 python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/r5k/bench_fp32_exact_nopconv.json 2> /dev/null
