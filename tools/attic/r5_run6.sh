mkdir -p gpurun_out/r5j
python3 -m pytest tests/test_gpu_training.py tests/test_gpu_bench_config.py tests/test_gpu_rowmaps.py -x -q > gpurun_out/r5j/test.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5j/bench_default.json 2> gpurun_out/r5j/bench_default.err
