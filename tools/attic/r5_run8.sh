mkdir -p gpurun_out/r5l
python3 -m pytest tests -m gpu -x -q > gpurun_out/r5l/test_all.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5l/bench_default.json 2> gpurun_out/r5l/bench_default.err
