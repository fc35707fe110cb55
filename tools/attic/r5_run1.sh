mkdir -p gpurun_out/r5a
./tools/probe/tr_probe > gpurun_out/r5a/tr_probe.log 2>&1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r5a/bench_default.json 2> gpurun_out/r5a/bench_default.err
python3 bench.py --workload kd_step --no-cpu-baseline > gpurun_out/r5a/bench_kd.json 2> gpurun_out/r5a/bench_kd.err
python3 bench.py --workload teacher_step --no-cpu-baseline > gpurun_out/r5a/bench_teacher.json 2> gpurun_out/r5a/bench_teacher.err
python3 tools/bench_dw.py > gpurun_out/r5a/bench_dw.log 2>&1
