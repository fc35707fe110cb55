mkdir -p gpurun_out/r5b
python3 -m pytest tests/test_gpu_backward.py tests/test_gpu_planes.py -x -q > gpurun_out/r5b/test_bwd.log 2>&1
python3 tools/bench_dw.py > gpurun_out/r5b/bench_dw.log 2>&1
FCL_DW_WORKGROUPS=256 python3 tools/bench_dw.py > gpurun_out/r5b/bench_dw_256.log 2>&1
FCL_DW_WORKGROUPS=1024 python3 tools/bench_dw.py > gpurun_out/r5b/bench_dw_1024.log 2>&1
python3 -m pytest tests/test_gpu_training_fullsize.py tests/test_gpu_train_native.py -x -q > gpurun_out/r5b/test_full.log 2>&1
python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule > gpurun_out/r5b/bench_kd.json 2> gpurun_out/r5b/bench_kd.err
python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule > gpurun_out/r5b/bench_teacher.json 2> gpurun_out/r5b/bench_teacher.err
