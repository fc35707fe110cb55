mkdir -p gpurun_out/r5f
python3 -m pytest tests/test_gpu_train_native.py tests/test_gpu_training_fullsize.py tests/test_gpu_training.py tests/test_gpu_backward.py -x -q > gpurun_out/r5f/test_train.log 2>&1
for w in kd_step teacher_step; do
  python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule > gpurun_out/r5f/bench_$w.json 2> gpurun_out/r5f/bench_$w.err
  FCL_TRAIN_WAVEFRONT=0 python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule > gpurun_out/r5f/bench_${w}_nowf.json 2> /dev/null
done
