mkdir -p gpurun_out/r5g
python3 -m pytest tests/test_gpu_train_native.py -x -q > gpurun_out/r5g/test_native.log 2>&1
for w in kd_step teacher_step; do
  python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule > gpurun_out/r5g/bench_$w.json 2> gpurun_out/r5g/bench_$w.err
  FCL_PLSTM_PAIR_2STAGE_MIN_WG=300 python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule > gpurun_out/r5g/bench_${w}_2st.json 2> /dev/null
  FCL_PLSTM_PAIR_2STAGE_MIN_WG=150 python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule > gpurun_out/r5g/bench_${w}_2st150.json 2> /dev/null
done
