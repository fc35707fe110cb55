# HIP / HSA runtime switches not covered by tools/env_sweep.sh, on the three main lines, same box
OUT=gpurun_out/${1:-r6ENV}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % (d['value'] if d['value'] < 1e3 else d['value']/1e6))"; }
for rep in 1 2; do
  for cfg in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "HSA_ENABLE_SDMA=0" "GPU_MAX_HW_QUEUES=16" "HIP_LAUNCH_BLOCKING=0" "HSA_OVERRIDE_CPU_AFFINITY_DEBUG=0"; do
    a=$(env $cfg python3 bench.py --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    k=$(env $cfg python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    t=$(env $cfg python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep $cfg  synthesis $a M  kd_step $k ms  teacher_step $t ms" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
