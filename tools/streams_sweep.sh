# headline synthesis line against the number of passes in flight, with GPU_MAX_HW_QUEUES at the package default (8) and at 16
OUT=gpurun_out/${1:-streams}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
 for q in 8 16; do
  for s in 3 4 5 6 8; do
    v=$(GPU_MAX_HW_QUEUES=$q python3 bench.py --streams $s --batches $s --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep queues $q streams $s  $v" >> $OUT/sweep.log
  done
 done
done
cat $OUT/sweep.log
