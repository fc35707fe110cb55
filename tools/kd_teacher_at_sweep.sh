# where the next batch's frozen-teacher forward is enqueued (FCL_KD_TEACHER_AT: -1 = at the update's start, 0 .. 3 = behind that backward stage), same box
OUT=gpurun_out/${1:-kdat}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])"; }
for rep in 1 2 3; do
  for at in -1 0 1 2 3; do
    v=$(FCL_KD_TEACHER_AT=$at python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep teacher_at $at  kd_step_ms $v" >> $OUT/sweep.log
  done
done
cat $OUT/sweep.log
