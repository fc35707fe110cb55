# same-box A/B of two builds of the library on the training updates: tools/ab/libfcl_base.so (FCL_LIB) against the in-tree build, alternating
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d.get('launches_per_update'))"; }
for rep in 1 2 3; do
 for w in kd_step teacher_step; do
  a=$(FCL_LIB=$PWD/tools/ab/libfcl_pk.so python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  b=$(python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  echo "$w rep $rep base $a new $b" >> $OUT/ab.log
 done
done
cat $OUT/ab.log
