# the two defaults the round-6 re-scan changed, against their old values, three alternating repetitions, same box
OUT=gpurun_out/${1:-r6TS3}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % d['value'])"; }
for rep in 1 2 3; do
 for w in teacher_step kd_step; do
  line="$w rep $rep"
  for cfg in "X=0" "BPTT_PLANES_MIN_M=256" "PLSTM_PAIR_2STAGE_MIN_WG=1073741824" "BPTT_PLANES_MIN_M=256 FCL_PLSTM_PAIR_2STAGE_MIN_WG=1073741824"; do
    x=$(env FCL_$cfg python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    line="$line | $cfg: $x"
  done
  echo "$line" >> $OUT/ab.log
 done
done
cat $OUT/ab.log
