# pair-kernel tile thresholds for the KD update (student pairs: 160 tiles of 128 rows / 320 of 64 rows per problem; frozen teacher: 640 of 128 rows), same box
OUT=gpurun_out/${1:-r6TS4}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % d['value'])"; }
for rep in 1 2; do
  for cfg in "X=0" "PLSTM_PAIR_BIG_MIN=200" "PLSTM_PAIR_BIG_MIN=100" "PLSTM_PAIR_MID_MIN=400" "PLSTM_PAIR_MID_MIN=100" "PLSTM_PAIR_ROW32_M=600" "PLSTM_PAIR_ROW32_M=2000" "PLSTM_PAIR_2STAGE_MAX_WG=1024" "PLSTM_PAIR_2STAGE_MIN_WG=128"; do
    k=$(env FCL_$cfg python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    t=$(env FCL_$cfg python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep FCL_$cfg  kd_step $k  teacher_step $t" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
