# follow-up of tools/tunable_scan_synth.sh: the three candidates and their combinations, three alternating repetitions, same box (M frames/s)
OUT=gpurun_out/${1:-r6SS2}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % (d['value']/1e6))"; }
for rep in 1 2 3; do
  for cfg in "X=0" "PLSTM_2STAGE_MIN_WG=200" "PLSTM_2STAGE_MIN_WG=150" "PLSTM_2STAGE_MIN_WG=100" "PLSTM_2STAGE_MIN_WG=50" "PLSTM_2STAGE_MIN_WG=150 FCL_PLSTM_BIG_MIN_S=150" \
             "PLSTM_2STAGE_MIN_WG=150 FCL_GEMM_TM2=0" "PLSTM_2STAGE_MIN_WG=150 FCL_PLSTM_BIG_MIN_S=150 FCL_GEMM_TM2=0" "PLSTM_2STAGE_MIN_WG=150 FCL_PGEMM_2STAGE_MIN_WG=150"; do
    v=$(env FCL_$cfg python3 bench.py --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep FCL_$cfg  $v" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
