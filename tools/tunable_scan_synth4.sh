# second pass of the synthesis re-scan, on top of FCL_PLSTM_2STAGE_MIN_WG=125 (the new default): the other two-stage / tile options, three alternating repetitions (M frames/s)
OUT=gpurun_out/${1:-r6SS4}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f cal %.3f' % (d['value']/1e6, d.get('value_calibrated_caps',0)/1e6))"; }
for rep in 1 2 3; do
  for cfg in "X=0" "PGEMM_2STAGE_MIN_WG=150" "PGEMM_2STAGE_MIN_WG=80" "PCONV_2STAGE=1" "GEMM_TM2=0" "PLSTM_BIG_MIN_S=150" "PLSTM_MID_MIN_S=160" "PCONV_BIG_MIN=60" "PGEMM_BIG_MIN=60" "PCONV_2STAGE=1 FCL_PGEMM_2STAGE_MIN_WG=150 FCL_GEMM_TM2=0"; do
    v=$(env FCL_$cfg python3 bench.py --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep FCL_$cfg  $v" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
