# synthesis tunables on the FCL-taco2-T line (bench.py --model teacher), final build, same box (M frames/s)
OUT=gpurun_out/${1:-r6ST}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % (d['value']/1e6))"; }
for rep in 1 2; do
  for cfg in "X=0" "PLSTM_CFG=1" "PLSTM_CFG=6" "PLSTM_2STAGE_MIN_WG=1000" "PLANES_LOADERS=2" "TILE_GROUP=4" "TILE_GROUP=16" "LSTM_SMALL_M=32" "LSTM_SMALL_M=128" "LSTM_SMALL_M=256" "PLSTM_TILE96=1" \
             "PGEMM_2STAGE_MIN_WG=150" "PGEMM_2STAGE_MIN_WG=1000" "PGEMM_BIG_MIN=60" "GEMM_TM2=0" "PLSTM_BIG_MIN=80" "PLSTM_MID_MIN=100" "PLSTM_ROW32_M=600"; do
    v=$(env FCL_$cfg python3 bench.py --model teacher --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep FCL_$cfg  $v" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
