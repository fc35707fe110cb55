# Re-scan of synthesis-side tunables on the headline line (four passes in flight, fresh feed), final build, same box: value in M frames/s
OUT=gpurun_out/${1:-r6SS}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % (d['value']/1e6))"; }
for rep in 1 2; do
  for cfg in "X=0" "LSTM_SMALL_M=256" "LSTM_SMALL_M=384" "LSTM_SMALL_M=768" "LSTM_SMALL_M=1024" "PLSTM_MID_MIN_S=40" "PLSTM_MID_MIN_S=160" "PLSTM_BIG_MIN_S=150" "PLSTM_BIG_MIN_S=600" \
             "PLSTM_ROW32_M=600" "PLSTM_ROW32_M=2000" "PLSTM_2STAGE_MIN_WG=150" "PLSTM_2STAGE_MIN_WG=100000" "PLANES_LOADERS=2" "TILE_GROUP=4" "TILE_GROUP=16" "DEC_TILE_MIN_ROWS=512" \
             "DEC_TILE_MIN_ROWS=2048" "FP_SPLIT_RT=1" "FP_SPLIT_RT=2" "LSTM_SMALL_PAIR=0" "GEMM_TM2=0"; do
    v=$(env FCL_$cfg python3 bench.py --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep FCL_$cfg  $v" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
