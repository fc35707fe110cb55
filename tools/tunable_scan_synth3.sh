# FCL_PLSTM_2STAGE_MIN_WG 300 (old default) against 100 and 125 on every synthesis line that the S-size step kernels serve, same box (M frames/s)
OUT=gpurun_out/${1:-r6SS3}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % (d['value']/1e6), 'cal %.3f' % (d.get('value_calibrated_caps',0)/1e6), 'dd %.3f' % (d.get('value_decode_driver',0)/1e6))"; }
for rep in 1 2 3; do
  for v in 300 100 125; do
    a=$(FCL_PLSTM_2STAGE_MIN_WG=$v python3 bench.py --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    b=$(FCL_PLSTM_2STAGE_MIN_WG=$v python3 bench.py --streams 1 --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    c=$(FCL_PLSTM_2STAGE_MIN_WG=$v python3 bench.py --batch 64 --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    d=$(FCL_PLSTM_2STAGE_MIN_WG=$v python3 bench.py --feed replay --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep MIN_WG=$v | 4 streams: $a | 1 stream: $b | B=64: $c | replay: $d" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
