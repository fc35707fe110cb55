# FCL_PCONV_BIG_MIN 150 (old default) against 60 on the synthesis lines, same box (M frames/s)
OUT=gpurun_out/${1:-r6SS5}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % (d['value']/1e6))"; }
for rep in 1 2 3; do
  for v in 150 60; do
    a=$(FCL_PCONV_BIG_MIN=$v python3 bench.py --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    b=$(FCL_PCONV_BIG_MIN=$v python3 bench.py --streams 1 --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    c=$(FCL_PCONV_BIG_MIN=$v python3 bench.py --batch 64 --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    d=$(FCL_PCONV_BIG_MIN=$v python3 bench.py --model teacher --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    e=$(FCL_PRECISION=0 FCL_PCONV_BIG_MIN=$v python3 bench.py --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep PCONV_BIG_MIN=$v | 4 streams: $a | 1 stream: $b | B=64: $c | T: $d | exact fp32: $e" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
