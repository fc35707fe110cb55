# the decode driver (depth 4, nothing written: synthesis + D2H of every mel) with and without the measured placement of its pass streams, same box
OUT=gpurun_out/${1:-placedec}
mkdir -p $OUT
for rep in 1 2 3; do
 for place in 0 1; do
  v=$(FCL_PLACE_STREAMS=$place BENCH_DECODE_DEPTHS=4 BENCH_DECODE_REPS=3 python3 tools/bench_decode.py 4096 2>>$OUT/err.log | grep "nothing written" | sed 's/.*= //' | tr '\n' ' ')
  echo "rep $rep FCL_PLACE_STREAMS=$place  $v" >> $OUT/ab.log
 done
done
cat $OUT/ab.log
