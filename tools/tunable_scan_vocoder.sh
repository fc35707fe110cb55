# vocoder-side tunables on the generator-alone line (bench.py --workload vocoder: wall s per audio s, lower is better), final build, same box
OUT=gpurun_out/${1:-r6VO}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.7f' % d['value'])"; }
for rep in 1 2; do
  for cfg in "X=0" "PWG_LOADERS=2" "PWG_PERSIST=0" "PWG_PERSIST_WGS=2" "PWG_CFG=1" "PWG_CFG=3" "PWG_LAST_FUSED=0" "TILE_GROUP=4"; do
    v=$(env FCL_$cfg python3 bench.py --workload vocoder --steps 5 --warmup 2 --no-cpu-baseline 2>>$OUT/err.log | val)
    echo "rep $rep FCL_$cfg  $v" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
