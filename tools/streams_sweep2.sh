# where does the 4 -> 5 passes-in-flight cliff come from: feed (fresh / replay) x launch form (graphs / eager)
OUT=gpurun_out/${1:-streams2}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['ms_per_step'])"; }
for s in 4 5 6 8; do
 for feed in fresh replay; do
  for form in "" "--eager"; do
    v=$(python3 bench.py --streams $s --batches $s --feed $feed $form --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    echo "streams $s feed $feed form '$form'  $v" >> $OUT/sweep.log
  done
 done
done
cat $OUT/sweep.log
