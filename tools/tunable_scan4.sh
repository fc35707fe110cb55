# GEMM / Conv1d tile thresholds on the two training updates (the synthesis line was swept in tools/pconv_tile_sweep.sh), same box
OUT=gpurun_out/${1:-r6TS5}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % d['value'])"; }
for rep in 1 2; do
  for cfg in "X=0" "PCONV=2" "PCONV_2STAGE=1" "PCONV_BIG_MIN=60" "PCONV_BIG_MIN=400" "PGEMM_BIG_MIN=60" "PGEMM_BIG_MIN=400" "PGEMM_2STAGE_MIN_WG=150" "PGEMM_2STAGE_MIN_WG=1000" "PCONV_LOADERS=4" "TE_R6=1" "TE_R6=2"; do
    k=$(env FCL_$cfg python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    t=$(env FCL_$cfg python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep FCL_$cfg  kd_step $k  teacher_step $t" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
