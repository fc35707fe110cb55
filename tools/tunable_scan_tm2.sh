# FCL_GEMM_TM2 (64 x 128 tiles of the fp32-operand GEMM) 1 (default) against 0, and FCL_TILE_GROUP 8 (default) against 4, on four lines, same box
OUT=gpurun_out/${1:-r6TM2}; mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % (d['value'] if d['value'] < 1e3 else d['value']/1e6))"; }
for rep in 1 2 3; do
  for cfg in "X=0" "GEMM_TM2=0" "TILE_GROUP=4"; do
    a=$(env FCL_$cfg python3 bench.py --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    b=$(env FCL_$cfg python3 bench.py --model teacher --no-cpu-baseline --no-extras --regions 5 2>>$OUT/err.log | val)
    k=$(env FCL_$cfg python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    t=$(env FCL_$cfg python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep FCL_$cfg  S $a M  T $b M  kd_step $k ms  teacher_step $t ms" >> $OUT/scan.log
  done
done
cat $OUT/scan.log
