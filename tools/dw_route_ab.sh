# weight gradients: from how many outputs on the transposed-planes route (pack_planes_t + the LDS-DMA GEMM) is taken instead of dw_mfma_kernel (FCL_DW_PLANES_MIN;
# 0 = the per-role default: 256 k in the teacher's own update, 1 M in the student's), same box
OUT=gpurun_out/${1:-dwroute}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])"; }
for rep in 1 2 3; do
 for w in teacher_step kd_step; do
  line="$w rep $rep"
  for v in 0 1048576 4194304 1073741824; do
    x=$(FCL_DW_PLANES_MIN=$v python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    line="$line  min=$v: $x"
  done
  echo "$line" >> $OUT/ab.log
 done
done
cat $OUT/ab.log
