# KD update with the frozen teacher's stream restricted to n compute units (fcl_stream_create_cus), and the weight-gradient stream likewise:
# gpurun_out/$1/cus_sweep.log  (round 6).  A CU-masked queue cannot be placed (ops.stream_apart): it measures as contending with every other queue whatever its
# pipe -- 12 of 12 candidates rejected -- which is the mask's cost, not a placement accident.
OUT=gpurun_out/${1:-r6c}
mkdir -p $OUT
B="python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5"
val() { python3 -c "import json,sys; print(json.loads(sys.stdin.read().strip().split('\n')[-1])['value'])"; }
for t in 0 64 96 128 160 192 224; do
  v=$(FCL_KD_TEACHER_CUS=$t $B 2>>$OUT/err.log | val)
  echo "teacher_cus=$t side_cus=0 kd_step_ms=$v" >> $OUT/cus_sweep.log
done
for s in 0 64 128 192; do
  v=$(FCL_TE_SIDE_CUS=$s python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  echo "teacher_step side_cus=$s ms=$v" >> $OUT/cus_sweep.log
done
cat $OUT/cus_sweep.log
