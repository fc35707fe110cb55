# round 6: FCL_PCONV=2 (the Conv1d stencil kernel also at Cin >= 512: FCL-taco2-T's postnet / encoder convolutions) against the default (K-term GEMM there), same box
OUT=gpurun_out/${1:-r6X}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])"; }
for rep in 1 2 3; do for p in 1 2; do
  a=$(FCL_PCONV=$p python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  b=$(FCL_PCONV=$p python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  c=$(FCL_PCONV=$p python3 bench.py --model teacher --no-cpu-baseline --no-extras 2>>$OUT/err.log | val)
  echo "rep $rep FCL_PCONV=$p  kd_step_ms $a  teacher_step_ms $b  teacher_synthesis $c" >> $OUT/pconv2_ab.log
done; done
cat $OUT/pconv2_ab.log
