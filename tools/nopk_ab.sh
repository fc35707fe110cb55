# round 6: the library WITHOUT packed-FP32 instructions (in-tree build) against the same sources WITH them (tools/ab/libfcl_pk.so), same box, alternating
OUT=gpurun_out/${1:-r6N}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])"; }
for rep in 1 2 3; do
  for lib in pk nopk; do
    if [ $lib = pk ]; then export FCL_LIB=$PWD/tools/ab/libfcl_pk.so; else unset FCL_LIB; fi
    a=$(python3 bench.py --no-cpu-baseline --no-extras --regions 7 2>>$OUT/err.log | val)
    b=$(python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    c=$(python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
    echo "rep $rep $lib: headline $a  kd_step_ms $b  teacher_step_ms $c" >> $OUT/nopk_ab.log
  done
done
unset FCL_LIB
for lib in pk nopk; do
  if [ $lib = pk ]; then export FCL_LIB=$PWD/tools/ab/libfcl_pk.so; else unset FCL_LIB; fi
  echo "== bilstm_bench $lib (student, teacher)" >> $OUT/nopk_ab.log
  python3 tools/bilstm_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -8 >> $OUT/nopk_ab.log
  BILSTM_BENCH_MODEL=teacher python3 tools/bilstm_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -8 >> $OUT/nopk_ab.log
done
unset FCL_LIB
cat $OUT/nopk_ab.log
