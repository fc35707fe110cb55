# round 6: the 160-row T-size LSTM tile (quantisation pick) on / off, same box: isolated steps, T-model synthesis, teacher update, KD update
OUT=gpurun_out/${1:-r6I}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])"; }
for on in 0 1 0 1; do
  a=$(FCL_PLSTM_TILE160=$on python3 bench.py --model teacher --no-cpu-baseline --no-extras 2>>$OUT/err.log | val)
  b=$(FCL_PLSTM_TILE160=$on python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  c=$(FCL_PLSTM_TILE160=$on python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  echo "tile160=$on  teacher_synthesis $a  teacher_step_ms $b  kd_step_ms $c" >> $OUT/tile160_ab.log
done
for on in 0 1; do echo "== isolated T-size steps, FCL_PLSTM_TILE160=$on" >> $OUT/tile160_ab.log; FCL_PLSTM_TILE160=$on LSTM_BENCH_MODEL=teacher python3 tools/lstm_bench.py 2>&1 | grep -v -i "warn\|amdgpu" | tail -12 >> $OUT/tile160_ab.log; done
cat $OUT/tile160_ab.log
