mkdir -p gpurun_out/r5o
for cfg in "" "FCL_PLSTM_PAIR_NST2=1" "FCL_PLSTM_PAIR_NST2=1 FCL_PLSTM_PAIR_BIG_MIN=300" "FCL_PLSTM_PAIR_BIG_MIN=300" "FCL_PLSTM_PAIR_BIG_MIN=600 FCL_PLSTM_PAIR_MID_MIN=100"; do
  for w in kd_step teacher_step; do
    v=$(env $cfg python3 bench.py --workload $w --no-cpu-baseline --no-dp-schedule 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.readline())['value'],3))")
    echo "$w [$cfg] $v" >> gpurun_out/r5o/pair_ab.log
  done
done
