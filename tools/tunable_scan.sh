# Re-scan of training-side tunables whose defaults were set in earlier rounds, on the final build (the update's pace is set by the bytes it moves now: the
# weight-gradient route threshold, last measured in round 3, was off by 2.6 %).  One line per (tunable, value): teacher update / KD update ms, same box.
OUT=gpurun_out/${1:-tscan}
mkdir -p $OUT
val() { python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%.3f' % d['value'])"; }
run() {  # name value
  t=$(env FCL_$1=$2 python3 bench.py --workload teacher_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  k=$(env FCL_$1=$2 python3 bench.py --workload kd_step --no-cpu-baseline --no-dp-schedule --regions 5 2>>$OUT/err.log | val)
  echo "FCL_$1=$2  teacher_step $t  kd_step $k" >> $OUT/scan.log
}
run NOTHING 0
run BPTT_PLANES_MIN_M 128
run BPTT_PLANES_MIN_M 512
run DW_BIG_TILES_MIN_ROWS 4096
run DW_BIG_TILES_MIN_ROWS 16384
run DW_MIN_CHUNKS 4
run DW_MIN_CHUNKS 16
run DW_WORKGROUPS_SMALL 512
run DW_WORKGROUPS_SMALL 1024
run TN_WORKGROUPS 512
run TN_WORKGROUPS 2048
run GEMM_SMALLM 128
run GEMM_SMALLM 512
run PLSTM_PAIR_NST2 1
run TE_DX_PLANES 1
run NOTHING 1
cat $OUT/scan.log
