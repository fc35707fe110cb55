mkdir -p gpurun_out/r5u
for cfg in "A FCL_BILSTM_KSPLIT=1 FCL_BILSTM_GROUP_LL=1" "B FCL_BILSTM_KSPLIT=0 FCL_BILSTM_GROUP_LL=1" "C FCL_BILSTM_KSPLIT=1 FCL_BILSTM_GROUP_LL=0" "D FCL_BILSTM_KSPLIT=0 FCL_BILSTM_GROUP_LL=0"; do
  set -- $cfg
  for rep in 1 2; do
    env $2 $3 timeout 600 python3 -m pytest tests/test_gpu_training.py -x -q -k "one_rank_nccl" > gpurun_out/r5u/nccl_$1_$rep.log 2>&1
    echo "$cfg rep $rep: $(tail -1 gpurun_out/r5u/nccl_$1_$rep.log)" >> gpurun_out/r5u/summary.log
  done
done
python3 tools/time_smallm.py > gpurun_out/r5u/smallm.log 2>&1
