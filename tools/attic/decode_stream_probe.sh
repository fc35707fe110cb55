# decode-driver rate vs which of torch's pool streams the pass streams are (developer aid)
mkdir -p gpurun_out/r5n
for k in 0 1 2 4 5 8; do
  echo "FCL_STREAM_SKIP=$k" >> gpurun_out/r5n/decode_skip.log
  FCL_STREAM_SKIP=$k python3 tools/bench_decode.py 4096 2>&1 | grep "depth 4" | cut -c1-150 >> gpurun_out/r5n/decode_skip.log
done
FCL_TE_STAMPS=1 python3 tools/kd_phases.py kd > gpurun_out/r5n/kd_phases.log 2>&1
FCL_TE_STAMPS=1 python3 tools/kd_phases.py teacher > gpurun_out/r5n/teacher_phases.log 2>&1
