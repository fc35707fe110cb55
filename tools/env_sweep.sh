#!/bin/bash
# Headline line under HIP-runtime switches (queues / graph dispatch); one JSON value per setting.  Usage: bash tools/env_sweep.sh <outdir>
out=${1:-gpurun_out/env_sweep}; mkdir -p $out
run() {  # label, streams, env...
  label=$1; streams=$2; shift 2
  v=$(env "$@" timeout 300 python3 bench.py --steps 20 --warmup 5 --regions 5 --no-extras --no-cpu-baseline --streams $streams 2>$out/$label.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f M  %.4f ms' % (d['value']/1e6, d['ms_per_step']))")
  echo "$label streams=$streams $* -> $v" | tee -a $out/sweep.log
}
run base4 4 X=0
run base6 6 X=0
for s in 4 6; do
  run dynq0_$s $s DEBUG_HIP_DYNAMIC_QUEUES=0
  run dynq1_$s $s DEBUG_HIP_DYNAMIC_QUEUES=1
  run fgq1_$s $s DEBUG_HIP_FORCE_GRAPH_QUEUES=1
  run fgq4_$s $s DEBUG_HIP_FORCE_GRAPH_QUEUES=4
  run gbs1_$s $s DEBUG_HIP_GRAPH_BATCH_SIZE=1
  run gbs256_$s $s DEBUG_HIP_GRAPH_BATCH_SIZE=256
  run pcap0_$s $s DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run pcap1_$s $s DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  run rings8_$s $s GPU_NUM_COMPUTE_RINGS=8
  run dd0_$s $s AMD_DIRECT_DISPATCH=0
  run faq1_$s $s DEBUG_HIP_FORCE_ASYNC_QUEUE=1
  run cpw1_$s $s GPU_STREAMOPS_CP_WAIT=1
  run cpw0_$s $s GPU_STREAMOPS_CP_WAIT=0
done
run base4b 4 X=0
