# Collects the round's profile evidence on the GPU box into gpurun_out/prof_$1/ (copy the summaries to profiles/ afterwards):
#   default bench JSON (fresh feed, 4 capacity graphs in flight), rocprofv3 kernel stats of that command and of the eager single-stream pass,
#   PMC FETCH_SIZE / WRITE_SIZE / SQ counters (separate --pmc passes, --kernel-trace only), training-step JSONs + kernel stats + PMC traffic.
TAG=${1:-r6}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
stats() { find $1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $2; }
pmc() { f=$(find $1 -name "*counter_collection.csv" | head -1); python3 tools/$3 $f > $2 2>/dev/null; rm -rf $1; }
rocprofv3 --kernel-trace --stats -d $OUT/fresh4 -o g --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 40 --regions 3 > $OUT/fresh4_bench.json 2>/dev/null
stats $OUT/fresh4 $OUT/fresh_4streams_kernel_stats.csv; rm -rf $OUT/fresh4
rocprofv3 --kernel-trace --stats -d $OUT/eager1 -o e --output-format csv -- python3 bench.py --streams 1 --eager --feed replay --no-cpu-baseline --no-extras --regions 3 > $OUT/eager1_bench.json 2>/dev/null
stats $OUT/eager1 $OUT/eager_1stream_kernel_stats.csv; rm -rf $OUT/eager1
# one stream: the capacity graph (fresh feed) against the baked graph of one prepared batch, and the host cost of the feed
rocprofv3 --kernel-trace --stats -d $OUT/cap1 -o c --output-format csv -- python3 bench.py --streams 1 --no-cpu-baseline --no-extras --regions 3 > $OUT/cap1_bench.json 2>/dev/null
stats $OUT/cap1 $OUT/capacity_graph_1stream_kernel_stats.csv; rm -rf $OUT/cap1
rocprofv3 --kernel-trace --stats -d $OUT/bak1 -o b --output-format csv -- python3 bench.py --streams 1 --feed replay --no-cpu-baseline --no-extras --regions 3 > $OUT/bak1_bench.json 2>/dev/null
stats $OUT/bak1 $OUT/baked_graph_1stream_kernel_stats.csv; rm -rf $OUT/bak1
python3 tools/feed_cost.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/feed_cost.log
python3 tools/host_pieces.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/host_pieces.log
(python3 tools/kd_host_lead.py kd; python3 tools/kd_host_lead.py teacher) 2>&1 | grep "host enqueue" > $OUT/kd_host_lead.log
E1="python3 bench.py --streams 1 --eager --feed replay --no-cpu-baseline --no-extras --steps 6 --warmup 2 --regions 1"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- $E1 > /dev/null 2>&1
pmc $OUT/pmc_fetch $OUT/pmc_fetch_size.csv pmc_summary.py
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- $E1 > /dev/null 2>&1
pmc $OUT/pmc_write $OUT/pmc_write_size.csv pmc_summary.py
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $OUT/pmc_sq -o s --output-format csv -- $E1 > /dev/null 2>&1
pmc $OUT/pmc_sq $OUT/pmc_sq_counters.csv pmc_multi.py
# the exact-fp32 mode's own traffic pair (FCL_PRECISION=0: the LDS-DMA kernels on fp32 lines since round 4) and kernel stats
FCL_PRECISION=0 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_ffetch -o f --output-format csv -- $E1 > /dev/null 2>&1
pmc $OUT/pmc_ffetch $OUT/pmc_fp32_fetch_size.csv pmc_summary.py
FCL_PRECISION=0 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_fwrite -o w --output-format csv -- $E1 > /dev/null 2>&1
pmc $OUT/pmc_fwrite $OUT/pmc_fp32_write_size.csv pmc_summary.py
FCL_PRECISION=0 rocprofv3 --kernel-trace --stats -d $OUT/eagerf -o e --output-format csv -- python3 bench.py --streams 1 --eager --feed replay --no-cpu-baseline --no-extras --regions 3 > /dev/null 2>&1
stats $OUT/eagerf $OUT/fp32_eager_1stream_kernel_stats.csv; rm -rf $OUT/eagerf
# training workloads: bench JSON (with the CPU baseline) in both arithmetic modes, kernel stats and PMC traffic of the KD and teacher steps
for w in kd_step teacher_step; do
  python3 bench.py --workload $w > $OUT/bench_$w.json 2> /dev/null
  python3 bench.py --workload $w --amp bf16 --no-cpu-baseline > $OUT/bench_${w}_bf16.json 2> /dev/null
  # (--no-dp-schedule: the profiled update is the update alone, not followed by the one-rank RCCL leg's three more timed regions -- ADVICE r4)
  rocprofv3 --kernel-trace --stats -d $OUT/$w -o t --output-format csv -- python3 bench.py --workload $w --steps 10 --warmup 3 --regions 1 --no-cpu-baseline --no-dp-schedule > /dev/null 2>&1
  stats $OUT/$w $OUT/${w}_kernel_stats.csv; rm -rf $OUT/$w
  T1="python3 bench.py --workload $w --steps 3 --warmup 1 --regions 1 --no-cpu-baseline --no-overlap --no-dp-schedule"
  rocprofv3 --pmc FETCH_SIZE -d $OUT/pf -o f --output-format csv -- $T1 > /dev/null 2>&1
  pmc $OUT/pf $OUT/pmc_${w}_fetch_size.csv pmc_summary.py
  rocprofv3 --pmc WRITE_SIZE -d $OUT/pw -o w --output-format csv -- $T1 > /dev/null 2>&1
  pmc $OUT/pw $OUT/pmc_${w}_write_size.csv pmc_summary.py
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $OUT/ps -o s --output-format csv -- $T1 > /dev/null 2>&1
  pmc $OUT/ps $OUT/pmc_${w}_sq_counters.csv pmc_multi.py
done
# the weight-gradient GEMM shapes of both updates: old kernel / dw_mfma_kernel (round 5)
python3 tools/bench_dw2.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/bench_dw_shapes.log
# round 5: the BiLSTM recurrences old / new at both widths, the small-M GEMM tile forms, the per-shape GEMM tables and the KD phase stamps
python3 tools/bilstm_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/bilstm_bench.log
BILSTM_BENCH_MODEL=teacher python3 tools/bilstm_bench.py 2>&1 | grep -v -i "warn\|amdgpu.ids" >> $OUT/bilstm_bench.log
python3 tools/time_smallm.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/smallm_tiles.log
python3 tools/gemm_shapes.py teacher 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/gemm_shapes_teacher_step.log
python3 tools/gemm_shapes.py kd 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/gemm_shapes_kd_step.log
python3 tools/kd_phases.py 2>&1 | grep -v -i "warn\|amdgpu.ids" > $OUT/kd_phases.log
python3 bench.py --workload forward_tf > $OUT/bench_forward_tf.json 2> /dev/null
python3 bench.py --batch 64 --no-cpu-baseline --no-extras > $OUT/bench_batch64.json 2> /dev/null
python3 bench.py --model teacher --no-cpu-baseline --no-extras > $OUT/bench_teacher_synthesis.json 2> /dev/null
python3 bench.py --feed replay --no-cpu-baseline --no-extras > $OUT/bench_replay.json 2> /dev/null
FCL_PRECISION=0 python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_fp32_exact.json 2> /dev/null
# BASELINE configs[4]: phoneme -> waveform (FCL-taco2-S + Parallel WaveGAN), the generator alone
python3 bench.py --workload tts_e2e --steps 5 --warmup 2 > $OUT/bench_tts_e2e.json 2> /dev/null
python3 bench.py --workload vocoder --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_vocoder.json 2> /dev/null
python3 tools/bench_decode.py 4096 2>&1 | grep -v -i warn | grep depth > $OUT/bench_decode.log
ls -la $OUT | head -60
