# Collects the round's profile evidence on the GPU box into gpurun_out/prof_$1/ (copy the summaries to profiles/ afterwards):
#   default bench JSON, rocprofv3 kernel stats (hipGraph x 4 streams = the default command; eager x 1 stream), PMC FETCH_SIZE / WRITE_SIZE (separate passes)
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats -d $OUT/graph4 -o g --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/graph4_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/eager1 -o e --output-format csv -- python3 bench.py --streams 1 --eager --no-cpu-baseline > $OUT/eager1_bench.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f --output-format csv -- python3 bench.py --streams 1 --eager --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w --output-format csv -- python3 bench.py --streams 1 --eager --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS -d $OUT/pmc_sq -o s --output-format csv -- python3 bench.py --streams 1 --eager --no-cpu-baseline --steps 6 --warmup 2 > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch/f_counter_collection.csv > $OUT/pmc_fetch_size.csv
python3 tools/pmc_summary.py $OUT/pmc_write/w_counter_collection.csv > $OUT/pmc_write_size.csv
python3 tools/pmc_multi.py $OUT/pmc_sq/s_counter_collection.csv > $OUT/pmc_sq_counters.csv 2>/dev/null
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq   # raw per-dispatch rows are large; the summaries are what gets committed
ls -la $OUT $OUT/graph4 | head -30
# training workloads: bench JSON (with the CPU baseline) in both arithmetic modes, kernel stats of the KD and teacher steps
for w in kd_step teacher_step; do
  python3 bench.py --workload $w > $OUT/bench_$w.json 2> /dev/null
  python3 bench.py --workload $w --amp bf16 --no-cpu-baseline > $OUT/bench_${w}_bf16.json 2> /dev/null
  rocprofv3 --kernel-trace --stats -d $OUT/$w -o t --output-format csv -- python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
done
python3 bench.py --workload forward_tf > $OUT/bench_forward_tf.json 2> /dev/null
python3 bench.py --batch 64 --no-cpu-baseline > $OUT/bench_b64.json 2> /dev/null
python3 bench.py --model teacher --no-cpu-baseline > $OUT/bench_teacher_synthesis.json 2> /dev/null
# BASELINE configs[4]: phoneme -> waveform (FCL-taco2-S + Parallel WaveGAN), the generator alone, its kernel stats and PMC traffic
python3 bench.py --workload tts_e2e --steps 5 --warmup 2 > $OUT/bench_tts_e2e.json 2> /dev/null
python3 bench.py --workload vocoder --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_vocoder.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/vocoder -o v --output-format csv -- python3 bench.py --workload vocoder --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_vf -o f --output-format csv -- python3 bench.py --workload vocoder --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_vw -o w --output-format csv -- python3 bench.py --workload vocoder --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT/pmc_vf/f_counter_collection.csv > $OUT/pmc_vocoder_fetch_size.csv
python3 tools/pmc_summary.py $OUT/pmc_vw/w_counter_collection.csv > $OUT/pmc_vocoder_write_size.csv
rm -rf $OUT/pmc_vf $OUT/pmc_vw
