TAG=r2v
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --workload tts_e2e --steps 5 --warmup 2 > $OUT/bench_tts_e2e.json 2> /dev/null
python3 bench.py --workload vocoder --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_vocoder.json 2> /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/vocoder -o v --output-format csv -- python3 bench.py --workload vocoder --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_vf -o f --output-format csv -- python3 bench.py --workload vocoder --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_vw -o w --output-format csv -- python3 bench.py --workload vocoder --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT/pmc_vf/f_counter_collection.csv > $OUT/pmc_vocoder_fetch_size.csv
python3 tools/pmc_summary.py $OUT/pmc_vw/w_counter_collection.csv > $OUT/pmc_vocoder_write_size.csv
rm -rf $OUT/pmc_vf $OUT/pmc_vw
