# refresh of the exact-fp32 (FCL_PRECISION=0) evidence only: bench line, eager kernel stats, PMC traffic pair -> gpurun_out/prof_$1/
TAG=${1:-r4x}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
stats() { find $1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $2; }
pmc() { f=$(find $1 -name "*counter_collection.csv" | head -1); python3 tools/$3 $f > $2 2>/dev/null; rm -rf $1; }
E1="python3 bench.py --streams 1 --eager --feed replay --no-cpu-baseline --no-extras --steps 6 --warmup 2 --regions 1"
FCL_PRECISION=0 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_ffetch -o f --output-format csv -- $E1 > /dev/null 2>&1
pmc $OUT/pmc_ffetch $OUT/pmc_fp32_fetch_size.csv pmc_summary.py
FCL_PRECISION=0 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_fwrite -o w --output-format csv -- $E1 > /dev/null 2>&1
pmc $OUT/pmc_fwrite $OUT/pmc_fp32_write_size.csv pmc_summary.py
FCL_PRECISION=0 rocprofv3 --kernel-trace --stats -d $OUT/eagerf -o e --output-format csv -- python3 bench.py --streams 1 --eager --feed replay --no-cpu-baseline --no-extras --regions 3 > /dev/null 2>&1
stats $OUT/eagerf $OUT/fp32_eager_1stream_kernel_stats.csv; rm -rf $OUT/eagerf
FCL_PRECISION=0 python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_fp32_exact.json 2> /dev/null
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> /dev/null
ls -la $OUT
