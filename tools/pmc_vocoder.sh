# HBM traffic of the vocoder kernels (separate --pmc passes, MI355X_MICROARCH.md §HBM): bash tools/pmc_vocoder.sh [batch] [frames]
OUT=gpurun_out/pmc_voc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B=${1:-16}; F=${2:-800}
rocprofv3 --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python3 tools/bench_vocoder.py $B $F > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/w -o w --output-format csv -- python3 tools/bench_vocoder.py $B $F > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT/f/f_counter_collection.csv | head -8
python3 tools/pmc_summary.py $OUT/w/w_counter_collection.csv | head -8
rm -rf $OUT/f $OUT/w
