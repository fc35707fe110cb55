# rocprofv3 kernel stats of the KD / teacher update (no one-rank RCCL leg) -> gpurun_out/$1/
OUT=gpurun_out/${1:-r5c}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in kd_step teacher_step; do
  rocprofv3 --kernel-trace --stats -d $OUT/$w -o t --output-format csv -- python3 bench.py --workload $w --steps 10 --warmup 3 --regions 1 --no-cpu-baseline --no-dp-schedule > $OUT/prof_$w.json 2> $OUT/prof_$w.err
  find $OUT/$w -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${w}_kernel_stats.csv; rm -rf $OUT/$w
done
