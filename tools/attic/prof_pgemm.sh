cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/pgprof -o p --output-format csv -- python3 tools/time_pgemm_data.py > /dev/null 2>&1
f=$(find gpurun_out/pgprof -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-160 > gpurun_out/pgprof.log
rocprofv3 --kernel-trace --stats -d gpurun_out/pgprof2 -o q --output-format csv -- tools/probe/mainloop_probe > /dev/null 2>&1
f=$(find gpurun_out/pgprof2 -name "*kernel_stats.csv" | head -1); grep "probe_kernel<6\|probe_kernel<0" $f | cut -c1-160 >> gpurun_out/pgprof.log
rm -rf gpurun_out/pgprof gpurun_out/pgprof2
