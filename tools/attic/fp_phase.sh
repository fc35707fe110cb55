# developer aid: duration of the fused feat/prenet kernel when it returns after phase k (FCL_FP_DBG=k), from a one-stream eager kernel trace
# usage: fp_phase.sh [FCL_FP_SPLIT [FCL_FP_SPLIT_RT]]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export FCL_FP_SPLIT=${1:-2} FCL_FP_SPLIT_RT=${2:-0}
for k in 0 1 2 3; do
  FCL_FP_DBG=$k rocprofv3 --kernel-trace --stats -d gpurun_out/fp_$k -o t --output-format csv -- python3 bench.py --steps 3 --warmup 2 --streams 1 --eager --no-cpu-baseline --no-extras > /dev/null 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/fp_$k/t_kernel_trace.csv")) if "feat_prenet" in r["Kernel_Name"]]
last=rows[-25:]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in last]
print("SPLIT=$FCL_FP_SPLIT RT=$FCL_FP_SPLIT_RT FP_DBG=$k", " ".join("%.1f"%x for x in d))
PY
done
