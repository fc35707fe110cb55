# developer aid: duration of the fused feat/prenet kernel when it returns after phase k (FCL_FP_DBG=k), from a one-stream eager kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 0 1 2 3; do
  FCL_FP_DBG=$k rocprofv3 --kernel-trace --stats -d gpurun_out/fp_$k -o t --output-format csv -- python3 bench.py --steps 3 --warmup 2 --streams 1 --eager --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/fp_$k/t_kernel_trace.csv")) if "feat_prenet" in r["Kernel_Name"]]
last=rows[-25:]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in last]
print("FP_DBG=$k", " ".join("%.1f"%x for x in d))
PY
done
