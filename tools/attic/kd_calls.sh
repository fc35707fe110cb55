# Per-kernel launch counts of the KD step (rocprofv3 kernel stats sorted by calls): where the host-enqueue time of an update goes.
OUT=gpurun_out/kdprof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT -o kd --output-format csv -- python3 bench.py --workload ${1:-kd_step} --amp ${2:-none} --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2>/dev/null
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r["Calls"]))
print("total calls", sum(int(r["Calls"]) for r in rows), "over 13 updates")
rows.sort(key=lambda r: -float(r["Percentage"]))
for r in rows[:28]:
    print("%6d %8.1f us avg  %5.2f%%  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:110]))
PY
