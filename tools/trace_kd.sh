# Kernel trace (per-dispatch start / end / queue) of a few KD updates -> gpurun_out/kdtrace/kd_trace.csv (analysed by tools/trace_analyse.py)
W=${1:-kd_step}
OUT=gpurun_out/kdtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $OUT/raw -o t --output-format csv -- python3 bench.py --workload $W --steps 6 --warmup 3 --regions 1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/err.log
f=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python3 - "$f" $OUT/${W}_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ["Queue_Id", "Stream_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X"]
keep = [k for k in keep if k in rows[0]]
rows = rows[-4000:]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(keep)
    for r in rows:
        o = []
        for k in keep:
            v = r[k]
            if k.endswith("Timestamp"): v = int(v) - t0
            if k == "Kernel_Name": v = v[:70]
            o.append(v)
        w.writerow(o)
PY
rm -rf $OUT/raw
ls -la $OUT
