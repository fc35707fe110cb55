# Kernel trace (per-dispatch start / end / stream) of the default synthesis feed -> gpurun_out/kdtrace/synth_trace.csv (tools/trace_analyse.py)
OUT=gpurun_out/kdtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $OUT/raws -o t --output-format csv -- python3 bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 8 --regions 1 $* > $OUT/bench_synth.json 2> $OUT/err_synth.log
f=$(find $OUT/raws -name "*kernel_trace.csv" | head -1)
python3 - "$f" $OUT/synth_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [k for k in ["Queue_Id", "Stream_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Workgroup_Size_X"] if k in rows[0]]
rows = rows[-3000:]
t0 = min(int(r["Start_Timestamp"]) for r in rows)
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(keep)
    for r in rows:
        w.writerow([(int(r[k]) - t0 if k.endswith("Timestamp") else (r[k][:70] if k == "Kernel_Name" else r[k])) for k in keep])
PY
rm -rf $OUT/raws
