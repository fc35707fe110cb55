"""Critical-path view of a kernel trace written by tools/trace_kd.sh: per stream busy time, idle gaps, and the kernels in front of the largest gaps."""
import csv
import sys
from collections import Counter, defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
key = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
adam = [r for r in rows if "adam_kernel" in r["Kernel_Name"]]
if len(adam) >= 3:  # one update = between two Adam launches
    lo, hi = adam[-3]["e"], adam[-2]["e"]
else:
    lo, hi = rows[0]["s"], rows[-1]["e"]
win = [r for r in rows if r["s"] >= lo and r["e"] <= hi]
print("window %.3f ms, %d launches" % ((hi - lo) / 1e6, len(win)))
by = defaultdict(list)
for r in win:
    by[r[key]].append(r)
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    v.sort(key=lambda r: r["s"])
    busy = sum(r["e"] - r["s"] for r in v)
    gaps = [(v[i + 1]["s"] - v[i]["e"], v[i]["Kernel_Name"], v[i + 1]["Kernel_Name"]) for i in range(len(v) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print("\n%s %s: %d launches, busy %.3f ms, span %.3f ms, idle between launches %.3f ms (median gap %.1f us)" % (
        key, k, len(v), busy / 1e6, (v[-1]["e"] - v[0]["s"]) / 1e6, sum(g[0] for g in pos) / 1e6, sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0))
    c = Counter()
    for r in v:
        c[r["Kernel_Name"].split("(")[0][:60]] += r["e"] - r["s"]
    for n, t in c.most_common(12):
        print("    %-62s %7.3f ms" % (n, t / 1e6))
    if len(sys.argv) > 2:
        for g in sorted(pos, reverse=True)[:int(sys.argv[2])]:
            print("    gap %7.1f us after %-45s before %s" % (g[0] / 1e3, g[1].split("(")[0][-45:], g[2].split("(")[0][-45:]))
