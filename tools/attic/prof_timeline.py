"""Per-dispatch timeline of the LAST pass in a rocprofv3 rocpd (sqlite) kernel trace: duration, gap to the previous kernel, grid (developer tool).
usage: python tools/prof_timeline.py <results.db> [marker-substring of the pass's first kernel]"""
import re
import sqlite3
import subprocess
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "gather_rows_kernelIlE"
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(db.execute("select d.start, d.end, s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from %s d join %s s "
                           "on d.kernel_id=s.id order by d.start" % (kd, ks)))
    names = sorted(set(r[2] for r in rows))
    dem = subprocess.run(["c++filt"] + [n.replace(".kd", "") for n in names], capture_output=True, text=True).stdout.splitlines()
    dm = dict(zip(names, dem))
    idx = [i for i, r in enumerate(rows) if marker in r[2]]
    st = idx[-1] if idx else 0
    prev_end, tot, agg = None, 0.0, {}
    for s, e, n, gx, gy, wg in rows[st:]:
        gap = (s - prev_end) / 1e3 if prev_end else 0
        short = re.sub(r"\(.*", "", dm[n]).replace("void fcl::", "")[:58]
        print("%7.1f us gap %5.1f  grid %5d x %4d  %s" % ((e - s) / 1e3, gap, gx // max(wg, 1), gy, short))
        prev_end = e
        tot += (e - s) / 1e3
        a = agg.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e3
    print("sum kernel us %.1f  span %.1f" % (tot, (rows[-1][1] - rows[st][0]) / 1e3))
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("  %-60s %3d  %8.1f us  avg %6.1f" % (k, c, t, t / c))


if __name__ == "__main__":
    main()
