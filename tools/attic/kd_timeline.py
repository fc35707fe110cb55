"""Queue-level timeline of the training step from a rocprofv3 kernel trace CSV (developer tool): per HIP stream busy time, the union over
streams (GPU not idle) and the idle gaps of the update's main stream, per update (updates are delimited by adam_kernel).
usage: python tools/kd_timeline.py gpurun_out/kdprof/kd_kernel_trace.csv"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Kernel_Name"]) for r in rows))
    adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[3]]
    lo, hi = adam[-6], adam[-1]
    span = ev[lo:hi]
    n_up = 5
    t0, t1 = span[0][0], max(e[1] for e in span)
    print("wall per update %.2f ms" % ((t1 - t0) / 1e6 / n_up))
    qs = {}
    for s, e, q, n in span:
        qs.setdefault(q, []).append((s, e, n))
    for q, lst in sorted(qs.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in lst)
        gaps = sorted(((lst[i + 1][0] - lst[i][1]) / 1e3 for i in range(len(lst) - 1)))
        small = sum(g for g in gaps if 0 < g < 20)
        print("stream %s: %5d launches/update, busy %.2f ms/update, gaps<20us sum %.2f ms/update (median gap %.1f us)" % (
            q, len(lst) / n_up, busy / 1e6 / n_up, small / 1e3 / n_up, gaps[len(gaps) // 2] if gaps else 0))
    # union
    cur_s, cur_e, tot = None, None, 0
    for s, e, _, _ in span:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    tot += cur_e - cur_s
    print("GPU busy (union over streams) %.2f ms/update = %.0f%% of wall" % (tot / 1e6 / n_up, 100.0 * tot / (t1 - t0)))
    # concurrency histogram
    pts = sorted([(s, 1) for s, e, _, _ in span] + [(e, -1) for s, e, _, _ in span])
    lvl, last, hist = 0, pts[0][0], {}
    for t, d in pts:
        hist[lvl] = hist.get(lvl, 0) + (t - last)
        lvl += d
        last = t
    print("time at concurrency k:", {k: "%.0f%%" % (100.0 * v / (t1 - t0)) for k, v in sorted(hist.items())})


if __name__ == "__main__":
    main()
