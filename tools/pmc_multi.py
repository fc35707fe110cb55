"""Per-kernel sums of several rocprofv3 PMC counters (counter_collection.csv) -> table."""
import csv
import sys
from collections import defaultdict


def main(path):
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    names = []
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            c = r["Counter_Name"]
            if c not in names:
                names.append(c)
            agg[k][c] += float(r["Counter_Value"])
            if c == names[0]:
                cnt[k] += 1
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches"] + ["mean_" + n for n in names])
    for k in sorted(agg, key=lambda k: -agg[k][names[0]]):
        w.writerow([k, cnt[k]] + ["%.0f" % (agg[k][n] / max(cnt[k], 1)) for n in names])


if __name__ == "__main__":
    main(sys.argv[1])
