#!/usr/bin/env python3
"""Where is the GPU idle?  From a rocprofv3 kernel trace: idle time between consecutive kernels, summed by (kernel before -> kernel after).
    python scripts/debug/gap_pairs.py <kernel_trace.csv> [min gap us] [top n]"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n).replace("(anonymous namespace)::", "").replace("at::native::", "")
    return re.sub(r"[<(].*$", "", n)[:44]


rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
# the busiest window only: skip everything before the last long pause (setup, CPU baseline ...)
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 50e6:
        cut = i
rows = rows[cut:]
pairs = collections.defaultdict(lambda: [0.0, 0])
idle = 0.0
end = rows[0][1]
for i in range(1, len(rows)):
    gap = (rows[i][0] - end) / 1e3
    if gap >= min_gap:
        k = (short(rows[i - 1][2]), short(rows[i][2]))
        pairs[k][0] += gap
        pairs[k][1] += 1
        idle += gap
    end = max(end, rows[i][1])
span = (rows[-1][1] - rows[0][0]) / 1e3
print(f"window {span / 1e3:.1f} ms, {len(rows)} launches, idle in gaps >= {min_gap} us: {idle / 1e3:.2f} ms ({100 * idle / span:.1f} %)")
for (a, b), (t, n) in sorted(pairs.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"  {t / 1e3:7.3f} ms  {n:5d} x {t / n:7.1f} us   {a}  ->  {b}")
