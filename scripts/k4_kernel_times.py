#!/usr/bin/env python3
"""Per-kernel medians of the 2048 x 2 launches in a rocprofv3 kernel trace of scripts/bench_k4.py:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/k4t -o k4 -- python3 scripts/bench_k4.py
    python scripts/k4_kernel_times.py gpurun_out/k4t/k4_kernel_trace.csv"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    d[(n, r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    if k[2] == "2" and ("conv" in k[0] or "fc_" in k[0]):  # two encoders: the 2048- and 256-image launches share a grid
        big = [x for x in v if x > 0.5 * max(v)]
        print(f"{k[0]:50s} {len(big):3d} launches  median {statistics.median(big):8.1f} us")
