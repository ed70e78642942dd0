"""GPU-busy fraction of the dense regions of a rocprofv3 kernel trace.

    python scripts/gpu_busy.py <..._kernel_trace.csv> [--gap-ms 2] [--top 6]

Kernels are grouped into regions separated by idle gaps longer than ``--gap-ms`` (host-side setup,
synchronisation points); for each region: span, summed kernel time, busy = sum / span, launches, and the
mean idle gap between consecutive kernels.  The largest regions are the timed loops.
"""
import argparse
import csv
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--gap-ms", type=float, default=2.0)
    ap.add_argument("--top", type=int, default=6)
    ap.add_argument("--kernels", type=int, default=3, help="kernels listed per region")
    ap.add_argument("--gaps", type=int, default=0, help="list the N largest idle gaps of each region (us, kernel before -> after)")
    args = ap.parse_args()
    rows = []
    with open(args.trace) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    regions, cur = [], []
    for s, e, n in rows:
        if cur and s - max(x[1] for x in cur[-8:]) > args.gap_ms * 1e6:
            regions.append(cur)
            cur = []
        cur.append((s, e, n))
    if cur:
        regions.append(cur)
    out = []
    for reg in regions:
        span = max(x[1] for x in reg) - reg[0][0]
        busy = sum(e - s for s, e, _ in reg)
        names = {}
        for s, e, n in reg:
            k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
            names[k] = names.get(k, 0) + (e - s)
        top = sorted(names.items(), key=lambda kv: -kv[1])[:args.kernels]
        gaps = []
        if args.gaps:
            end = reg[0][1]
            prev = reg[0][2]
            for s0, e0, n0 in reg[1:]:
                if s0 > end:
                    gaps.append((round((s0 - end) / 1e3, 1), round((s0 - reg[0][0]) / 1e6, 3), prev.split("(")[0][-40:], n0.split("(")[0][-40:]))
                if e0 > end:
                    end, prev = e0, n0
            gaps.sort(reverse=True)
        out.append({"gaps_us_at_ms_before_after": gaps[: args.gaps], "span_ms": span / 1e6, "kernel_ms": busy / 1e6, "busy": busy / max(span, 1), "launches": len(reg),
                    "mean_gap_us": (span - busy) / max(len(reg) - 1, 1) / 1e3,
                    "top": [(k, round(v / 1e6, 3)) for k, v in top]})
    out.sort(key=lambda r: -r["span_ms"])
    for r in out[: args.top]:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
