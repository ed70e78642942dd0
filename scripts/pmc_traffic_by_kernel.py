"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass, as
MI355X_MICROARCH.md prescribes), for this repo's kernels: bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE
reports half of the bytes of wide coalesced streaming reads).

    python scripts/pmc_traffic_by_kernel.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> "<command description>"
"""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from collect_profiles import OURS, short  # noqa: E402


def main(fetch_dir, write_dir, what):
    per = {}
    for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        hits = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)
        if not hits:
            raise SystemExit(f"no counter_collection.csv under {d}")
        with open(max(hits, key=os.path.getmtime)) as fh:
            for row in csv.DictReader(fh):
                if any(k in row["Kernel_Name"] for k in OURS) and row["Counter_Name"] == counter:
                    name = re.sub(r"^void ", "", row["Kernel_Name"]).replace("(anonymous namespace)::", "")
                    name = re.sub(r"\(.*$", "", name).strip()
                    per.setdefault(name, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
    kernels, total = {}, 0.0
    for name, c in sorted(per.items()):
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
            w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
            b = (2.0 * f + w) * 1024.0
            kernels[name] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_corrected": b, "launches": len(c["FETCH_SIZE"]),
                             "hbm_bytes_all_launches": b * len(c["FETCH_SIZE"])}
            total += b * len(c["FETCH_SIZE"])
    print(json.dumps({"command": f"rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python3 {what} (one counter per pass)",
                      "unit": "KB per launch (rocprofv3 derived metric), averaged over the launches; hbm_bytes_corrected = (2 FETCH_SIZE + WRITE_SIZE) * 1024",
                      "hbm_bytes_all_kernels_all_launches": total, "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
