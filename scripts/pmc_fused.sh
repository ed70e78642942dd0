# SQ counters + effective clock of the fused training kernels (csrc/particle_net_fused.hip), round 5.
#   bash scripts/pmc_fused.sh [tag]   (GPU box)  -> gpurun_out/pmc_fused/<tag>.json
# Workload: scripts/debug/fused_check.py --sizes 32x8192 --quiet --timing (24 launches per network kind).
# Two counter passes (8 SQ slots each), --kernel-trace only beside them.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-base}
OUT=$R/gpurun_out/pmc_fused
P=/tmp/mmf_pmc_fused_$TAG
rm -rf $P; mkdir -p $OUT $P
cd /tmp && export TMPDIR=/tmp
CMD="$R/scripts/debug/fused_check.py --sizes 32x8192 --quiet --timing"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $P/a -o p -- python3 $CMD > /dev/null 2> $P/a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $P/b -o p -- python3 $CMD > /dev/null 2> $P/b.err
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $P/c -o p -- python3 $CMD > /dev/null 2> $P/c.err
cd $R
python3 - "$P" "$OUT/$TAG.json" <<'PY'
import collections, csv, glob, json, re, sys
src, dst = sys.argv[1:3]
dur = collections.defaultdict(list)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
def key(name):
    m = re.search(r"particle_net_train_fused_kernel<(\d+), (\d+), (\d+), (\d+)>", name)
    if not m:
        return None
    part = {"0": "full", "1": "trunk", "2": "enc", "3": "enc_fwd"}[m.group(4)]
    return f"d{m.group(1)}_{'meas' if m.group(3) == '1' else 'dyn'}_{part}"
for f in glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for f in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = key(r["Kernel_Name"])
        if k:
            cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "scripts/pmc_fused.sh: rocprofv3 --pmc <8 counters> --kernel-trace -- python3 scripts/debug/fused_check.py --sizes 32x8192 --quiet --timing, three passes",
       "unit": "counter value per launch (sum over the chip), averaged over the launches; us = kernel-trace duration under the counters", "kernels": {}}
for k in sorted(dur):
    c = {n: sum(v) / len(v) for n, v in sorted(cnt[k].items())}
    us = sum(dur[k]) / max(len(dur[k]), 1)
    row = {"launches": len(dur[k]), "avg_us_under_pmc": round(us, 2), "counters": {n: round(v) for n, v in c.items()}}
    if "GRBM_GUI_ACTIVE" in c and us > 0:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        row["effective_clock_GHz"] = round(cyc / us * 1e-3, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row["mfma_busy_fraction"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 3)
        if "SQ_INSTS_VALU" in c and c.get("SQ_INSTS_MFMA"):
            row["valu_per_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"], 2)
    out["kernels"][k] = row
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
