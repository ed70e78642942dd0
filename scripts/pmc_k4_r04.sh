# SQ counters + effective clock of the K4 kernels of the EKF bench (door crossmodal EKF, 1024 trajectories, f16x3), round 4.
#   bash scripts/pmc_k4_r04.sh [tag]   (GPU box)  -> gpurun_out/pmc_k4_r04/<tag>.json
# Two passes (8 SQ slots per pass; GRBM_GUI_ACTIVE rides on both): never combined with a trace domain other than
# --kernel-trace.  MMF_LIB_PATH selects a variant library.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-base}
OUT=$R/gpurun_out/pmc_k4_r04
P=/tmp/mmf_pmc_k4_$TAG
rm -rf $P; mkdir -p $OUT $P
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-precision-study --no-reference-sizes --no-f32-mode --no-kernel-timers --no-configs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $P/a -o p -- python3 $R/bench.py --workload door_ekf --steps 16 --warmup 0 $LEAN --preroll-seconds 0 > /dev/null 2> $P/a.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $P/b -o p -- python3 $R/bench.py --workload door_ekf --steps 16 --warmup 0 $LEAN --preroll-seconds 0 > /dev/null 2> $P/b.err
cd $R
python3 - "$P" "$OUT/$TAG.json" <<'PY'
import collections, csv, glob, json, sys
src, dst = sys.argv[1:3]
want = {"resident": "image_encoder_resident_kernel<false>", "conv2b_conv3": "conv2b_conv3_kernel<false, 2, true>", "stem_conv2a": "stem_conv2a_kernel<false>", "fc_partial": "fc_partial_f16x3_kernel", "jacobian": "particle_net_kernel<3, 3, 2",
        "traj_program": "traj_program_kernel"}
dur = collections.defaultdict(list)
for f in glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, pat in want.items():
            if pat in r["Kernel_Name"]:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
cnt = {k: collections.defaultdict(list) for k in want}
for f in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k, pat in want.items():
            if pat in r["Kernel_Name"]:
                cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"command": "scripts/pmc_k4_r04.sh: rocprofv3 --pmc <8 counters> --kernel-trace -- python3 bench.py --workload door_ekf --steps 16 --warmup 0 --preroll-seconds 0 (lean), two passes",
       "unit": "counter value per launch (sum over the chip), averaged over the launches of the pass; us = kernel-trace duration under the counters",
       "kernels": {}}
for k in want:
    if not dur[k]:
        continue  # a kernel of an earlier round
    c = {n: sum(v) / len(v) for n, v in sorted(cnt[k].items())}
    us = sum(dur[k]) / max(len(dur[k]), 1)
    row = {"launches": len(dur[k]), "avg_us_under_pmc": round(us, 2), "counters": {n: round(v) for n, v in c.items()}}
    if "GRBM_GUI_ACTIVE" in c and us > 0:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # rocprofv3 sums the 8 XCDs
        row["effective_clock_GHz"] = round(cyc / us * 1e-3, 3)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            row["mfma_busy_fraction"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, 3)  # 1024 SIMDs
        if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]:
            row["valu_per_mfma"] = round((c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / c["SQ_INSTS_MFMA"], 2)
    out["kernels"][k] = row
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
PY
