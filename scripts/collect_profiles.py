"""Copy the judged summaries of a `scripts/profile_round_r05.sh` run from gpurun_out/ into profiles/<round>/.

    python scripts/collect_profiles.py gpurun_out/prof_final profiles/r01

Keeps: rocprofv3 kernel stats / domain stats, the kernel-trace and PMC rows of this repo's
kernels, the per-kernel HBM traffic derived from the two PMC passes (FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950), and the bench JSON lines.
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

OURS = ("particle_net_kernel", "pf_reweight_resample_kernel", "pf_resample_systematic_kernel", "particle_net_train", "small_grads_kernel", "combine_", "dyn_epilogue", "philox_", "ekf_step_kernel", "conv_kernel",
        "conv_f16x3_kernel", "fc_partial_kernel", "fc_partial_f16x3_kernel", "fc_tail_kernel", "weight_grad_kernel", "pf_init_particles_kernel", "fuse_sensors_kernel", "traj_program_kernel",
        "pack_particle_net_kernel", "pack_encoder_kernel", "image_encoder_resident_kernel", "stem_conv2a_kernel", "conv2b_conv3_kernel", "conv4_kernel",
        "ukf_sigma_points_kernel", "ukf_moments_kernel", "ekf_")


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name).strip()


def one(pattern):
    hits = glob.glob(pattern, recursive=True)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return max(hits, key=os.path.getmtime)  # gpurun_out/ accumulates runs: take the latest


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    shutil.copy(one(f"{src}/stats/**/*_kernel_stats.csv"), f"{dst}/door_pf_kernel_stats.csv")
    shutil.copy(one(f"{src}/stats/**/*_domain_stats.csv"), f"{dst}/door_pf_domain_stats.csv")
    with open(one(f"{src}/stats/**/*_kernel_trace.csv")) as fh, open(f"{dst}/door_pf_kernel_trace_mmf_kernels.csv", "w") as out:
        rd = csv.reader(fh)
        wr = csv.writer(out)
        header = next(rd)
        wr.writerow(header)
        col = header.index("Kernel_Name")
        for row in rd:
            if any(k in row[col] for k in OURS):
                wr.writerow(row)
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with open(one(f"{src}/pmc_{counter}/**/*_counter_collection.csv")) as fh, \
                open(f"{dst}/pmc_{counter}_mmf_kernels.csv", "w") as out:
            rd = csv.DictReader(fh)
            wr = csv.DictWriter(out, fieldnames=rd.fieldnames)
            wr.writeheader()
            for row in rd:
                if any(k in row["Kernel_Name"] for k in OURS) and row["Counter_Name"] == counter:
                    wr.writerow(row)
                    per.setdefault(short(row["Kernel_Name"]), {}).setdefault(counter, []).append(float(row["Counter_Value"]))
    kernels = {}
    for name, c in sorted(per.items()):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        kernels[name] = {"FETCH_SIZE": f, "launches_FETCH_SIZE": len(c["FETCH_SIZE"]),
                         "WRITE_SIZE": w, "launches_WRITE_SIZE": len(c["WRITE_SIZE"]),
                         "hbm_bytes_raw": (f + w) * 1024.0, "hbm_bytes_corrected": (2.0 * f + w) * 1024.0}
    with open(f"{dst}/pmc_hbm_traffic.json", "w") as fh:
        json.dump({
            "workload": "door crossmodal PF, N=256, M=4096 (bench.py default, f16x3), 1x MI355X",
            "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace --output-format csv -- python3 bench.py "
                       "--steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timers --no-f32-mode (one counter per pass)",
            "unit": "KB per launch (rocprofv3 derived metric), averaged over the launches of the run",
            "note": "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide coalesced streaming "
                    "read; 'hbm_bytes_corrected' doubles FETCH_SIZE. Image-encoder and K7 kernels: launches over the "
                    "warm-up (256 images) and the timed chunk (1024 images) are averaged together.",
            "kernels": kernels}, fh, indent=1)
    # optional extras of the round script: f32-mode and EKF kernel stats, K4-alone traffic
    for tag in ("f32", "ekf"):
        hits = glob.glob(f"{src}/stats_{tag}/**/*_kernel_stats.csv", recursive=True)
        if hits:
            shutil.copy(max(hits, key=os.path.getmtime), f"{dst}/door_{'pf_f32_mode' if tag == 'f32' else 'ekf'}_kernel_stats.csv")
    # HBM traffic of the variants (exact-f32 kernels, in-kernel philox noise): same reduction, own files
    for tag in ("f32", "philox", "ekf"):
        per_v = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            hits = glob.glob(f"{src}/pmc_{tag}_{counter}/**/*_counter_collection.csv", recursive=True)
            if not hits:
                continue
            with open(max(hits, key=os.path.getmtime)) as fh:
                for row in csv.DictReader(fh):
                    if any(k in row["Kernel_Name"] for k in OURS) and row["Counter_Name"] == counter:
                        per_v.setdefault(short(row["Kernel_Name"]), {}).setdefault(counter, []).append(float(row["Counter_Value"]))
        rows_v = {}
        for name, c in sorted(per_v.items()):
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
                w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
                rows_v[name] = {"FETCH_SIZE": f, "WRITE_SIZE": w, "hbm_bytes_corrected": (2.0 * f + w) * 1024.0,
                                "launches": len(c["FETCH_SIZE"])}
        if rows_v:
            with open(f"{dst}/pmc_hbm_traffic_{tag}.json", "w") as fh:
                json.dump({"workload": "door crossmodal EKF, N=1024 (bench.py --workload door_ekf --steps 8 --warmup 0: two launch sequences of 4096 images x 2 encoders)" if tag == "ekf" else
                                       f"door crossmodal PF, N=256, M=4096, variant {tag} (bench.py --{'precision f32' if tag == 'f32' else 'noise philox'})",
                           "unit": "KB per launch; hbm_bytes_corrected = (2 FETCH_SIZE + WRITE_SIZE) * 1024", "kernels": rows_v}, fh, indent=1)
    k4 = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        hits = glob.glob(f"{src}/pmc_k4_{counter}/**/*_counter_collection.csv", recursive=True)
        if not hits:
            continue
        with open(max(hits, key=os.path.getmtime)) as fh:
            for row in csv.DictReader(fh):
                if any(k in row["Kernel_Name"] for k in OURS) and row["Counter_Name"] == counter:
                    key = (short(row["Kernel_Name"]), row["Grid_Size_X"] if "Grid_Size_X" in row else row.get("Grid_Size", ""))
                    k4.setdefault(key, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
    if k4:
        rows = {}
        for (name, grid), c in sorted(k4.items()):
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                f = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
                w = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
                rows[f"{name} grid={grid}"] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_corrected": (2.0 * f + w) * 1024.0,
                                               "launches": len(c["FETCH_SIZE"])}
        with open(f"{dst}/pmc_k4_traffic.json", "w") as fh:
            json.dump({"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --kernel-trace -- python3 scripts/bench_k4.py "
                                  "(fused / bf16 / per-layer paths on 2048x2, 1024x3, 256x2, 32x3 image-encoders)",
                       "note": "bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction); launches grouped by kernel and grid size",
                       "kernels": rows}, fh, indent=1)
    for f in glob.glob(f"{src}/bench_*.json") + glob.glob(f"{src}/pytest_*.txt") + glob.glob(f"{src}/bench_*.txt"):
        shutil.copy(f, dst)
    print("kept", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
