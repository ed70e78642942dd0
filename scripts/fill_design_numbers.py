#!/usr/bin/env python3
"""DESIGN.md is GENERATED: ``DESIGN.md.in`` holds the text with @@NAME@@ placeholders for every number of the round's
sections, this script fills them from the committed files of the profile round, so each such number is read off an
artefact (``tests/test_docs_cpu.py`` checks that DESIGN.md is what this script makes of the template and the files).

    python scripts/fill_design_numbers.py profiles/r04 DESIGN.md.in DESIGN.md [--check | --stdout]
    --check: print the table of values, write nothing;  --stdout: write the filled text to stdout
"""
import csv
import json
import re
import sys


def line(path):
    ls = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(ls[-1])


def lines(path):
    return [json.loads(l) for l in open(path).read().splitlines() if l.startswith("{")]


def avg_us(path, prefix):
    for row in csv.DictReader(open(path)):
        name = row["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        if name.startswith(prefix):
            return float(row["AverageNs"]) / 1e3, int(row["Calls"])
    raise KeyError(prefix)


def e9(v):
    return f"{v / 1e9:.3f}e9"


def e6(v):
    return f"{v / 1e6:.3f}e6"


def main(d, design, out_path, check, to_stdout=False):
    pf, drv, ekf = line(f"{d}/bench_door_pf_n1.json"), line(f"{d}/bench_driver_flags_door_pf.json"), line(f"{d}/bench_door_ekf_n1.json")
    cfg = pf["configs"]
    by = lambda p: next(v for k, v in cfg.items() if k.startswith(p))
    ess, par = pf["ess_over_m"], pf["parity_vs_oracle"]
    ref = pf["reference_sized"]["eval_32x300"]
    v = {}
    v["HEAD"], v["HEADMS"] = e9(pf["value"]), f"{pf['ms_per_step']:.3f}"
    v["DRV"], v["DRVMS"] = e9(drv["value"]), f"{drv['ms_per_step']:.3f}"
    try:  # the same kernels on another box of the pool (an earlier profile round of this round)
        v["HEADOB"], v["DRVOB"] = e9(line(f"{d}/bench_door_pf_n1_other_box.json")["value"]), e9(line(f"{d}/bench_driver_flags_door_pf_other_box.json")["value"])
        v["EKFOB"] = e6(line(f"{d}/bench_door_ekf_n1_other_box.json")["value"])
        v["HEADOBB"], v["DRVOBB"] = e9(line(f"{d}/bench_door_pf_n1_other_box_b.json")["value"]), e9(line(f"{d}/bench_driver_flags_door_pf_other_box_b.json")["value"])
        v["EKFOBB"] = e6(line(f"{d}/bench_door_ekf_n1_other_box_b.json")["value"])
    except OSError:
        pass
    v["ESSLO"], v["ESSHI"] = f"{ess['per_step_batch_mean_min']:.3f}", f"{ess['per_step_batch_mean_max']:.3f}"
    v["BENCHS"] = f"{pf['bench_seconds']:.0f}"
    v["EKF"], v["EKFMS"] = e6(ekf["value"]), f"{ekf['ms_per_step']:.3f}"
    v["EKFDRV"] = e6(line(f"{d}/bench_driver_flags_door_ekf.json")["value"])
    blk = by("blackout_0.4_door_crossmodal_ekf")
    v["BLKRATIO"] = f"{blk['ms_per_step_over_plain_ekf']:.3f}"
    eb, pb = line(f"{d}/bench_door_ekf_blackout.json"), line(f"{d}/bench_door_pf_blackout.json")
    v["EKFBLK"], v["EKFBLKMS"], v["PFBLK"], v["PFBLKMS"] = e6(eb["value"]), f"{eb['ms_per_step']:.3f}", e9(pb["value"]), f"{pb['ms_per_step']:.3f}"
    v["REFMS"], v["REFUS"] = f"{ref['ms_per_step']:.3f}", f"{1e3 * ref['ms_per_step']:.1f}"
    v["PERSREC"], v["ENCUS"] = f"{1e3 * ref['recursion_ms_per_step']:.1f}", f"{1e3 * ref['encoders_ms_per_step']:.1f}"
    c5 = by("C5")
    v["C5MS"], v["C5GB"] = f"{c5['ms_per_step']:.1f}", f"{c5['peak_memory_GB']:.2f}"
    ab = lines(f"{d}/bench_train_compact_ab.txt")  # default, exact-fp32 backward, + exact-fp32 recompute, + fp32 buffers, default again
    v["C5DEF"], v["C5F32BMS"], v["C5F32RMS"], v["C5F32MS"] = (f"{ab[i]['ms_per_train_step']:.1f}" for i in range(4))
    v["F32"], v["CPU"] = e9(pf["f32_mode"]["value"]), f"{pf['cpu_baseline']['value']:.3g}"
    for key, f in (("S800", "bench_door_pf_800_steps.json"), ("PHILOX", "bench_door_pf_philox.json"), ("PUSH", "bench_push_pf_n1.json"),
                   ("N32", "bench_door_pf_n32_m4096.json"), ("N1024", "bench_door_pf_n1024_m4096.json"), ("G2", "bench_gpus2_gloo_one_gpu.json")):
        j = line(f"{d}/{f}")
        v[key], v[key + "MS"] = e9(j["value"]), f"{j['ms_per_step']:.3f}"
    j = line(f"{d}/bench_c4_door_ekf_n8192_one_gpu.json")
    v["C4G"], v["C4GMS"] = e6(j["value"]), f"{j['ms_per_step']:.2f}"
    for key, p in (("C2", "C2"), ("C3", "C3")):
        v[key], v[key + "MS"] = e9(by(p)["value"]), f"{by(p)['ms_per_step']:.3f}"
    v["REFTRAIN"] = f"{pf['reference_sized']['train_e2e_32x30x16']['ms_per_optimiser_step']:.1f}"
    launch = [l for l in lines(f"{d}/bench_persistent_loop_ab.txt") if l.get("regime") == "eval"]
    v["REFLAUNCH"] = f"{launch[-1]['ms_per_step']:.3f}"
    us, calls = avg_us(f"{d}/door_pf_kernel_stats.csv", "particle_net_kernel<3, 2, 1, 2, 1, 2, true, false>")
    v["K2US"], v["K2CALLS"] = f"{us:.1f}", f"{calls:,}"
    v["K2TF"], v["K2FRAC"] = f"{6.067e10 / (us * 1e-6) / 1e12:.0f}", f"{6.067e10 / (us * 1e-6) / 2.5e15:.3f}"
    v["K2FRACLINE"], v["K1FRAC"] = f"{pf['roofline']['frac']:.3f}", f"{pf['roofline_k1']['frac']:.3f}"
    v["DYNUS"] = f"{avg_us(f'{d}/door_pf_kernel_stats.csv', 'particle_net_kernel<3, 3, 0, 2, 1, 2, true, false>')[0]:.1f}"
    v["K1US"] = f"{avg_us(f'{d}/door_pf_kernel_stats.csv', 'pf_resample_systematic_kernel<3, true>')[0]:.1f}"
    v["EKFC23"] = f"{avg_us(f'{d}/door_ekf_kernel_stats.csv', 'conv2b_conv3_kernel<false, 2, true>')[0]:.1f}"
    v["EKFSTEM"] = f"{avg_us(f'{d}/door_ekf_kernel_stats.csv', 'stem_conv2a_kernel<false>')[0]:.1f}"
    v["EKFFRAC"] = f"{ekf['roofline']['frac']:.3f}"
    tr = json.load(open(f"{d}/pmc_hbm_traffic.json"))["kernels"]
    mb = lambda p: next(x["hbm_bytes_corrected"] for n, x in tr.items() if n.startswith(p)) / 1e6
    v["TRM"], v["TRD"], v["TRK1"] = (f"{mb(p):.1f}" for p in ("particle_net_kernel<3, 2, 1", "particle_net_kernel<3, 3, 0", "pf_resample_systematic_kernel<3"))
    k2 = json.load(open(f"{d}/pmc_k2_sq_counters.json"))
    m0, m4 = k2["variant0_column_half_pipeline"]["measurement"], k2["variant4_row_tile_pipeline"]["measurement"]
    v["MFMABUSY0"], v["CLK0"], v["MFMABUSY4"], v["CLK4"] = (f"{m0['mfma_busy_fraction']:.3f}", f"{m0['effective_clock_GHz']:.2f}",
                                                          f"{m4['mfma_busy_fraction']:.3f}", f"{m4['effective_clock_GHz']:.2f}")
    try:
        k4 = json.load(open(f"{d}/pmc_k4_sq_counters.json"))["kernels"]
        c23, cst = k4["conv2b_conv3"], k4["stem_conv2a"]
        v["K4BUSY23"], v["K4CLK23"] = f"{c23['mfma_busy_fraction']:.3f}", f"{c23['effective_clock_GHz']:.2f}"
        v["K4BUSYSTEM"], v["K4CLKSTEM"] = f"{cst['mfma_busy_fraction']:.3f}", f"{cst['effective_clock_GHz']:.2f}"
        v["K4CONF23"] = f"{c23['counters']['SQ_LDS_BANK_CONFLICT'] / c23['counters']['SQ_INSTS_LDS']:.2f}"
        v["K4CONFSTEM"] = f"{cst['counters']['SQ_LDS_BANK_CONFLICT'] / cst['counters']['SQ_INSTS_LDS']:.2f}"
    except OSError:
        pass
    bare = float(re.search(r"32x32x16 fill 0: .*? ([0-9.]+) TFLOP/s", open(f"{d}/ubench_mfma_shape.txt").read()).group(1))
    v["BARE"], v["BAREFRAC"] = f"{bare:,.0f}", f"{bare / 2500:.3f}"
    exec_pf = lambda m: m["counters"]["SQ_INSTS_MFMA"] * 32768.0 / (m["avg_us_under_pmc"] * 1e-6) / 1e15
    v["K2EXECPF"], v["K2EXECPF4"] = f"{exec_pf(m0):.2f}", f"{exec_pf(m4):.2f}"
    v["LDS0"], v["LDS4"] = f"{m0['counters']['SQ_INSTS_LDS'] / 1e6:.2f}", f"{m4['counters']['SQ_INSTS_LDS'] / 1e6:.2f}"
    sweep = {j["N"]: j["us"] for j in lines(f"{d}/bench_k1_dephase_ab.txt")}
    for n in (64, 128, 256, 512, 1024):
        v[f"K1S{n}"] = f"{sweep[n]:.1f}"
    v["K1SWEEP"] = " / ".join(f"{sweep[n]:.1f}" for n in (64, 128, 256, 512, 1024))
    wg = avg_us(f"{d}/train_kernel_stats.csv", "weight_grad_h_kernel")[0]
    v["WGUS"] = f"{wg:.0f}"
    v["WGTBS"] = f"{0.5625e9 / (wg * 1e-6) / 1e12:.1f}"  # 8 layers x 262,144 rows x (128 B dz + 128 B stash + 4 B scale)
    with open(f"{d}/train_kernel_stats.csv") as fh:
        rows = [r for r in csv.DictReader(fh) if "particle_net_train_bwd_kernel" in r["Name"]]
    steps = sum(int(r["Calls"]) for r in rows) / 45.0  # 15 transitions x 3 networks per optimiser step
    v["BWDMS"] = f"{sum(int(r['TotalDurationNs']) for r in rows) / 1e6 / steps:.1f}"
    with open(f"{d}/train_kernel_stats.csv") as fh:
        allrows = list(csv.DictReader(fh))
    per_step = lambda key: sum(int(r["TotalDurationNs"]) for r in allrows if key in r["Name"]) / 1e6 / steps
    v["FWDMS"], v["WGMS"] = f"{per_step('particle_net_train_fwd_kernel'):.1f}", f"{per_step('weight_grad_h_kernel'):.1f}"
    t16, t32 = par["teacher_forced"], par["teacher_forced_f32"]
    v["TFMM16"], v["TFMM32"] = f"{t16['resample_index_mismatch_fraction']:.1e}", f"{t32['resample_index_mismatch_fraction']:.1e}"
    v["DQ"] = f"{max(t16['mismatch_certificate']['max_D_over_Q'], t32['mismatch_certificate']['max_D_over_Q']):.1e}"
    v["TFMEAN"] = f"{max(t16['max_rel_err_posterior_mean'], t32['max_rel_err_posterior_mean']):.1e}"
    gpu = re.search(r"(\d+) passed", open(f"{d}/pytest_gpu.txt").read()).group(1)
    v["NGPU"] = gpu
    text = open(design).read()
    names = set(re.findall(r"@@([A-Z0-9]+)@@", text))
    missing = sorted(names - set(v))
    if check:
        for k in sorted(v):
            print(f"{k:12s} {v[k]}")
    if missing:
        print("no value for:", missing, file=sys.stderr)
    if not check:
        for k, val in v.items():
            text = text.replace(f"@@{k}@@", val)
        if to_stdout:
            sys.stdout.write(text)
            return
        open(out_path, "w").write(text)
        print(f"filled {len(names) - len(missing)} placeholders, {len(missing)} left -> {out_path}")


if __name__ == "__main__":
    flags = [a for a in sys.argv[1:] if a.startswith("--")]
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(pos[0].rstrip("/"), pos[1], pos[2] if len(pos) > 2 else pos[1].replace(".in", ""), "--check" in flags, "--stdout" in flags)
