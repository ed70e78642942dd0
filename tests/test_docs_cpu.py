"""DESIGN.md quotes measured numbers; the measurements live under ``profiles/r06`` (rocprofv3 CSVs, bench JSON
lines, ``SUMMARY.md`` generated from them by ``scripts/profiles_summary.py``).  Round 2's verdict found three
numbers in the docs that no committed file held.  These tests tie the headline figures of DESIGN.md section 5 to
the committed files mechanically: a re-profile that is not followed by a doc update fails here."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles", "r06")


def _line(name):
    with open(os.path.join(PROF, name)) as fh:
        return json.loads([l for l in fh.read().splitlines() if l.startswith("{")][-1])


def _design():
    with open(os.path.join(ROOT, "DESIGN.md")) as fh:
        return fh.read()


def _avg_us(csv_name, prefix):
    with open(os.path.join(PROF, csv_name)) as fh:
        for row in csv.DictReader(fh):
            name = row["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            if name.startswith(prefix):
                return float(row["AverageNs"]) / 1e3
    raise AssertionError(f"{prefix} not in {csv_name}")


def test_headline_numbers_in_design_are_the_committed_bench_lines():
    text = _design()
    pf, drv, ekf = _line("bench_door_pf_n1.json"), _line("bench_driver_flags_door_pf.json"), _line("bench_door_ekf_n1.json")
    assert f"{pf['value'] / 1e9:.3f}e9" in text, "headline value (bench_door_pf_n1.json)"
    assert f"{pf['ms_per_step']:.3f}" in text
    assert f"{drv['value'] / 1e9:.3f}e9" in text, "driver-flags value (bench_driver_flags_door_pf.json)"
    assert f"{ekf['value'] / 1e6:.3f}e6" in text, "EKF value (bench_door_ekf_n1.json)"
    assert drv["steps"] == 20 and drv["warmup"] == 5 and pf["steps"] == 128
    # the strict-mode parity the line carries
    strict = pf["parity_vs_oracle"]["strict_f32_free_running"]
    assert strict["differing_ancestors"] == 0 and strict["differing_estimate_values"] == 0 and strict["rmse_rel_diff"] == 0.0
    # round 4: the weight regime the line was measured in, the certificate taken there, and every BASELINE config
    ess = pf["ess_over_m"]
    assert ess["steps_with_batch_mean_in_band"] == ess["steps"] == 128
    assert 0.05 <= ess["per_step_batch_mean_min"] and ess["per_step_batch_mean_max"] <= 0.5
    assert f"{ess['per_step_batch_mean_min']:.3f}" in text and f"{ess['per_step_batch_mean_max']:.3f}" in text
    for key in ("teacher_forced", "teacher_forced_f32"):
        cert = pf["parity_vs_oracle"][key]["mismatch_certificate"]
        assert cert["unexplained"] == 0 and cert["k1_inexact_on_own_weights"] == 0, key
        assert pf["parity_vs_oracle"][key]["max_rel_err_posterior_mean"] < 1e-4, key
    assert {"C2", "C3", "C4", "C5"} <= {k[:2] for k in pf["configs"]} and sum(k.startswith("blackout") for k in pf["configs"]) == 2
    c5 = next(v for k, v in pf["configs"].items() if k.startswith("C5"))
    assert f"{c5['ms_per_step']:.1f}" in text, "C5 training step (configs of bench_door_pf_n1.json)"
    # round 5: the oracle's own reproducibility floor sits beside the engine's numbers, in the line and in the text
    floor = pf["parity_vs_oracle"]["oracle_self"]
    assert floor["fp32_one_thread"]["teacher_forced"]["resample_index_mismatch_fraction"] == 0.0
    assert f"{floor['fp64']['teacher_forced']['resample_index_mismatch_fraction']:.2e}" in text
    assert f"{pf['parity_vs_oracle']['teacher_forced']['resample_index_mismatch_fraction']:.2e}" in text
    train = _line("bench_push_train_n1.json")
    assert f"{train['ms_per_step']:.1f}" in text and train["config"]["world_size_seen"] == 1


def test_design_is_one_tracked_file():
    """Round 4's verdict: track the text or its template, not both."""
    assert not os.path.exists(os.path.join(ROOT, "DESIGN.md.in")) and not os.path.exists(os.path.join(ROOT, "scripts", "fill_design_numbers.py"))
    assert "@@" not in _design()


def test_kernel_durations_in_design_are_the_committed_rocprof_averages():
    text = _design()
    for csv_name, prefix in (("door_pf_kernel_stats.csv", "particle_net_kernel<3, 2, 1, 2, 1, 2, true>"),
                             ("door_pf_kernel_stats.csv", "particle_net_kernel<3, 3, 0, 2, 1, 2, true>"),
                             ("door_pf_kernel_stats.csv", "pf_resample_systematic_kernel<3, true>"),
                             ("door_ekf_kernel_stats.csv", "image_encoder_resident_kernel<false>")):
        us = _avg_us(csv_name, prefix)
        assert f"{us:.1f}" in text, f"{prefix}: {us:.1f} us ({csv_name}) is not what DESIGN.md quotes"
    # the roofline fraction follows from the measurement kernel's average: 6.067e10 FLOP per launch
    us = _avg_us("door_pf_kernel_stats.csv", "particle_net_kernel<3, 2, 1, 2, 1, 2, true>")
    frac = 6.067e10 / (us * 1e-6) / 2.5e15
    assert f"{frac:.3f}" in text


def test_summary_is_what_the_script_generates_from_the_committed_files():
    with open(os.path.join(PROF, "SUMMARY.md")) as fh:
        committed = fh.read()
    out = subprocess.run([sys.executable, os.path.join("scripts", "profiles_summary.py"), os.path.join("profiles", "r06"), "--stdout"],
                         capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-1000:]
    assert out.stdout.strip() == committed.strip()


def test_bench_traffic_is_the_newest_committed_profile():
    """Round 5's verdict: ``roofline.traffic`` was read from an older round than the one the line was measured in.  The
    bench now walks ``profiles/rNN`` newest first and names the file (``traffic_source``); this ties both to the files."""
    sys.path.insert(0, ROOT)
    import bench

    newest = bench.profile_rounds()[0]
    assert os.path.basename(newest) == os.path.basename(PROF)
    with open(os.path.join(PROF, "pmc_hbm_traffic.json")) as fh:
        k = json.load(fh)["kernels"]
    want = next(v for name, v in k.items() if name.startswith("particle_net_kernel<3, 2, 1, 2, 1, 2, true>"))["hbm_bytes_corrected"]
    got, src = bench.pmc_traffic("particle_net_kernel<3, 2, 1, 2, 1, 2, true>")
    assert got == want and src == os.path.join("profiles", os.path.basename(PROF), "pmc_hbm_traffic.json")
    with open(os.path.join(PROF, "pmc_hbm_traffic_ekf.json")) as fh:
        k = json.load(fh)["kernels"]
    want = sum(next(v for name, v in k.items() if name.startswith(prefix))["hbm_bytes_corrected"] for prefix in bench.K4_SEQUENCES[0])
    got, src = bench.pmc_traffic_k4_ekf()
    assert got == want and src == os.path.join("profiles", os.path.basename(PROF), "pmc_hbm_traffic_ekf.json")
    # the committed lines of this round carry the same pair (scripts/profile_round_r06.sh installs its counter passes before
    # it runs the bench lines, so a line and the file it names come from the same run)
    for name, traffic in (("bench_door_pf_n1.json", bench.pmc_traffic("particle_net_kernel<3, 2, 1, 2, 1, 2, true>")),
                          ("bench_door_ekf_n1.json", bench.pmc_traffic_k4_ekf())):
        roof = _line(name)["roofline"]
        assert roof["traffic_source"] == traffic[1] and roof["traffic"] == traffic[0], name
    # and a kernel no round profiled has no traffic rather than a stale one
    assert bench.pmc_traffic("no_such_kernel") == (None, None)
