#!/usr/bin/env python3
"""Benchmark of the filter hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload door_pf|push_pf|door_ekf]

A *step* is one filter time step over one batch: for the particle-filter workloads the
per-trajectory encoders (CNNs, control / observation MLPs, modality weights), the fused
per-particle dynamics + measurement networks (K2) and reweight + resample (K1) for N x M
particles.  The default workload is the one BASELINE.json's metric is quoted on: the door
crossmodal particle filter with 4096 particles (batch 256 trajectories per GPU, weak
scaling: rank r owns its own 256 trajectories; the only collective is the all-gather of
per-sequence squared errors, ``multimodalfilter_amd/distributed.py``).

Inputs (observations, controls, pre-drawn noise) are resident in HBM when the timed region
starts; the timed region is ``forward_loop`` over exactly K steps, bracketed by barrier +
``torch.cuda.synchronize()``; the time is the max over ranks.  Rank 0 prints ONE JSON line.

A particle filter's timed steps are steps ``B+W+1 .. B+W+K`` of ONE filter run over one trajectory:
``B`` = ``BURN_IN`` untimed steps from the reference's initial belief (``states[0]``, ``0.1 I``:
``eval_helpers.py:125-131``), then the W warm-up steps, then the K timed ones, the belief carried
from one ``forward_loop`` call to the next -- so the timed steps are a filter that is TRACKING, with the
measurement heads calibrated (``calibrate_to_band``) to hold ESS/M inside SURVEY.md 8d's [0.05, 0.5]
there (the line reports the engine's own ESS/M over the timed steps: ``ess_over_m``).

``configs`` (one GPU, default on): BASELINE.json's other configurations and the blackout workloads
(``tasks/_door.py:181-197``) as bounded legs of the same run -- each with its value,
ms per step, its dominant kernel's roofline fraction and one parity number against the oracle.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from bench_parity import (CPU_THREADS, _leg_parity_pf, cpu_baseline_ekf, cpu_baseline_pf,  # noqa: E402,F401
                          image_encoder_precision_errors, jacobian_precision_errors, oracle_pf_run, precision_errors, run_pf,
                          strict_parity, teacher_forced_parity, to_device)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)
F16_MFMA_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (MI355X_MICROARCH.md)
MFMA_PEAK = {"f32": FP32_MFMA_PEAK_TFLOPS, "f16x3": F16_MFMA_PEAK_TFLOPS}
PREC_CODE = {"f32": 0, "f16x3": 1}


def k2_kernel_name(d: int, prec: str) -> str:
    """<D, NRES, KIND=measure, CT=2 (64-particle tiles), PREC, WPS=2 waves/SIMD, PIPE (f16x3: the two 32-particle
    halves run half a layer apart)>"""
    pipe = "true" if prec == "f16x3" else "false"
    return f"particle_net_kernel<{d}, 2, 1, 2, {PREC_CODE[prec]}, 2, {pipe}>"


def k2_kernel_key(d: int, prec: str) -> str:
    """Prefix that finds the kernel in this round's AND earlier rounds' profile files (r01-r03 printed seven template
    arguments, r04 eight, r05 seven again)."""
    return ", ".join(k2_kernel_name(d, prec).split(", ")[:6])


WORKLOADS = {
    "door_pf": dict(task="door", cls="DoorCrossmodalParticleFilter", kind="pf", batch=256, particles=4096,
                    desc="door crossmodal particle filter"),
    "push_pf": dict(task="push", cls="PushCrossmodalParticleFilter", kind="pf", batch=256, particles=4096,
                    desc="push crossmodal particle filter"),
    "door_ekf": dict(task="door", cls="DoorCrossmodalKalmanFilter", kind="ekf", batch=1024, particles=1,
                     desc="door crossmodal EKF"),
    # SURVEY.md 8d's second run: image_blackout_ratio 0.4 (tasks/_door.py:181-197) through the filters that KNOW
    # about blackouts (door_models/crossmodal_pf.py:99-104 via ...Seq5; door_models/crossmodal_kf.py:43-98)
    "door_pf_blackout": dict(task="door", cls="DoorCrossmodalParticleFilterSeq5", kind="pf", batch=256, particles=4096,
                             blackout=0.4, desc="door crossmodal particle filter (know_image_blackout), 40 % of the frames blacked out"),
    "door_ekf_blackout": dict(task="door", cls="DoorCrossmodalKalmanFilter", kind="ekf", batch=1024, particles=1,
                              blackout=0.4, ctor=dict(know_image_blackout=True),
                              desc="door crossmodal EKF (know_image_blackout), 40 % of the frames blacked out"),
    # BASELINE config 5 as its own line (round 5): one step = one optimiser step of the push unimodal PF's end-to-end
    # training (train_helpers.py:124-162) on a per-rank batch, gradients all-reduced over the ranks INSIDE the clock
    "push_train": dict(task="push", cls="PushUnimodalParticleFilter", kind="train", batch=32, particles=8192, length=16,
                       desc="push unimodal particle filter, end-to-end training step (forward + backward + gradient all-reduce + SGD)"),
}


def workload_desc(wl, batch, particles, total_batch, world, scaling) -> str:
    """Built from the numbers the run actually used (never a static string)."""
    what = wl["desc"] + (f", {particles} particles" if wl["kind"] == "pf" else "")
    if scaling == "strong":
        return f"{what}, {total_batch} trajectories in total sharded over {world} GPU(s) (this rank: {batch})"
    return f"{what}, batch {batch} trajectories per GPU x {world} GPU(s)"


BURN_IN = 24  # untimed particle-filter steps ahead of the warm-up: the cloud contracts from the 0.1 I
              # initial belief to its tracking width within ~8 steps (ESS/M 0.35 -> 0.9 under round 3's
              # calibration at t = 0); the heads are calibrated to the TRACKING regime


def profile_rounds():
    """``profiles/rNN`` directories, newest round first."""
    import re
    base = os.path.join(ROOT, "profiles")
    try:
        rounds = [d for d in os.listdir(base) if re.fullmatch(r"r\d+", d) and os.path.isdir(os.path.join(base, d))]
    except OSError:
        return []
    return [os.path.join(base, d) for d in sorted(rounds, key=lambda d: int(d[1:]), reverse=True)]


def _kernels_of(path):
    with open(path) as fh:
        return json.load(fh)["kernels"]


def pmc_traffic(kernel_key: str, files=("pmc_hbm_traffic.json", "pmc_hbm_traffic_f32.json")):
    """``(HBM bytes per launch, the committed file they come from)`` -- the rocprofv3 PMC passes of this same command
    (FETCH_SIZE and WRITE_SIZE collected in separate runs; FETCH_SIZE doubled as the gfx950 note of MI355X_MICROARCH.md
    prescribes), from the NEWEST ``profiles/rNN`` that profiled this kernel.  ``(None, None)`` when none did."""
    for rnd in profile_rounds():
        for fname in files:
            path = os.path.join(rnd, fname)
            try:
                hits = [v for name, v in _kernels_of(path).items() if name.startswith(kernel_key)]
                if hits:
                    return hits[0]["hbm_bytes_corrected"], os.path.relpath(path, ROOT)
            except (OSError, KeyError, ValueError):
                pass
    return None, None


# the kernels of ONE image-encoder launch sequence, by generation of K4 (newest first): round 6's resident kernel keeps B in
# LDS; rounds 3-5 ran stem_conv2a + conv2b_conv3 (conv 16->8 inside); both end in the split-K linear layer
K4_SEQUENCES = (("image_encoder_resident_kernel<false", "fc_partial_f16x3_kernel", "fc_tail_kernel<false"),
                ("stem_conv2a_kernel<false", "conv2b_conv3_kernel<false", "fc_partial_f16x3_kernel", "fc_tail_kernel<false"))


def pmc_traffic_k4_ekf():
    """``(HBM bytes, source file)`` of ONE image-encoder launch sequence of the EKF bench (4096 images x 2 encoders), summed
    over its kernels, from the PMC passes of that very command (newest ``profiles/rNN/pmc_hbm_traffic_ekf.json``)."""
    for rnd in profile_rounds():
        path = os.path.join(rnd, "pmc_hbm_traffic_ekf.json")
        try:
            k = _kernels_of(path)
        except (OSError, KeyError, ValueError):
            continue
        for seq in K4_SEQUENCES:
            hits = [[v for name, v in k.items() if name.startswith(prefix)] for prefix in seq]
            if all(hits):
                return sum(h[0]["hbm_bytes_corrected"] for h in hits), os.path.relpath(path, ROOT)
    return None, None


def build_filter(wl, device, seed=0):
    import multimodalfilter_amd as mmf

    torch.manual_seed(seed)
    f = mmf.model_types(wl["task"])[wl["cls"]](**wl.get("ctor", {}))
    f.to(device).eval()
    return f


def make_inputs(wl, steps, batch, seed, device, state_dim):
    from multimodalfilter_amd import synthetic

    traj = synthetic.make_trajectories(state_dim=state_dim, T=steps, N=batch, seed=seed,
                                       image_blackout_ratio=wl.get("blackout", 0.0))
    return traj, to_device(traj, device)


class FilterRun:
    """ONE filter run over ``traj`` (``(T+1, N, ...)`` device tensors, index 0 = the initial time step) cut into
    consecutive ``forward_loop`` segments that share the belief: ``start()`` initialises it the reference's way
    (``states[0]``, ``0.1 I``: ``eval_helpers.py:125-131``) and rewinds the noise, ``steps(t0, t1)`` filters time
    steps ``t0+1 .. t1``.  ``noise``: ``("philox", seed)`` -- counter-based, generated inside the dynamics kernel --
    or the pre-drawn blocks ``(eps0 (N, M, d), eps (T, N, M, d), u (T, N))`` (consumed in place)."""

    def __init__(self, f, traj, noise=None, *, particles=None, traj_offset=0):
        self.f, self.traj, self.noise, self.M, self.traj_offset = f, traj, noise, particles, traj_offset
        self.obs = {k: traj[k] for k in ("image", "gripper_pos", "gripper_sensors")}

    def start(self):
        import multimodalfilter_amd as mmf

        f, states = self.f, self.traj["states"]
        N, d = states.shape[1:]
        if self.M is not None:
            f.num_particles, f.resample_mode = self.M, "systematic"
            f.noise = (mmf.CounterNoise(self.noise[1], traj_offset=self.traj_offset) if self.noise[0] == "philox"
                       else mmf.StackedNoise(*self.noise))
        cov = (torch.eye(d, device=states.device) * 0.1)[None].expand(N, d, d)
        with torch.no_grad():
            f.initialize_beliefs(mean=states[0], covariance=cov)
        return self

    def steps(self, t0, t1):
        with torch.no_grad():
            return self.f.forward_loop(observations={k: v[t0 + 1:t1 + 1] for k, v in self.obs.items()},
                                       controls=self.traj["controls"][t0 + 1:t1 + 1])


def device_noise(T, N, M, d, seed, device):
    """Pre-drawn noise blocks generated ON the device (pre-roll / calibration runs no oracle has to replay)."""
    g = torch.Generator(device=device).manual_seed(seed)
    return (torch.randn((N, M, d), generator=g, device=device), torch.randn((T, N, M, d), generator=g, device=device),
            torch.rand((T, N), generator=g, device=device))


def ess_over_m(loglik):
    """``(T, N, M)`` log-likelihoods of steps that start from uniform weights (every step of a resampling
    filter) -> ``(T, N)`` effective sample size as a fraction of M."""
    w = torch.softmax(loglik.double(), dim=-1)
    return (1.0 / (w * w).sum(-1) / loglik.shape[-1]).float()


def engine_ess(run, segments):
    """The engine's OWN ESS/M on the steps of ``run``: the same segments again, the native loop keeping every
    step's log-likelihoods (``record_indices``; bit-identical to the unrecorded run) -> ``(T, N)`` per segment."""
    f = run.f
    rec, f.record_indices = f.record_indices, True
    out = []
    try:
        run.start()
        for t0, t1 in segments:
            run.steps(t0, t1)
            out.append(ess_over_m(f.last_log_likelihoods).cpu())
    finally:
        f.record_indices = rec
        f.last_log_likelihoods = f.last_resample_indices = f.last_log_weights_in = None
    return out


def ess_summary(ess, lo=0.05, hi=0.5):
    """``(T, N)`` -> the figures the line carries: per-step batch means (min / max / mean over the steps) and the
    share of single (step, trajectory) cells inside SURVEY.md 8d's band."""
    per_step = ess.mean(1)
    return {"band": [lo, hi], "per_step_batch_mean_min": float(per_step.min()), "per_step_batch_mean_max": float(per_step.max()),
            "mean": float(ess.mean()), "steps_with_batch_mean_in_band": int(((per_step >= lo) & (per_step <= hi)).sum()),
            "steps": int(ess.shape[0]), "cells_in_band_fraction": float(((ess >= lo) & (ess <= hi)).float().mean())}


def calibrate_to_band(f, wl, device, d, M, *, target=0.25, accept=(0.2, 0.31), iters=8, batch=64, steps=16, seed=909):
    """Scale the measurement heads until the TRACKING filter holds ESS/M ~ ``target`` (non-degenerate
    weights, ESS/M in [0.05, 0.5]; flat log-likelihoods make resampling the identity and flatter a benchmark,
    peaked ones leave a handful of survivors).  Round 3 calibrated once on the initial belief; the cloud then
    contracted and the weights went flat (ESS/M 0.35 -> 0.9).  Here the filter ITSELF is run -- ``BURN_IN`` steps
    from the initial belief, then ``steps`` more on ``batch`` trajectories -- and a head's scale is moved by the
    log-normal rule ``ESS/M = exp(-var(loglik))`` on the steady-state mean until that mean is inside ``accept``
    (sharper heads narrow the cloud, hence the iteration).  A crossmodal measurement model is a per-trajectory
    mixture of its unimodal heads -- whichever modality the weight model prefers sets that trajectory's weights -- so
    every head is first tuned ALONE (``enabled_models`` one-hot: the reference's own masking, ``base_models/
    crossmodal_pf.py:106-120``) and the mixture last (a common factor): a batch mean inside the band must not be the
    average of flat trajectories and degenerate ones (round 4, measured on the push filter: quantiles 0.04 / 0.72 /
    0.75 when only the mixture was tuned).  Returns the trace ``[(what, factor applied, ESS/M before it)]``."""
    from multimodalfilter_amd import synthetic

    T = BURN_IN + steps
    # (never blacked-out frames here: a head tuned ALONE on a frame whose modality weight is -inf has no weights at all)
    traj = to_device(synthetic.make_trajectories(state_dim=d, T=T, N=batch, seed=seed), device)
    run = FilterRun(f, traj, device_noise(T, batch, M, d, seed + 1, device), particles=M)
    meas = f.measurement_model
    subs = list(getattr(meas, "measurement_models", [meas]))
    heads = [m.shared_layers[4] for m in subs]
    trace = []

    def tune(hs, what):
        for _ in range(iters):
            cur = float(engine_ess(run, [(0, BURN_IN), (BURN_IN, T)])[1].mean())
            if accept[0] <= cur <= accept[1]:
                trace.append((what, 1.0, cur))
                return
            factor = (np.log(1.0 / target) / max(np.log(1.0 / min(cur, 1.0)), 1e-7)) ** 0.5
            factor = float(min(max(factor, 0.25), 150.0))
            with torch.no_grad():
                for h in hs:
                    h.weight.mul_(factor)
                    h.bias.mul_(factor)
            trace.append((what, factor, cur))

    if len(subs) > 1 and hasattr(meas, "enabled_models"):
        everything = list(meas.enabled_models)
        try:
            for k in range(len(subs)):
                meas.enabled_models = [i == k for i in range(len(subs))]
                tune([heads[k]], f"head {k} alone")
        finally:
            meas.enabled_models = everything
    tune(heads, "all heads")
    return trace


def belief_independent_ms(f, traj, repeats=3):
    """Milliseconds of the part of a particle filter's ``forward_loop`` that does not depend on the belief -- the
    image encoders (K4) and the per-trajectory encoders / weight model / hoisted join-layer halves (K7) of all
    ``T*N`` rows, evaluated once ahead of the recursion (``filters.ParticleFilter.forward_loop``) -- so that a loop's
    time splits into `encoders` and `recursion`."""
    obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
    ctrl = traj["controls"][1:]
    T, N = ctrl.shape[:2]
    flat = lambda x: x.reshape((T * N,) + tuple(x.shape[2:]))
    best = None
    with torch.no_grad():
        for _ in range(repeats + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f.measurement_model.encode_observations({k: flat(v) for k, v in obs.items()})
            f.dynamics_model.encode_controls(flat(ctrl))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
    return 1e3 * best


def reference_sized_regimes(device):
    """The sizes the REFERENCE runs (scripts/bench_reference_sizes.py has the CPU twins): evaluation of the door
    crossmodal PF at 32 trajectories x 300 particles (``door_models/pf.py:24-27``, ``eval_helpers.py:125-142``)
    and one end-to-end training step at 32 x 30 particles x 16 steps (``train_door.py:63-71``), forward +
    backward + Adam through the native K6 recursion."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic, train

    d = 3
    out = {}
    # --- evaluation
    N, M, T = 32, 300, 200
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(device).eval()
    synthetic.stabilise_dynamics(f)
    traj = to_device(synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=5), device)
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=6)
    noise = (eps0.to(device), torch.stack(eps).to(device), torch.stack(us).to(device))
    times = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_pf(f, traj, noise, M)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    enc_ms = belief_independent_ms(f, traj)
    from multimodalfilter_amd import _abi, engine as _eng
    out["eval_32x300"] = {"steps": T, "ms_per_step": 1e3 * best / T, "particle_steps_per_s": N * M * T / best,
                          "encoders_ms_per_step": enc_ms / T, "recursion_ms_per_step": (1e3 * best - enc_ms) / T,
                          "recursion": ("ONE persistent launch for all steps (csrc/pf_persistent.inc): role-specialised workgroups, "
                                        "tagged-granule hand-offs through L2" if _eng.PF_PERSISTENT and _abi.pf_persistent_plan(N, M, 2) > 0
                                        else "four launches per step (mmf_pf_forward_loop)"),
                          "note": "ms_per_step = whole forward_loop / steps: image + vector encoders of all T*N frames (K4, K7: encoders_ms_per_step) "
                                  "ahead of the recursion, then the recursion"}
    # --- training
    N, M, L = 32, 30, 16
    ft = mmf.door_models.DoorCrossmodalParticleFilter().to(device).train()
    batch = to_device(synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=11), device)
    cov = torch.eye(d, device=device) * 0.1
    engine.set_training_backend("hip")
    try:
        opt = torch.optim.Adam(ft.parameters(), lr=1e-4)
        ft.noise = mmf.NoiseSource(seed=5)
        times = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            train.train_filter_step(ft, batch, opt, initial_covariance=cov, noise=ft.noise)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    finally:
        engine.set_training_backend(None)
    times = sorted(times[2:])
    out["train_e2e_32x30x16"] = {"ms_per_optimiser_step": 1e3 * times[len(times) // 2],
                                 "backend": "hip (native K6 recursion: mmf_pf_train_forward / _backward)"}
    # the same step captured once as a hipGraph and replayed (train.GraphedFilterStep: same kernels, same bits; the eager
    # step is host-bound at this size), image-encoder training forward through the resident K4 kernel
    engine.set_training_backend("hip")
    engine.set_image_encoder_precision("f16x3")
    try:
        torch.manual_seed(0)
        fg = mmf.door_models.DoorCrossmodalParticleFilter().to(device).train()
        fg.noise = mmf.NoiseSource(seed=5)
        step = train.GraphedFilterStep(fg, torch.optim.Adam(fg.parameters(), lr=1e-4, capturable=True, fused=True), initial_covariance=cov,
                                       noise=fg.noise, eager_steps=2)
        times = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(batch)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        times = sorted(times[4:])
        out["train_e2e_32x30x16"]["hipgraph_replay_ms_per_optimiser_step"] = 1e3 * times[len(times) // 2]
        out["train_e2e_32x30x16"]["hipgraph_note"] = ("train.GraphedFilterStep: forward + backward + Adam(capturable) captured once, replayed per batch; "
                                                       "image-encoder training forward = the resident K4 kernel (f16x3)")
    except Exception as e:  # a capture problem must not take the bench line with it
        out["train_e2e_32x30x16"]["hipgraph_error"] = f"{type(e).__name__}: {e}"[:300]
    finally:
        engine.set_image_encoder_precision(None)
        engine.set_training_backend(None)
    return out


def _free_port():
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n_ranks: int, argv) -> int:
    """``python bench.py --gpus N`` without a launcher around it: this process starts N fresh
    children (one per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, the
    same contract ``torch.distributed.run`` provides) and waits for them.  It never touches the
    GPU itself -- no HIP call, no ``.so`` load -- and nothing is ever re-exec'ed.  Rank 0's JSON
    line passes through on stdout.  Any non-zero child ends the others and becomes the exit code."""
    import subprocess

    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    procs = []
    for r in range(n_ranks):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e))
    rc = 0
    pending = list(procs)
    kill_at = None  # a rank stuck in a collective or a kernel may ignore SIGTERM: SIGKILL after a grace period
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:  # exact PIDs of our own children only
                    q.terminate()
                kill_at = time.monotonic() + 10.0
        if kill_at is not None and time.monotonic() > kill_at:
            for q in pending:
                q.kill()
            kill_at = time.monotonic() + 10.0
        time.sleep(0.05)
    return rc


def dry_run(args):
    """``MMF_BENCH_DRY=1``: the N-rank plumbing without the GPU work (CPU test of the launcher):
    rendezvous, the same shard bounds / all-gather / max-over-ranks the real run uses, one JSON line from rank 0."""
    from multimodalfilter_amd import distributed

    rank, world, local = distributed.init_from_env()
    if os.environ.get("MMF_BENCH_DRY") == f"fail{rank}":
        raise SystemExit(3)
    rows = distributed.all_gather_rows(torch.full((rank + 1, 2), float(rank)))
    slowest = distributed.max_over_ranks(float(rank), torch.device("cpu"))
    wl = dict(WORKLOADS[args.workload])
    out = {"dry_run": True, "n_gpus": world, "world_size_seen": world, "gathered_rows": int(rows.shape[0]),
           "max_over_ranks": slowest, "steps": args.steps, "warmup": args.warmup, "workload": args.workload}
    if args.global_batch:  # strong scaling (BASELINE config 4): every rank's shard of the global batch
        lo, hi = distributed.shard_bounds(args.global_batch, rank, world)
        spans = distributed.all_gather_rows(torch.tensor([[float(rank), float(lo), float(hi)]]))
        out["shards"] = [[int(r), int(a), int(b)] for r, a, b in spans.tolist()]
        out["scaling"], out["global_batch"] = "strong", args.global_batch
    else:
        out["scaling"], out["global_batch"] = "weak", (args.batch or wl["batch"]) * world
    if wl["kind"] == "train":
        # the collective of the training leg on a stand-in with the filter's parameter count (X2: 696,993 fp32), rank-
        # dependent gradients: every rank must end with the average, bit for bit the same
        class _Params(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w = torch.nn.Parameter(torch.zeros(696993))
        m = _Params()
        m.w.grad = torch.full_like(m.w, float(rank + 1))
        t0 = time.perf_counter()
        n = distributed.all_reduce_gradients(m)
        out["allreduce_ms"] = 1e3 * (time.perf_counter() - t0)
        out["allreduce_elements"] = n
        want = sum(range(1, world + 1)) / world
        ok = torch.tensor([[float(bool(torch.all(m.w.grad == want)))]])
        out["allreduce_average_exact_on_every_rank"] = bool(distributed.all_gather_rows(ok).min() == 1.0)
    distributed.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def main_train(args, wl):
    """``--workload push_train`` (BASELINE config 5, SURVEY.md 8e "Collective (training, C5)"): K timed optimiser steps of
    the push unimodal particle filter's end-to-end training (train mode: no resampling; bf16 image-encoder forward;
    native K6 recursion), per-rank batch ``N x M x L``, rank-private data, replicated weights, ONE flat all-reduce of
    the gradients per step inside the timed region (``distributed.all_reduce_gradients``: RCCL over xGMI)."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, distributed, engine, synthetic, train

    torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
    _abi.load()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    t_start = time.perf_counter()
    rank, world, local = distributed.init_from_env()
    local_dev = local % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    d = 2
    N, M, L = args.batch or wl["batch"], args.particles or wl["particles"], wl["length"]
    K, W = args.steps, max(1, args.warmup)
    engine.set_training_backend("hip")
    engine.set_image_encoder_precision("bf16")
    torch.manual_seed(0)  # the same initial weights on every rank
    f = mmf.push_models.PushUnimodalParticleFilter().to(device).train()
    f.num_particles = M
    batch = to_device(synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=11 + 1000 * rank), device)
    cov = torch.eye(d, device=device) * 0.1
    opt = torch.optim.SGD(f.parameters(), lr=1e-4)
    f.noise = mmf.NoiseSource(seed=5 + rank)
    # the collective's share of a step, from events on the stream around it
    marks = []
    real = distributed.all_reduce_gradients

    def timed_all_reduce(module, average=True):
        return real(module, average, marks)   # events (start, packed, reduced, end) on the stream around the exchange

    distributed.all_reduce_gradients = timed_all_reduce
    try:
        for _ in range(W):
            train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=f.noise, all_reduce=world > 1)
        marks.clear()
        torch.cuda.reset_peak_memory_stats()
        distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = [train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=f.noise, all_reduce=world > 1) for _ in range(K)]
        torch.cuda.synchronize()
        distributed.barrier()
        dt = distributed.max_over_ranks(time.perf_counter() - t0, device)
    finally:
        distributed.all_reduce_gradients = real
        engine.set_training_backend(None)
        engine.set_image_encoder_precision(None)
    # replicated weights must still be replicated: every rank's parameter checksum
    chk = torch.stack([p.detach().double().sum() for p in f.parameters()]).sum().reshape(1, 1).float()
    sums = distributed.all_gather_rows(chk)
    if rank != 0:
        return
    ar_ms = [m[1].elapsed_time(m[2]) for m in marks]      # the collective alone (gloo: incl. its host round trip)
    ex_ms = [m[0].elapsed_time(m[3]) for m in marks]      # pack kernel + collective + scale kernel; no copy back
    out = {"metric": "training particle-steps/sec (batch x particles x subsequence steps, forward + backward + optimiser)",
           "value": world * N * M * (L - 1) * K / dt, "unit": "particle-steps/s (forward + backward)",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None,
           "dtype": "f32 via f16x3 (per-particle networks, forward and backward); bf16 image-encoder forward (BASELINE config 5)",
           "data": "synthetic",
           "config": {"workload": f"{wl['desc']}, batch {N} x {M} particles x {L}-step subsequences per GPU x {world} GPU(s)",
                      "filter": wl["cls"], "batch_per_gpu": N, "particles": M, "subsequence_length": L, "global_batch": N * world,
                      "world_size_seen": world, "parallelism": f"data-parallel x{world}: replicated weights, one flat gradient all-reduce per step"},
           "allreduce_ms": (sum(ar_ms) / len(ar_ms)) if ar_ms else 0.0,
           "gradient_exchange_ms": (sum(ex_ms) / len(ex_ms)) if ex_ms else 0.0,
           "gradient_exchange": "torch.cat into one flat buffer, ONE in-place all-reduce, one scale kernel, p.grad re-pointed at views (no copy back)",
           "allreduce_elements": sum(p.numel() for p in f.parameters() if p.requires_grad),
           "weights_identical_across_ranks": bool((sums == sums[0]).all()),
           "loss_first_last": [losses[0], losses[-1]],
           "peak_memory_GB": torch.cuda.max_memory_allocated() / 2 ** 30,
           "training_recursion": "fused" if engine.TRAIN_FUSED else "three-pass",
           "roofline": None, "cpu_baseline": None,
           "bench_seconds": round(time.perf_counter() - t_start, 1)}
    print(json.dumps(out), flush=True)


def leg_pf(name, wl, *, K, W, device, share_state=None, seed=7000):
    """A particle-filter configuration as one bounded leg: burn-in + W warm-up + K timed steps of one tracking filter
    (counter-based process noise: no ``(T, N, M, d)`` tensor has to be drawn on the host), run twice -- the first run
    pays the one-time host costs of its shapes -- and timed on the second."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, evaluation, synthetic

    t_leg = time.perf_counter()
    spec = mmf.door_models._ns.task if wl["task"] == "door" else mmf.push_models._ns.task
    d, B, M = spec.state_dim, wl["batch"], wl["particles"]
    f = build_filter(wl, device)
    synthetic.stabilise_dynamics(f)
    f.num_particles = M
    cal = None
    if share_state is not None:
        f.load_state_dict(share_state)  # the headline filter's calibrated weights (same architecture)
    else:
        cal = calibrate_to_band(f, wl, device, d, M, batch=min(B, 256))
    T = BURN_IN + W + K
    _, traj = make_inputs(wl, T, B, seed, device, d)
    run = FilterRun(f, traj, ("philox", seed + 1), particles=M)
    f.reserve(steps=K, batch=B, particles=M)
    segs = [(0, BURN_IN), (BURN_IN, BURN_IN + W), (BURN_IN + W, T)]

    def once(timer):
        run.start()
        run.steps(*segs[0])
        if W > 0:
            run.steps(*segs[1])
        engine.set_kernel_timer(timer)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred = run.steps(*segs[2])
        mse = evaluation.per_trajectory_mse(pred, traj["states"][segs[2][0] + 1:T + 1], start=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        engine.set_kernel_timer(None)
        return dt, mse

    once(None)
    timer = engine.KernelTimer(loop_stride=K if K < 64 else (K + 2) // 3)
    dt, mse = once(timer)
    out = {"workload": workload_desc(wl, B, M, B, 1, "weak"), "filter": wl["cls"], "steps": K, "warmup": W, "burn_in": BURN_IN,
           "value": B * M * K / dt, "unit": "particle-steps/s", "ms_per_step": 1e3 * dt / K, "process_noise": "philox",
           "posterior_rmse_vs_truth": [float(x) for x in evaluation.raw_rmse(mse)]}
    ks = timer.summary()
    dom = ks.get("particle_net_measure")
    if dom:
        ach = dom["flops_per_launch"] / (dom["avg_ms"] * 1e-3) / 1e12
        prec = engine.DEFAULT_PRECISION
        out["roofline"] = {"kernel": "particle_net_kernel (measurement network)", "bound": "mfma", "achieved": ach,
                           "peak": MFMA_PEAK[prec], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK[prec], "avg_us": 1e3 * dom["avg_ms"]}
    k1 = ks.get("pf_reweight_resample")
    if k1:
        gbs = k1["bytes_per_launch"] / (k1["avg_ms"] * 1e-3) / 1e9
        out["roofline_k1"] = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                              "avg_us": 1e3 * k1["avg_ms"]}
    # the engine's own weights on the timed steps of (up to) its first 32 trajectories: counter-based noise keeps
    # a trajectory's draws the same whatever batch it sits in
    nb = min(B, 32)
    sub = FilterRun(f, {k: v[:, :nb].contiguous() for k, v in traj.items()}, ("philox", seed + 1), particles=M)
    out["ess_over_m"] = ess_summary(engine_ess(sub, segs)[2])
    if cal is not None:
        out["head_calibration_trace"] = [[w, round(a, 3), round(b, 4)] for w, a, b in cal]
    out["parity"] = _leg_parity_pf(wl, f, d, M, device)
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    del run, sub, traj
    return out, f


def leg_ekf(name, wl, *, K, W, device, seed=7100, cpu_batch=256, cpu_steps=6, plain_ms=None):
    """An EKF configuration as one bounded leg (config 4's per-GPU share; its blackout twin): W warm-up + K timed
    steps of one filter run, image encoders inside the timed region, with the K4 launch sequence's roofline and a
    bounded CPU-oracle sample for the baseline and the parity number."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, evaluation, synthetic

    t_leg = time.perf_counter()
    d, B = 3, wl["batch"]
    f = build_filter(wl, device)
    synthetic.stabilise_dynamics(f)
    T = W + K
    _, traj = make_inputs(wl, T, B, seed, device, d)
    run = FilterRun(f, traj)

    def once(timer):
        run.start()
        if W > 0:
            run.steps(0, W)
        engine.set_kernel_timer(timer)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pred = run.steps(W, T)
        mse = evaluation.per_trajectory_mse(pred, traj["states"][W + 1:T + 1], start=0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        engine.set_kernel_timer(None)
        return dt, mse

    once(None)
    timer = engine.KernelTimer()
    dt, mse = once(timer)
    out = {"workload": workload_desc(wl, B, 1, B, 1, "weak"), "filter": wl["cls"], "steps": K, "warmup": W,
           "value": B * K / dt, "unit": "trajectory-steps/s", "ms_per_step": 1e3 * dt / K,
           "posterior_rmse_vs_truth": [float(x) for x in evaluation.raw_rmse(mse)]}
    if wl.get("blackout", 0.0) > 0:
        dark = (traj["image"][W + 1:T + 1].abs().sum((-1, -2)) < 1e-8)
        out["blacked_out_frames_fraction"] = float(dark.float().mean())
        out["native_loop"] = "mmf_ekf_forward_loop; the batch-global branch of door_models/crossmodal_kf.py:59-62 is a per-step device flag"
        if plain_ms:
            out["ms_per_step_over_plain_ekf"] = out["ms_per_step"] / plain_ms
    dom = timer.summary().get("image_encoder")
    if dom:
        ach = dom["flops_per_launch"] / (dom["avg_ms"] * 1e-3) / 1e12
        traffic, source = pmc_traffic_k4_ekf() if (B == 1024 and not wl.get("blackout")) else (None, None)
        out["roofline"] = {"kernel": "image encoder launch sequence (K4) per chunk of images", "bound": "mfma", "achieved": ach,
                           "peak": MFMA_PEAK["f16x3"], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK["f16x3"], "avg_us": 1e3 * dom["avg_ms"],
                           "traffic": traffic, "traffic_source": source}
    cores = min(CPU_THREADS, os.cpu_count() or 1)
    base, parity = cpu_baseline_ekf(wl, f, d, cores, sample_batch=cpu_batch, sample_steps=cpu_steps)
    out["cpu_baseline"] = base
    out["parity"] = {"max_rel_err_posterior_mean_vs_oracle": parity["max_rel_err_posterior_mean"],
                     "max_rel_err_posterior_covariance_vs_oracle": parity["max_rel_err_posterior_covariance"],
                     "max_rel_err_posterior_covariance_per_matrix_vs_oracle": parity["max_rel_err_posterior_covariance_per_matrix"],
                     "max_rel_err_posterior_mean_per_vector_vs_oracle": parity["max_rel_err_posterior_mean_per_vector"],
                     "sample": f"{cpu_batch} trajectories x {cpu_steps + 1} steps, same weights / inputs; CPU oracle"}
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    del run, traj
    return out


def leg_train(device, *, iters=4):
    """BASELINE config 5's per-GPU shape: push unimodal particle filter, 8192 particles, batch 32 (N x M = 2^18),
    subsequences of 16 steps, train mode (no resampling), bf16 image-encoder forward, forward + backward + SGD
    through the native K6 recursion (``mmf_pf_train_forward`` / ``_backward``).  Parity: the loss and the gradients of
    a 4 x 30 x 3 twin of the same step against the CPU oracle's autograd."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic, train
    from oracle import models as om
    from oracle.tf.base import ReplayNoise as OReplay

    t_leg = time.perf_counter()
    d, N, M, L = 2, 32, 8192, 16
    out = {"workload": f"push unimodal particle filter, train mode, {N} x {M} particles x {L}-step subsequences, forward + backward + SGD",
           "filter": "PushUnimodalParticleFilter", "unit": "particle-steps/s (forward + backward)"}
    engine.set_training_backend("hip")
    engine.set_image_encoder_precision("bf16")
    try:
        torch.manual_seed(0)
        f = mmf.push_models.PushUnimodalParticleFilter().to(device).train()
        f.num_particles = M
        batch = to_device(synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=11), device)
        cov = torch.eye(d, device=device) * 0.1
        opt = torch.optim.SGD(f.parameters(), lr=1e-4)
        f.noise = mmf.NoiseSource(seed=5)
        times = []
        torch.cuda.reset_peak_memory_stats()
        held = torch.cuda.memory_allocated()  # the bench's own resident tensors are not the step's
        for _ in range(iters + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=f.noise)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        best = min(times[1:])
        out.update({"ms_per_step": 1e3 * best, "value": N * M * (L - 1) / best, "peak_memory_GB": (torch.cuda.max_memory_allocated() - held) / 2 ** 30,
                    "image_encoder_forward": "bf16 (BASELINE config 5)", "backend": "hip (K6)"})
        del f, opt, batch
        # parity twin (exact-fp32 image encoders: the oracle's arithmetic)
        engine.set_image_encoder_precision(None)
        n, m, T = 4, 30, 3
        torch.manual_seed(1)
        ft = mmf.push_models.PushUnimodalParticleFilter().to(device).train()
        ft.num_particles = m
        tw = synthetic.make_trajectories(state_dim=d, T=T, N=n, seed=12)
        g = torch.Generator().manual_seed(13)
        eps0 = torch.randn((n, m, d), generator=g)
        eps = [torch.randn((n, m, d), generator=g) for _ in range(T)]
        o = om.build("PushUnimodalParticleFilter")
        o.load_state_dict({k: v.detach().cpu() for k, v in ft.state_dict().items()})
        o.train()
        o.num_particles = m
        covc = (torch.eye(d) * 0.1)[None].expand(n, d, d)
        obs = synthetic.observations_of(tw)

        def loss_of(filt, dev):
            mv = lambda t: t.to(dev)
            filt.initialize_beliefs(mean=mv(tw["states"][0]), covariance=mv(covc))
            pred = filt.forward_loop(observations={k: mv(v[1:]) for k, v in obs.items()}, controls=mv(tw["controls"][1:]))
            return torch.mean((pred - mv(tw["states"][1:])) ** 2)

        o.noise = OReplay([eps0] + eps, [])
        lo = loss_of(o, "cpu")
        lo.backward()
        ft.noise = mmf.ReplayNoise([eps0.to(device)] + [e.to(device) for e in eps], [])
        le = loss_of(ft, device)
        le.backward()
        worst = 0.0
        po = dict(o.named_parameters())
        for k, p in ft.named_parameters():
            if p.grad is None or po[k].grad is None:
                continue
            worst = max(worst, float((p.grad.cpu() - po[k].grad).abs().max() / po[k].grad.abs().max().clamp_min(1e-12)))
        out["parity"] = {"loss_rel_diff_vs_oracle": abs(float(le) - float(lo)) / max(abs(float(lo)), 1e-12),
                         "max_grad_diff_over_tensor_max_vs_oracle": worst,
                         "sample": f"{n} x {m} particles x {T} steps twin of the step, same weights / inputs / noise; CPU oracle autograd"}
        del ft, o
    finally:
        engine.set_training_backend(None)
        engine.set_image_encoder_precision(None)
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    torch.cuda.empty_cache()
    return out


def run_configs(args, device, headline_filter, d, K, W):
    """BASELINE.json's other configurations and the blackout workloads, each a bounded leg (one GPU)."""
    legs = {}
    Kl = min(K, 64)  # legs are bounded: at most 64 timed steps each

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            legs[name] = fn()
        except Exception as e:  # a failing leg must not take the headline line with it
            import traceback
            legs[name] = {"error": f"{type(e).__name__}: {e}", "where": traceback.format_exc().splitlines()[-3:]}
        torch.cuda.empty_cache()
        return time.perf_counter() - t0

    share = None
    if args.workload == "door_pf":
        share = {k: v.detach().clone() for k, v in headline_filter.state_dict().items()}
    # BASELINE config 1 is the reference's own CPU-runnable case (door unimodal EKF, 32 trajectories): here on the GPU, its
    # recursion as ONE persistent launch (csrc/ekf_persistent.inc), with the CPU oracle timed at the same batch beside it
    c1 = dict(task="door", cls="DoorUnimodalKalmanFilter", kind="ekf", batch=32, particles=1, desc="door unimodal EKF")
    guarded("C1_door_unimodal_ekf_32", lambda: leg_ekf("C1", c1, K=4 * Kl, W=W, device=device, cpu_batch=32, cpu_steps=200))
    c2 = dict(WORKLOADS["door_pf"], batch=256, particles=1024)
    guarded("C2_door_crossmodal_pf_256x1024", lambda: leg_pf("C2", c2, K=Kl, W=W, device=device, share_state=share)[0])
    c3 = dict(WORKLOADS["push_pf"], batch=1024, particles=4096)
    guarded("C3_push_crossmodal_pf_1024x4096", lambda: leg_pf("C3", c3, K=min(Kl, 32), W=min(W, 8), device=device)[0])
    c4 = dict(WORKLOADS["door_ekf"], batch=1024)
    guarded("C4_door_crossmodal_ekf_1024_per_gpu_share_of_8192", lambda: leg_ekf("C4", c4, K=Kl, W=W, device=device))
    guarded("C5_push_unimodal_pf_train_32x8192x16", lambda: leg_train(device))
    bp = dict(WORKLOADS["door_pf_blackout"])
    guarded("blackout_0.4_door_crossmodal_pf_256x4096", lambda: leg_pf("blackout_pf", bp, K=Kl, W=W, device=device, share_state=share)[0])
    plain = legs.get("C4_door_crossmodal_ekf_1024_per_gpu_share_of_8192", {}).get("ms_per_step")
    be = dict(WORKLOADS["door_ekf_blackout"])
    guarded("blackout_0.4_door_crossmodal_ekf_1024", lambda: leg_ekf("blackout_ekf", be, K=Kl, W=W, device=device, plain_ms=plain))
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="door_pf", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU")
    ap.add_argument("--particles", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--precision", default=None, choices=["f32", "f16x3"],
                    help="arithmetic of the per-particle 64x64 layers (default: engine default)")
    ap.add_argument("--no-precision-study", action="store_true",
                    help="skip the per-network error measurement against fp64")
    ap.add_argument("--no-f32-mode", action="store_true",
                    help="skip the extra timed pass in exact-f32 mode")
    ap.add_argument("--noise", default="tensor", choices=["tensor", "philox"],
                    help="process noise of the timed particle-filter passes: pre-drawn (T, N, M, d) tensor, or "
                         "counter-based, generated inside the dynamics kernel")
    ap.add_argument("--no-reference-sizes", action="store_true",
                    help="skip the two extra lines at the sizes the reference itself runs (32 x 300 eval, 32 x 30 x 16 training)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` legs (BASELINE.json's other configurations, the blackout workloads)")
    ap.add_argument("--no-calibration", action="store_true",
                    help="keep the randomly initialised measurement heads (flat weights; A/B only)")
    ap.add_argument("--preroll-seconds", type=float, default=0.5,
                    help="untimed repetitions of the whole sequence on OTHER inputs of the same shapes before the run that is timed (GPU clock ramp)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="STRONG scaling: this many trajectories in total, sharded over the ranks "
                         "(BASELINE config 4: --workload door_ekf --global-batch 8192)")
    args = ap.parse_args()

    # N > 1 without a launcher: become the launcher (before the HIP library or the GPU is touched)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("MMF_BENCH_DRY"):
        return dry_run(args)

    if WORKLOADS[args.workload]["kind"] == "train":
        if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
        return main_train(args, dict(WORKLOADS[args.workload]))

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, distributed, engine, evaluation, synthetic

    # The host has hundreds of hardware threads; torch's default intra-op pool (one thread per
    # core) leaves that many OpenMP workers spin-waiting after every CPU tensor op (synthetic
    # input generation), which starves the launch thread for the next ~0.1 s.
    torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
    _abi.load()  # fail loudly before touching the GPU if the HIP library is missing
    if args.precision:
        engine.set_default_precision(args.precision)
    precision = engine.DEFAULT_PRECISION
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    t_start = time.perf_counter()
    rank, world, local = distributed.init_from_env()
    local_dev = local % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["batch"] = args.batch
    if args.particles:
        wl["particles"] = args.particles
    scaling = "weak"
    if args.global_batch:
        lo, hi = distributed.shard_bounds(args.global_batch, rank, world)
        wl["batch"], scaling = hi - lo, "strong"
    K, W, B, M = args.steps, args.warmup, wl["batch"], wl["particles"]
    rows_total = args.global_batch if args.global_batch else B * world   # what the evaluation all-gather assembles
    spec = mmf.door_models._ns.task if wl["task"] == "door" else mmf.push_models._ns.task
    d = spec.state_dim
    pf = wl["kind"] == "pf"
    burn = BURN_IN if pf else 0
    T_all = burn + W + K
    segs = [(0, burn), (burn, burn + W), (burn + W, T_all)]
    # the evaluation statistic skips the first 30 steps of a sequence (eval_helpers.py:149-157)
    mse_start = min(max(0, 30 - (burn + W)), K // 2)

    f = build_filter(wl, device)
    # untrained dynamics are expanding maps; keep any --steps the driver asks for finite
    # (same arithmetic per step; see synthetic.stabilise_dynamics)
    synthetic.stabilise_dynamics(f)
    calibration = None
    if pf:
        f.num_particles = M
        if not args.no_calibration:
            # same weights on every rank: the calibration trajectory and noise do not depend on the rank.  (The
            # calibration batch has the workload's own shape where that is affordable: a rocprofv3 --stats average of
            # this command then averages launches of ONE size.)
            calibration = calibrate_to_band(f, wl, device, d, M, batch=min(B, 256))
    # rank-private trajectories (weak scaling): seed 20201025 + config id + rank.  `traj` is the run that is timed,
    # `traj_r` a second one of the same shapes for the pre-roll (never the timed inputs themselves)
    traj_cpu, traj = make_inputs(wl, T_all, B, 20201025 + 1000 * rank + 2, device, d)
    _, traj_r = make_inputs(wl, T_all, B, 20201025 + 1000 * rank + 1, device, d)
    if pf:
        if args.noise == "philox":
            # counter-based noise generated inside the dynamics kernel (include/mmf_philox.h): no (T, N, M, d)
            # tensor exists; trajectories keep their global index, so any sharding draws the same numbers
            noise, noise_r = ("philox", 78), ("philox", 77)
        else:
            nz = synthetic.draw_filter_noise(T=T_all, N=B, M=M, state_dim=d, seed=78 + rank)
            noise = (nz[0].to(device), torch.stack(nz[1]).to(device), torch.stack(nz[2]).to(device))
            noise_r = device_noise(T_all, B, M, d, 77 + rank, device)
        f.reserve(steps=K, batch=B, particles=M)  # memory planned before the warm-up
        run = FilterRun(f, traj, noise, particles=M, traj_offset=rank * B)
        run_r = FilterRun(f, traj_r, noise_r, particles=M, traj_offset=rank * B)
    else:
        run, run_r = FilterRun(f, traj), FilterRun(f, traj_r)

    def timed_pass():
        """Pre-roll on OTHER inputs, then ONE filter run: (burn-in,) W untimed warm-up steps, then exactly K timed
        steps; returns (seconds, timer, mse, prediction)."""
        # Round 3, measured (profiles/r03/bench_pass_timecourse.txt): after an idle stretch the chip takes ~30 ms of
        # load to get back to its clocks, and only the FIRST pass of a process over a new loop length pays one-time
        # host work (code objects of the torch kernels of the K-step statistic, the allocator's first blocks of those
        # sizes, the first event records).  Round 4: the pre-roll repeats the whole sequence -- burn-in, warm-up, a
        # K-step loop with its statistic -- on a SECOND trajectory / noise set of the same shapes (round 3 rehearsed
        # the timed inputs themselves, which also left them in the Infinity Cache); the run that is timed follows
        # back to back, and the GPU only idles at the synchronisation points the contract prescribes.
        def make_timer():  # its event pool costs ~10 ms of an idle GPU
            return None if args.no_kernel_timers else engine.KernelTimer(loop_stride=K if K < 64 else (K + 2) // 3)  # 3 sampled steps (1 for short passes)

        def sequence(r, tr, timer, sync_clock):
            r.start()
            if burn:
                r.steps(*segs[0])
            if W > 0:  # the W warm-up steps cover the whole path, including the evaluation statistic
                pred_w = r.steps(*segs[1])
                distributed.all_gather_rows(evaluation.per_trajectory_mse(pred_w, tr["states"][segs[1][0] + 1:segs[1][1] + 1], start=0), rows_total)
            engine.set_kernel_timer(timer)
            if sync_clock:
                distributed.barrier()
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred = r.steps(*segs[2])
            mse_local = evaluation.per_trajectory_mse(pred, tr["states"][segs[2][0] + 1:T_all + 1], start=mse_start)
            mse_all = distributed.all_gather_rows(mse_local, rows_total)  # RCCL all-gather of per-sequence errors: ONE collective when the shards are equal
            if sync_clock:
                torch.cuda.synchronize()
                distributed.barrier()
            dt = time.perf_counter() - t0
            engine.set_kernel_timer(None)
            return (distributed.max_over_ranks(dt, device) if sync_clock else dt), timer, mse_all, pred

        rehearsal_timer, timer = make_timer(), make_timer()
        if args.preroll_seconds > 0:
            t_pre = time.perf_counter()
            first = True
            while first or time.perf_counter() - t_pre < args.preroll_seconds:
                sequence(run_r, traj_r, rehearsal_timer if first else None, False)  # back to back, no synchronisation
                first = False
        return sequence(run, traj, timer, True)

    # Order of the GPU work (one GPU): the headline pass, the comparison pass in exact-f32 mode, the error study
    # against fp64, one second of idle, and the headline pass AGAIN -- each pass does its own pre-roll, burn-in and W
    # warm-up steps and times exactly K.  `value` is the LAST pass; the first is reported beside it as `pass_order`.
    # With more than one rank the extras are skipped and the single pass is the value.
    lean = world > 1
    first = None
    if not lean and (pf or not args.no_precision_study):
        first = timed_pass()[0]
    f32_pass = None
    if pf and precision != "f32" and not args.no_f32_mode and not lean:
        engine.set_default_precision("f32")
        f32_pass = timed_pass()
        engine.set_default_precision(precision)
    # arithmetic error of each mode against fp64, at the benchmark's size (rank 0, no collective)
    study = None
    if rank == 0 and not args.no_precision_study and not lean:
        if pf:
            study = precision_errors(wl, f, traj, B, M, raw_dynamics=build_filter(wl, device).dynamics_model)
        else:
            study = image_encoder_precision_errors(wl, f, traj)
            # on an un-stabilised twin: the bench scales the dynamics heads by 2e-3, which hides the
            # networks' error behind the rounding of x + tiny
            study["jacobians"] = jacobian_precision_errors(wl, build_filter(wl, device), traj)
    if first is not None:
        torch.cuda.synchronize()
        time.sleep(1.0)
    distributed.barrier()
    elapsed, timer, mse_all, pred_main = timed_pass()

    total_batch = args.global_batch if args.global_batch else B * world
    units_per_step = total_batch * M if pf else total_batch
    value = units_per_step * K / elapsed
    rmse = evaluation.raw_rmse(mse_all)

    if rank != 0:
        return
    out = {
        "metric": "filter steps/sec (batch x particles)" if pf else "filter steps/sec (trajectories)",
        "value": value,
        "unit": "particle-steps/s" if pf else "trajectory-steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f32" if precision == "f32" else
                 ("f32 via f16x3 (operands split into 2 f16 halves exact to 2^-22, 3 f16 MFMA products "
                  "per product, f32 accumulate)" if pf else
                  "f32; image encoders and dynamics Jacobians f32 via f16x3 (operands split into 2 f16 halves exact to "
                  "2^-22, 3 f16 MFMA products per product, f32 accumulate)"),
        "data": "synthetic",
        "config": {"workload": workload_desc(wl, B, M, total_batch, world, scaling), "filter": wl["cls"],
                   "batch_per_gpu": B, "particles": M,
                   "global_batch": total_batch, "state_dim": d, "resample": "systematic",
                   "world_size_seen": world,
                   "parallelism": f"trajectory-sharded x{world}"},
        "posterior_rmse_vs_truth": [float(x) for x in rmse],
        "sequence": {"burn_in_steps": burn, "warmup_steps": W, "timed_steps": K,
                     "note": "ONE filter run per pass: belief initialised at states[0] / 0.1 I (eval_helpers.py:125-131), burn-in, "
                             "W warm-up steps, then exactly K timed steps, the belief carried across the forward_loop calls"},
        "preroll_seconds": args.preroll_seconds, "process_noise": args.noise if pf else None,
        "pass_order": None if first is None else {
            "headline_opening_the_process_ms_per_step": 1e3 * first / K,
            "headline_after_f32_pass_fp64_study_and_1s_idle_ms_per_step": 1e3 * elapsed / K,
            "note": "value / ms_per_step are the LAST pass (pre-roll on other inputs, burn-in, W warm-up steps, then exactly K timed)"},
        "traffic_method": "every roofline object names the committed file its `traffic` was read from (`traffic_source`: the newest "
                          "profiles/rNN that profiled the kernel); rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this "
                          "command, bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (gfx950 correction)",
    }
    if calibration is not None:
        out["head_calibration_trace"] = [[w, round(a, 3), round(b, 4)] for w, a, b in calibration]

    def k2_roofline(ks, prec):
        dom = ks["particle_net_measure"]
        ach = dom["flops_per_launch"] / (dom["avg_ms"] * 1e-3) / 1e12
        default_shape = (args.workload == "door_pf" and B == 256 and M == 4096)
        traffic, source = pmc_traffic(k2_kernel_key(d, prec)) if default_shape else (None, None)
        r = {"kernel": k2_kernel_name(d, prec) + " (measurement network)", "bound": "mfma", "achieved": ach,
             "peak": MFMA_PEAK[prec], "unit": "TFLOP/s", "frac": ach / MFMA_PEAK[prec],
             "traffic": traffic, "traffic_source": source}
        if prec == "f16x3":
            r["note"] = ("achieved counts ALGORITHMIC fp32 FLOPs; the kernel executes 3 f16 MFMA "
                         "products per algorithmic product (executed-MFMA fraction = 3 x frac)")
        return r

    if timer is not None:
        ks = timer.summary()
        out["kernels"] = {k: {kk: (round(vv, 6) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                          for k, v in ks.items()}
        if pf and "particle_net_measure" in ks:
            default_shape = (args.workload == "door_pf" and B == 256 and M == 4096)
            out["roofline"] = k2_roofline(ks, precision)
            k1 = ks.get("pf_reweight_resample")
            if k1:
                gbs = k1["bytes_per_launch"] / (k1["avg_ms"] * 1e-3) / 1e9
                traffic, source = pmc_traffic("pf_resample_systematic_kernel") if default_shape else (None, None)
                out["roofline_k1"] = {"kernel": f"pf_resample_systematic_kernel<{d}, true>", "bound": "hbm",
                                      "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": source}
        if not pf and "image_encoder" in ks:
            # EKF steps are > 99 % image-encoder MACs: the K4 launch sequence
            # (stem + four 3x3 convolutions + linear tail) is the dominant "kernel"
            dom = ks["image_encoder"]
            ach = dom["flops_per_launch"] / (dom["avg_ms"] * 1e-3) / 1e12
            traffic, source = pmc_traffic_k4_ekf() if (B == 1024 and args.workload == "door_ekf") else (None, None)
            out["roofline"] = {"kernel": "image encoder launch sequence (K4: convolution stack, fc_partial_f16x3_kernel, "
                                         "fc_tail_kernel) per chunk of images",
                               "bound": "mfma", "achieved": ach, "peak": MFMA_PEAK["f16x3"], "unit": "TFLOP/s",
                               "frac": ach / MFMA_PEAK["f16x3"], "traffic": traffic, "traffic_source": source,
                               "note": "ALGORITHMIC fp32 FLOPs (26.12 MMAC per image per encoder); every product is 3 f16 MFMA "
                                       "products (executed-MFMA fraction = 3 x frac); traffic = one launch sequence over 4096 "
                                       "images x 2 encoders (the image virtual sensor's and the weight model's), PMC passes of this command"}
    if "roofline" not in out:
        out["roofline"] = None

    if pf:
        # the ENGINE's own weights over the timed steps: the same run again, every step's log-likelihoods kept
        # (bit-identical: same kernels, same inputs), on the first 64 trajectories' worth of memory at a time
        nb = B if 4.0 * B * M * T_all <= 3e9 else min(B, 64)  # the whole batch where its log-likelihood record fits 3 GB
        if args.noise == "philox":
            sub_noise = noise
        else:
            sub_noise = (noise[0][:nb].contiguous(), noise[1][:, :nb].contiguous(), noise[2][:, :nb].contiguous())
        sub = FilterRun(f, {k: v[:, :nb].contiguous() for k, v in traj.items()}, sub_noise, particles=M, traj_offset=rank * B)
        ess = engine_ess(sub, segs)
        out["ess_over_m"] = dict(ess_summary(ess[2]), trajectories=nb,
                                 per_step_batch_mean=[round(float(x), 4) for x in ess[2].mean(1)][:160],
                                 burn_in_per_step_batch_mean=[round(float(x), 4) for x in ess[0].mean(1)],
                                 note="engine's own normalised weights on the timed steps (SURVEY.md 8d asks for ESS/M in [0.05, 0.5])")
        del sub

    if f32_pass is not None:
        e32, t32, mse32, pred32 = f32_pass
        out["f32_mode"] = {"value": units_per_step * K / e32, "unit": out["unit"],
                           "ms_per_step": 1e3 * e32 / K, "dtype": "f32",
                           "posterior_rmse_vs_truth": [float(x) for x in evaluation.raw_rmse(mse32)]}
        if t32 is not None:
            out["f32_mode"]["roofline"] = k2_roofline(t32.summary(), "f32")
        # the two modes over the same free-running steps (same inputs, same noise): posterior
        # RMSE of each against the truth, and how far single estimates drift apart
        r16, r32 = rmse, evaluation.raw_rmse(mse32)
        out["mode_drift"] = {
            "steps": K, "rmse_f16x3": [float(x) for x in r16], "rmse_f32": [float(x) for x in r32],
            "rmse_rel_diff": float(max(abs(a - b) / b for a, b in zip(r16, r32))),
            "max_abs_diff_posterior_mean_first_timed_step": float((pred_main[0] - pred32[0]).abs().max()),
            "max_abs_diff_posterior_mean_all_steps": float((pred_main - pred32).abs().max()),
        }

    if pf and study is not None:
        out["precision_vs_fp64"] = study
        worst = max(study["f16x3_over_f32_max_err"].values())
        ok = worst <= 2.0
        out["precision_vs_fp64"]["rule"] = (
            "f16x3 is the headline arithmetic iff its max error against fp64 is <= 2x the f32-MFMA mode's on "
            f"every per-particle network; worst ratio here {worst:.2f} -> " + ("holds" if ok else "fails"))
        if precision == "f16x3" and not ok and f32_pass is not None:
            # demote: the unqualified number is the f32-mode one
            out["f16x3_mode"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "dtype": out["dtype"],
                                 "roofline": out.get("roofline")}
            out["value"], out["ms_per_step"] = out["f32_mode"]["value"], out["f32_mode"]["ms_per_step"]
            out["dtype"] = "f32"
            out["roofline"] = out["f32_mode"].get("roofline")
            value = out["value"]
        elif precision == "f16x3" and ok:
            out["dtype"] = ("f32-equivalent via f16x3: fp32 operands split into two round-to-nearest f16 halves "
                            "(x = hi + lo to 2^-22), 3 f16 MFMA products per product, f32 accumulate; error vs fp64 "
                            f"within {worst:.2f}x of the exact-f32-product mode on every network (precision_vs_fp64)")

    if not pf and study is not None:
        # the EKF's only non-f32 arithmetic is the image encoders' (K4 follows the engine's default mode)
        out["precision_vs_fp64"] = study
        worst = max(list(study["f16x3_over_f32_max_err"].values()) + list(study["jacobians"]["f16x3_over_f32_max_err"].values()))
        out["precision_vs_fp64"]["rule"] = (
            "f16x3 (image encoders, dynamics Jacobians) stands iff its max error against fp64 is <= 2x the f32-MFMA "
            f"mode's on every network; worst ratio here {worst:.2f} -> " + ("holds" if worst <= 2.0 else "FAILS: run with MMF_PRECISION=f32"))
        if precision != "f32":
            out["dtype"] = ("f32 (Kalman algebra, per-trajectory networks: exact fp32 products); image encoders and "
                            "dynamics Jacobians f32-equivalent via f16x3 (operands split into two round-to-nearest f16 "
                            f"halves, 3 f16 MFMA products per product, f32 accumulate; error vs fp64 within {worst:.2f}x "
                            "of the exact-f32-product mode on every network, precision_vs_fp64)")

    del run_r, traj_r
    if pf and args.noise != "philox":
        del noise_r
    torch.cuda.empty_cache()

    if world == 1 and pf and not args.no_reference_sizes:
        out["reference_sized"] = reference_sized_regimes(device)

    if world == 1 and not args.no_cpu_baseline:
        cores = min(CPU_THREADS, os.cpu_count() or 1)
        if pf:
            base, parity = cpu_baseline_pf(wl, f, d, cores)
        else:
            base, parity = cpu_baseline_ekf(wl, f, d, cores)
        out["cpu_baseline"] = base
        out["parity_vs_oracle"] = parity
        out["speedup_vs_cpu_baseline"] = value / base["value"]
    else:
        out["cpu_baseline"] = None

    if world == 1 and not args.no_configs:
        t_cfg = time.perf_counter()
        out["configs"] = run_configs(args, device, f, d, K, W)
        out["configs"]["seconds"] = round(time.perf_counter() - t_cfg, 1)
    out["bench_seconds"] = round(time.perf_counter() - t_start, 1)
    # the figures a reader of a truncated record needs, as the LAST key of the line (a driver that keeps the tail of stdout
    # keeps this) and, for the exact-f32 mode, inside `roofline` (kept whole by the driver's parser)
    summary = {"value": out["value"], "ms_per_step": out["ms_per_step"],
               "frac": (out.get("roofline") or {}).get("frac"), "traffic": (out.get("roofline") or {}).get("traffic")}
    f32m = out.get("f32_mode")
    if f32m:
        summary.update(f32_value=f32m["value"], f32_ms_per_step=f32m["ms_per_step"], f32_frac=(f32m.get("roofline") or {}).get("frac"))
        if out.get("roofline") is not None:
            out["roofline"]["f32_mode"] = {"value": f32m["value"], "ms_per_step": f32m["ms_per_step"],
                                           "frac": (f32m.get("roofline") or {}).get("frac"), "peak": MFMA_PEAK["f32"]}
    for key, short in (("C1_door_unimodal_ekf_32", "c1"), ("C2_door_crossmodal_pf_256x1024", "c2"), ("C3_push_crossmodal_pf_1024x4096", "c3"),
                       ("C4_door_crossmodal_ekf_1024_per_gpu_share_of_8192", "c4"), ("C5_push_unimodal_pf_train_32x8192x16", "c5")):
        leg = (out.get("configs") or {}).get(key) or {}
        if "ms_per_step" in leg:
            summary[short + "_ms_per_step"] = leg["ms_per_step"]
            if leg.get("roofline"):
                summary[short + "_frac"] = leg["roofline"]["frac"]
    out["summary"] = {k: (float(f"{v:.6g}") if isinstance(v, float) else v) for k, v in summary.items()}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
