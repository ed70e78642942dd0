"""Everything of ``bench.py`` that touches the CPU oracle (``oracle/``) or an fp64 reference: the bounded CPU baseline, the
parity legs against the oracle (teacher-forced with a certificate for every differing ancestor, strict-mode
free-running, free-running RMSE), the oracle's own reproducibility floor (round 5: the torch oracle against ITSELF
under another thread count and in fp64), and the per-network error studies against fp64.  None of this runs inside a
timed region: ``bench.py`` is the timed harness and imports these as checkers.
"""
import os
import time

import numpy as np
import torch

CPU_THREADS = 16  # measured on the GPU box's host (2 x EPYC 9575F, 256 hw threads): the oracle
                  # step is fastest at 16 torch threads (8: 0.88x, 32: 0.84x, 64: 0.45x, 128: 0.24x)


def to_device(traj, device):
    return {k: v.to(device) for k, v in traj.items()}


def run_pf(f, traj_dev, noise_dev, M, mode="systematic", traj_offset=0):
    """initialize at states[0] / 0.1 I (eval_helpers.py:125-131), then K steps."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import evaluation

    f.num_particles = M
    f.resample_mode = mode
    if noise_dev[0] == "philox":
        f.noise = mmf.CounterNoise(noise_dev[1], traj_offset=traj_offset)
        return evaluation.run_filter(f, traj_dev)
    eps0, eps, us = noise_dev
    # contiguous (T, ...) blocks: the native step loop reads them in place
    eps = eps if torch.is_tensor(eps) else torch.stack(list(eps))
    us = us if torch.is_tensor(us) else torch.stack(list(us))
    f.noise = mmf.StackedNoise(eps0, eps, us)
    return evaluation.run_filter(f, traj_dev)


def oracle_pf_run(cls, state_dict, traj, eps0, eps, us, M, *, mode="systematic", warm=0, keep_beliefs=True):
    """The oracle particle filter (CPU) over ``traj`` on pre-drawn randomness.  Returns the
    estimates ``(T, N, d)``, the seconds spent on the steps after ``warm``, and per step the
    belief the oracle held BEFORE the step plus the ancestor indices it drew (references, not
    copies: the oracle rebinds its belief tensors every step)."""
    from multimodalfilter_amd import synthetic
    from oracle import models as om
    from oracle.tf.base import ReplayNoise as OReplay

    T = len(eps)
    N, d = traj["states"].shape[1:]
    oracle = om.build(cls, **({"resample_mode": mode} if mode != "systematic" else {}))
    oracle.load_state_dict(state_dict)
    oracle.eval()
    oracle.num_particles = M
    oracle.noise = OReplay([eps0] + list(eps), list(us))
    obs = synthetic.observations_of(traj)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    ests, beliefs, ess, dt = [], [], [], 0.0
    resample = oracle._resample

    resampled_from = []

    def resample_and_record_ess():  # effective sample size of the weights about to be resampled
        w = torch.softmax(oracle.particle_log_weights, dim=1)
        ess.append((1.0 / (w * w).sum(1)) / w.shape[1])  # per trajectory, as a fraction of M
        resampled_from.append(oracle.particle_log_weights)
        resample()

    if keep_beliefs:
        oracle._resample = resample_and_record_ess
    with torch.no_grad():
        oracle.initialize_beliefs(mean=traj["states"][0], covariance=cov)
        for t in range(1, T + 1):
            before = (oracle.particle_states, oracle.particle_log_weights)
            t0 = time.perf_counter()
            ests.append(oracle(observations={k: v[t] for k, v in obs.items()}, controls=traj["controls"][t]))
            if t > warm:
                dt += time.perf_counter() - t0
            if keep_beliefs:
                # [5]: the (normalised) log-weights the oracle's resampler drew from -- for the certificate
                beliefs.append(before + (oracle.last_resample_indices, float(ess[-1].mean()), ess[-1], resampled_from[-1]))
    return torch.stack(ests), dt, beliefs


def teacher_forced_parity(engine_filter, traj, eps, us, beliefs, want, M, *, mode="systematic", start=0):
    """Engine against oracle with the recursion's chaos taken out: before EVERY step the engine's
    belief is overwritten with the belief the oracle held at that point, then one engine step
    runs on the same noise.  What remains is kernel arithmetic: the posterior mean of that step
    and the ancestor indices the resampler draws from log-weights that differ in the last ulp.

    Every differing ancestor is CERTIFIED (``oracle.resample.certify_mismatches``): (i) the engine's
    ancestors equal the integer resampler applied to the engine's OWN log-weights (K1 is exact on
    what it was given), and (ii) each mismatch against the oracle lies within the L1 distance of the
    two fixed-point weight vectors of the CDF boundary it crossed.  ``unexplained`` counts the rest."""
    import multimodalfilter_amd as mmf
    from oracle import resample as ors

    dev = next(engine_filter.parameters()).device
    f = engine_filter
    N, d = traj["states"].shape[1:]
    obs = {k: traj[k] for k in ("image", "gripper_pos", "gripper_sensors")}
    f.num_particles, f.resample_mode = M, mode
    rec, f.record_indices = f.record_indices, True
    f.noise = mmf.ReplayNoise([torch.zeros((N, M, d))], [])
    cov = (torch.eye(d, device=dev) * 0.1)[None].expand(N, d, d)
    f.initialize_beliefs(mean=traj["states"][0].to(dev), covariance=cov)
    scale = max(1.0, float(want.abs().max()))
    errs, flips = [], []
    cert = {"unexplained": 0, "k1_inexact_on_own_weights": 0, "max_slack_used": 0.0, "max_hop": 0, "max_D_over_Q": 0.0}
    for t, (S, W, idx, _, _, lw_o) in enumerate(beliefs):
        if t < start:  # the oracle's burn-in: the weight regime of the timed steps starts behind it
            continue
        f.particle_states = S.to(dev).contiguous()
        f.particle_log_weights = W.to(dev).contiguous()
        f._spare_states = None
        f.noise = mmf.ReplayNoise([eps[t]], [us[t]])
        est = f(observations={k: v[t + 1].to(dev) for k, v in obs.items()}, controls=traj["controls"][t + 1].to(dev))
        errs.append(float((est.cpu() - want[t]).abs().max()) / scale)
        got_idx = f.last_resample_indices.cpu().numpy()
        flips.append(int((got_idx.astype("int64") != idx.numpy()).sum()))
        if mode == "systematic":
            lw_e = (f.last_log_weights_in + f.last_log_likelihoods).cpu().numpy()  # one fp32 add, as K1 does
            u_t = us[t].cpu().numpy()
            cert["k1_inexact_on_own_weights"] += int((ors.resample_indices(lw_e, u_t, mode) != got_idx).sum())
            c = ors.certify_mismatches(lw_o.numpy(), lw_e, u_t, idx.numpy(), got_idx)
            assert c["mismatches"] == flips[-1]
            cert["unexplained"] += c["unexplained"]
            for k in ("max_slack_used", "max_hop", "max_D_over_Q"):
                cert[k] = max(cert[k], c[k])
    f.record_indices = rec
    return {"steps": f"{start + 1} .. {len(beliefs)} of the oracle's run (the first {start} are its burn-in)",
            "max_rel_err_posterior_mean_per_step": errs,
            "max_rel_err_posterior_mean": max(errs),
            "resample_index_mismatches_per_step": flips,
            "resample_index_mismatch_fraction": sum(flips) / float(len(errs) * N * M),
            "mismatch_certificate": cert,
            "oracle_ess_over_m_per_step": [round(b[3], 4) for b in beliefs]}


def strict_parity(cls, engine_filter, traj, eps0, eps, us, M):
    """Row N1: the engine in its exact-fp32 (bit-reproducible) mode against ``oracle/strict`` -- the CPU
    restatement of the same fmaf chains, itself within 2e-6 of the torch oracle -- both FREE-RUNNING over
    the whole sample from the same initial particles: differing ancestors, differing estimate bits, and
    the relative difference of the evaluation RMSE (``eval_helpers.py:149-160``).  All three must be 0."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine
    from oracle import models as om
    from oracle import strict

    dev = next(engine_filter.parameters()).device
    f = engine_filter
    T = len(eps)
    N, d = traj["states"].shape[1:]
    o = om.build(cls)
    o.load_state_dict({k: v.detach().cpu() for k, v in f.state_dict().items()})
    o.eval()
    obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
    ctrl = traj["controls"][1:]
    old, rec = engine.DEFAULT_PRECISION, f.record_indices
    engine.set_default_precision("f32")
    try:
        f.num_particles, f.resample_mode, f.record_indices = M, "systematic", True
        f.noise = mmf.StackedNoise(eps0.to(dev), torch.stack(list(eps)).to(dev), torch.stack(list(us)).to(dev))
        cov = (torch.eye(d, device=dev) * 0.1)[None].expand(N, d, d)
        f.initialize_beliefs(mean=traj["states"][0].to(dev), covariance=cov)
        s = strict.StrictParticleFilter(o)
        s.set_belief(f.particle_states.cpu().numpy(), f.particle_log_weights.cpu().numpy())
        got = f.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev)).cpu().numpy()
        idx = f.last_resample_indices.cpu().numpy()
    finally:
        engine.set_default_precision(old)
        f.record_indices = rec
    t0 = time.perf_counter()
    want, flips = [], []
    for t in range(T):
        want.append(s.step(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t], eps=eps[t], u=us[t]))
        flips.append(int((idx[t] != s.last_resample_indices).sum()))
    want = np.stack(want)
    truth = traj["states"][1:].numpy()
    rm_e = np.sqrt(((got - truth) ** 2).mean((0, 1)))
    rm_o = np.sqrt(((want - truth) ** 2).mean((0, 1)))
    return {"mode": "f32 (strict)", "steps": T, "batch": N, "particles": M,
            "differing_ancestors_per_step": flips, "differing_ancestors": int(sum(flips)),
            "differing_estimate_values": int((got != want).sum()),
            "final_particle_set_identical": bool(np.array_equal(f.particle_states.cpu().numpy(), s.states)),
            "rmse_rel_diff": float((np.abs(rm_e - rm_o) / rm_o).max()),
            "checker": "oracle/strict (C, fmaf chains in the kernels' k-order; <= 2e-6 from the torch oracle)",
            "checker_seconds": round(time.perf_counter() - t0, 1)}


def oracle_self_floor(cls, sd, state_dim, M, cores, *, batch=4, burn=8, steps=8, blackout=0.0):
    """The reference's own reproducibility floor (round 5): the TORCH oracle against ITSELF on one bounded sample --
    same weights, observations and noise -- (a) fp32 on 1 thread instead of ``cores`` (another summation order inside the
    library GEMMs), (b) in fp64 (every network and the weight algebra in double; the resampler's fixed-point CDF is
    fp32 by definition, ``oracle/resample.py``).  Reported exactly as the engine's parity numbers are: teacher-forced
    (the reference run's belief fed before every step: differing ancestors, posterior-mean error) and free-running
    (evaluation RMSE against the truth).  The engine's ``resample_index_mismatch_fraction`` / ``rmse_rel_diff`` are to be
    read against these: a difference the oracle shows against itself is not the engine's."""
    from multimodalfilter_amd import synthetic
    from oracle import models as om
    from oracle.tf.base import ReplayNoise as OReplay

    T = burn + steps
    torch.set_num_threads(cores)
    traj = synthetic.make_trajectories(state_dim=state_dim, T=T, N=batch, seed=4343, image_blackout_ratio=blackout)
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=batch, M=M, state_dim=state_dim, seed=4344)
    want, _, beliefs = oracle_pf_run(cls, sd, traj, eps0, eps, us, M)
    truth = traj["states"][1:]
    rm_ref = ((want - truth) ** 2).mean((0, 1)).sqrt()
    scale = max(1.0, float(want.abs().max()))
    obs_keys = ("image", "gripper_pos", "gripper_sensors")

    def variant(dtype, threads):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        cast = lambda x: x.to(dtype)
        o = om.build(cls)
        o.load_state_dict(sd)
        o = o.to(dtype).eval()
        o.num_particles = M
        tr = {k: cast(v) for k, v in traj.items()}
        cov = (torch.eye(state_dim, dtype=dtype) * 0.1)[None].expand(batch, state_dim, state_dim)
        with torch.no_grad():
            # free-running
            o.noise = OReplay([eps0] + list(eps), list(us))
            o.initialize_beliefs(mean=tr["states"][0], covariance=cov)
            free = torch.stack([o(observations={k: tr[k][t] for k in obs_keys}, controls=tr["controls"][t]) for t in range(1, T + 1)]).float()
            # teacher-forced on the reference run's beliefs
            errs, flips = [], []
            for t in range(burn, T):
                S, W, idx = beliefs[t][0], beliefs[t][1], beliefs[t][2]
                o.particle_states, o.particle_log_weights = cast(S), cast(W)
                o.noise = OReplay([eps[t]], [us[t]])
                est = o(observations={k: tr[k][t + 1] for k in obs_keys}, controls=tr["controls"][t + 1]).float()
                errs.append(float((est - want[t]).abs().max()) / scale)
                flips.append(int((o.last_resample_indices != idx).sum()))
        rm = ((free - truth) ** 2).mean((0, 1)).sqrt()
        torch.set_num_threads(cores)
        return {"teacher_forced": {"max_rel_err_posterior_mean": max(errs), "resample_index_mismatches_per_step": flips,
                                   "resample_index_mismatch_fraction": sum(flips) / float(len(flips) * batch * M)},
                "free_running": {"max_rel_err_posterior_mean_all_steps": float((free - want).abs().max()) / scale,
                                 "rmse": [float(x) for x in rm], "rmse_rel_diff": float(((rm - rm_ref).abs() / rm_ref).max())},
                "seconds": round(time.perf_counter() - t0, 1)}

    return {"sample": f"oracle PF (oracle/), {cls}, batch {batch} x {M} particles, {burn} burn-in + {steps} steps; reference run: fp32, {cores} threads",
            "rmse_reference_run": [float(x) for x in rm_ref],
            "fp32_one_thread": variant(torch.float32, 1),
            "fp64": variant(torch.float64, cores)}


def cpu_baseline_pf(wl, engine_filter, state_dim, cores, sample_batch=32, sample_steps=24, warm=1, burn=8):
    """The oracle (pure torch, fp32, CPU) on a bounded sample of the same workload, with the
    engine run on the identical sample (same weights, observations, noise) for parity:
    teacher-forced (kernel arithmetic, the 1e-4 bar; over the steps behind the oracle's first ``burn`` steps, i.e.
    in the weight regime of the timed steps) in BOTH arithmetic modes against the TORCH oracle, the exact-fp32 mode
    against its bit-exact twin, and free-running (both filters left alone for the whole horizon; resampling flips
    at CDF boundaries decorrelate a few particles)."""
    from multimodalfilter_amd import engine, synthetic

    M = wl["particles"]
    T = sample_steps + warm
    torch.set_num_threads(cores)
    traj = synthetic.make_trajectories(state_dim=state_dim, T=T, N=sample_batch, seed=4242,
                                       image_blackout_ratio=wl.get("blackout", 0.0))
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=sample_batch, M=M, state_dim=state_dim, seed=4243)
    sd = {k: v.detach().cpu() for k, v in engine_filter.state_dict().items()}
    want, dt, beliefs = oracle_pf_run(wl["cls"], sd, traj, eps0, eps, us, M, warm=warm)
    cpu_rate = sample_batch * M * sample_steps / dt

    dev = next(engine_filter.parameters()).device
    got = run_pf(engine_filter, to_device(traj, dev),
                 (eps0.to(dev), [e.to(dev) for e in eps], [u.to(dev) for u in us]), M).cpu()
    scale = max(1.0, float(want.abs().max()))
    rm_e = ((got - traj["states"][1:]) ** 2).mean((0, 1)).sqrt()
    rm_o = ((want - traj["states"][1:]) ** 2).mean((0, 1)).sqrt()
    tf = teacher_forced_parity(engine_filter, traj, eps, us, beliefs, want, M, start=burn)
    old = engine.DEFAULT_PRECISION
    try:  # the exact-fp32 mode against the TORCH oracle (the thing the golden vectors pin), not only against its twin
        engine.set_default_precision("f32")
        tf32 = teacher_forced_parity(engine_filter, traj, eps, us, beliefs, want, M, start=burn)
    finally:
        engine.set_default_precision(old)
    tf32.pop("oracle_ess_over_m_per_step", None)
    parity = {
        "oracle_self": oracle_self_floor(wl["cls"], sd, state_dim, M, cores, blackout=wl.get("blackout", 0.0)),
        "strict_f32_free_running": strict_parity(wl["cls"], engine_filter, traj, eps0, eps, us, M),
        "teacher_forced": tf,
        "teacher_forced_f32": tf32,
        "free_running": {
            # both filters run the whole horizon on their own beliefs: a 1e-7 difference in a
            # log-likelihood occasionally moves a resampling position across a CDF boundary,
            # after which a few particles differ ("Parity")
            "max_rel_err_posterior_mean_step1": float((got[0] - want[0]).abs().max()) / scale,
            "max_rel_err_posterior_mean_all_steps": float((got - want).abs().max()) / scale,
            "rmse_engine": [float(x) for x in rm_e], "rmse_oracle": [float(x) for x in rm_o],
            "rmse_rel_diff": float(((rm_e - rm_o).abs() / rm_o).max()),
        },
    }
    # also a single-thread figure and upstream's multinomial resampling (bounded: ~3 steps each)
    def small(batch, mode, threads, seed, steps=3):
        torch.set_num_threads(threads)
        tr = synthetic.make_trajectories(state_dim=state_dim, T=steps + 1, N=batch, seed=seed)
        e0, e, u = synthetic.draw_filter_noise(T=steps + 1, N=batch, M=M, state_dim=state_dim, seed=seed + 1, mode=mode)
        _, t, _ = oracle_pf_run(wl["cls"], sd, tr, e0, e, u, M, mode=mode, warm=1, keep_beliefs=False)
        torch.set_num_threads(cores)
        return {"value": batch * M * steps / t, "unit": "particle-steps/s", "cores": threads, "resample": mode,
                "sample": f"batch {batch} x {M} particles x {steps} steps after 1 warm-up, {t:.1f} s"}

    extra = {"single_thread": small(sample_batch, "systematic", 1, 5151, steps=6),
             "multinomial": small(sample_batch, "multinomial", cores, 5252)}
    return {"value": cpu_rate, "unit": "particle-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle PF (oracle/), {wl['cls']}, batch {sample_batch} x {M} particles x "
                      f"{sample_steps} steps after {warm} warm-up, {dt:.1f} s, systematic resampling",
            "other_settings": extra}, parity


def precision_errors(wl, engine_filter, traj, batch, particles, chunk=64, raw_dynamics=None):
    """Arithmetic error of the per-particle networks (K2) in each mode against an fp64 evaluation
    of the same networks on the same fp32 inputs, at the benchmark's size: dynamics (noise-free
    prediction) and every unimodal measurement network.  The fp64 checker is the oracle's torch
    module in double precision on the GPU (checker only; evaluated in chunks of trajectories).
    Errors are relative to max(1, max |fp64 value|), the scale the parity bar uses.
    ``raw_dynamics``: a dynamics model to measure instead of the filter's own (the bench scales its
    dynamics head by 2e-3 to keep long runs finite, which hides the network's error behind the
    rounding of ``x + tiny``; the un-scaled twin shows it)."""
    from multimodalfilter_amd import engine
    from oracle import models as om

    dev = next(engine_filter.parameters()).device
    f = engine_filter
    d = traj["states"].shape[-1]
    g = torch.Generator(device="cpu").manual_seed(99)
    states = (traj["states"][1][:batch, None, :].cpu() + 0.3 * torch.randn((batch, particles, d), generator=g)).to(dev)
    obs = {k: traj[k][1][:batch] for k in ("image", "gripper_pos", "gripper_sensors")}
    ctrl = traj["controls"][1][:batch]
    oracle = om.build(wl["cls"])
    oracle.load_state_dict({k: v.detach().cpu() for k, v in f.state_dict().items()})
    dyn_e = f.dynamics_model
    if raw_dynamics is not None:
        dyn_e = raw_dynamics
        oracle.dynamics_model.load_state_dict({k: v.detach().cpu() for k, v in raw_dynamics.state_dict().items()})
    oracle = oracle.double().to(dev).eval()
    meas_e = list(getattr(f.measurement_model, "measurement_models", [f.measurement_model]))
    meas_o = list(getattr(oracle.measurement_model, "measurement_models", [oracle.measurement_model]))
    want = {"dynamics": []}
    with torch.no_grad():
        for lo in range(0, batch, chunk):
            sl = slice(lo, min(batch, lo + chunk))
            x = states[sl].double()
            n = x.shape[0]
            pred, _ = oracle.dynamics_model(initial_states=x.reshape(n * particles, d),
                                            controls=ctrl[sl].double().repeat_interleave(particles, dim=0))
            want["dynamics"].append(pred.reshape(n, particles, d))
            o64 = {k: v[sl].double() for k, v in obs.items()}
            for k, m in enumerate(meas_o):
                want.setdefault(f"measurement_{k}", []).append(m(states=x, observations=o64))
    want = {k: torch.cat(v) for k, v in want.items()}
    out = {}
    old = engine.DEFAULT_PRECISION
    try:
        for mode in ("f32", "f16x3"):
            engine.set_default_precision(mode)
            with torch.no_grad():
                got = {"dynamics": dyn_e(initial_states=states.reshape(batch * particles, d),
                                                    controls=ctrl.repeat_interleave(particles, dim=0))[0]
                       .reshape(batch, particles, d)}
                for k, m in enumerate(meas_e):
                    got[f"measurement_{k}"] = m(states=states, observations=obs)
            out[mode] = {}
            for k, w in want.items():
                e = (got[k].double() - w).abs()
                scale = max(1.0, float(w.abs().max()))
                out[mode][k] = {"max_rel": float(e.max()) / scale, "rms_rel": float(e.pow(2).mean().sqrt()) / scale}
    finally:
        engine.set_default_precision(old)
    ratios = {k: out["f16x3"][k]["max_rel"] / max(out["f32"][k]["max_rel"], 1e-12) for k in want}
    out["f16x3_over_f32_max_err"] = ratios
    out["rows"] = batch * particles
    out["reference"] = "fp64 evaluation (oracle modules in double precision on the device) of the same fp32 inputs"
    return out


def image_encoder_precision_errors(wl, engine_filter, traj, n_images=512):
    """Arithmetic error of every image encoder of the filter (K4) in each mode against an fp64
    evaluation of the same stack (the oracle's torch module in double precision on the GPU; checker
    only) on ``n_images`` of the workload's images; relative to max(1, max |fp64 feature|)."""
    from multimodalfilter_amd import engine
    from oracle import models as om

    dev = next(engine_filter.parameters()).device
    images = traj["image"].reshape((-1,) + tuple(traj["image"].shape[-2:]))[:n_images].contiguous()
    oracle = om.build(wl["cls"])
    oracle.load_state_dict({k: v.detach().cpu() for k, v in engine_filter.state_dict().items()})
    oracle = oracle.double().to(dev).eval()
    stacks_o = {n: m for n, m in oracle.named_modules() if n.endswith("observation_image_layers")}
    stacks_e = {n: m for n, m in engine_filter.named_modules() if n.endswith("observation_image_layers")}
    assert stacks_o.keys() == stacks_e.keys() and stacks_e
    out = {"f32": {}, "f16x3": {}}
    old = engine.DEFAULT_PRECISION
    try:
        with torch.no_grad():
            want = {n: m(images[:, None].double()) for n, m in stacks_o.items()}
            for mode in ("f32", "f16x3"):
                engine.set_default_precision(mode)
                for n, m in stacks_e.items():
                    got = engine.encode_images([m], images)[0]
                    e = (got.double() - want[n]).abs()
                    scale = max(1.0, float(want[n].abs().max()))
                    out[mode][n] = {"max_rel": float(e.max()) / scale, "rms_rel": float(e.pow(2).mean().sqrt()) / scale}
    finally:
        engine.set_default_precision(old)
    out["f16x3_over_f32_max_err"] = {n: out["f16x3"][n]["max_rel"] / max(out["f32"][n]["max_rel"], 1e-12) for n in stacks_e}
    out["images"] = int(images.shape[0])
    out["reference"] = "fp64 evaluation (oracle modules in double precision on the device) of the same fp32 images"
    return out


def jacobian_precision_errors(wl, engine_filter, traj, n_rows=1024):
    """Arithmetic error of K5 (one-step prediction and forward-mode Jacobian of every sub-filter's
    dynamics network) in each mode against the oracle's module in fp64 with its autograd Jacobian, on
    ``n_rows`` (state, control) pairs of the workload; relative to max(1, max |fp64 value|)."""
    from multimodalfilter_amd import engine
    from oracle import models as om

    dev = next(engine_filter.parameters()).device
    oracle = om.build(wl["cls"])
    oracle.load_state_dict({k: v.detach().cpu() for k, v in engine_filter.state_dict().items()})
    oracle = oracle.double().to(dev).eval()
    x = traj["states"][1][:n_rows].contiguous()
    u = traj["controls"][1][:n_rows].contiguous()
    subs_e = [f.dynamics_model for f in engine_filter.filter_models]
    subs_o = [f.dynamics_model for f in oracle.filter_models]
    out = {"f32": {}, "f16x3": {}}
    old = engine.DEFAULT_PRECISION
    try:
        want = []
        for m in subs_o:
            with torch.no_grad():
                pred, _ = m(initial_states=x.double(), controls=u.double())
            want.append((pred, m.jacobian(initial_states=x.double(), controls=u.double()).detach()))
        for mode in ("f32", "f16x3"):
            engine.set_default_precision(mode)
            for k, m in enumerate(subs_e):
                with torch.no_grad():
                    pred, A, _ = m.predict_with_jacobian(x, m.encode_controls(u))
                res = {}
                for name, got, w in (("prediction", pred, want[k][0]), ("jacobian", A, want[k][1])):
                    e = (got.double() - w).abs()
                    res[name] = float(e.max()) / max(1.0, float(w.abs().max()))
                out[mode][f"dynamics_{k}"] = res
    finally:
        engine.set_default_precision(old)
    out["f16x3_over_f32_max_err"] = {k: max(out["f16x3"][k][n] / max(out["f32"][k][n], 1e-12) for n in ("prediction", "jacobian"))
                                     for k in out["f32"]}
    out["rows"] = int(x.shape[0])
    return out


def cpu_baseline_ekf(wl, engine_filter, state_dim, cores, sample_batch=256, sample_steps=6, warm=1):
    from multimodalfilter_amd import evaluation, synthetic
    from oracle import models as om

    T = sample_steps + warm
    torch.set_num_threads(cores)
    traj = synthetic.make_trajectories(state_dim=state_dim, T=T, N=sample_batch, seed=4242,
                                       image_blackout_ratio=wl.get("blackout", 0.0))
    oracle = om.build(wl["cls"], **wl.get("ctor", {}))
    oracle.load_state_dict({k: v.detach().cpu() for k, v in engine_filter.state_dict().items()})
    oracle.eval()
    obs = synthetic.observations_of(traj)
    d = state_dim
    cov = (torch.eye(d) * 0.1)[None].expand(sample_batch, d, d)
    ests = []
    with torch.no_grad():
        oracle.initialize_beliefs(mean=traj["states"][0], covariance=cov)
        for t in range(1, warm + 1):
            ests.append(oracle(observations={k: v[t] for k, v in obs.items()}, controls=traj["controls"][t]))
        t0 = time.perf_counter()
        for t in range(warm + 1, T + 1):
            ests.append(oracle(observations={k: v[t] for k, v in obs.items()}, controls=traj["controls"][t]))
        dt = time.perf_counter() - t0
    want = torch.stack(ests)
    dev = next(engine_filter.parameters()).device
    got = evaluation.run_filter(engine_filter, to_device(traj, dev)).cpu()
    scale = max(1.0, float(want.abs().max()))

    def elementwise(a, b, floor=1e-3):  # every entry against its own magnitude (floor: 1e-3 of the tensor's largest)
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float(((a - b).abs() / b.abs().clamp_min(floor * float(b.abs().max()))).max())

    # posterior covariances after the last step: every sub-filter's belief (and the fused one where the filter keeps it)
    subs_o = list(oracle.filter_models) if hasattr(oracle, "filter_models") else [oracle]
    subs_e = list(engine_filter.filter_models) if hasattr(engine_filter, "filter_models") else [engine_filter]
    def normwise(a, b, dims):  # worst matrix / vector of the batch against its own Frobenius norm
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        axes = tuple(range(b.dim() - dims, b.dim()))
        den = b.pow(2).sum(axes).sqrt()
        return float(((a - b).pow(2).sum(axes).sqrt() / den.clamp_min(1e-3 * float(den.max()))).max())

    cov_err = max(elementwise(fe._belief_covariance, fo._belief_covariance) for fo, fe in zip(subs_o, subs_e))
    cov_fro = max(normwise(fe._belief_covariance, fo._belief_covariance, 2) for fo, fe in zip(subs_o, subs_e))
    parity = {"max_rel_err_posterior_mean": float((got - want).abs().max()) / scale,
              "max_rel_err_posterior_mean_per_vector": normwise(got, want, 1),
              "max_rel_err_posterior_mean_elementwise": elementwise(got, want),
              "max_rel_err_posterior_covariance": cov_err,
              "max_rel_err_posterior_covariance_per_matrix": cov_fro,
              "note": "mean: relative to max(1, largest mean); *_per_vector / *_per_matrix: the worst vector / matrix of the batch against its "
                      "own Frobenius norm (the bar of tests/_tol.py: 1e-4); *_elementwise and covariance: every entry relative to "
                      "max(|its own value|, 1e-3 x the tensor's largest); sub-filter beliefs after the sample's last step"}
    return {"value": sample_batch * sample_steps / dt, "unit": "trajectory-steps/s", "cores": cores,
            "kind": "port",
            "sample": f"oracle EKF (oracle/), {wl['cls']}, batch {sample_batch} x {sample_steps} steps "
                      f"after {warm} warm-up, {dt:.1f} s"}, parity


# ------------------------------------------------------------------------------ `configs` legs
def _leg_parity_pf(wl, f, d, M, device, n=4, steps=2):
    """One parity number for a particle-filter leg: the engine against the oracle on ``n`` trajectories x ``M``
    particles, same weights / observations / pre-drawn noise, ``steps`` free-running steps from the initial belief.
    The FIRST step's posterior means are kernel arithmetic only (no resampling lies in front of them)."""
    from multimodalfilter_amd import synthetic

    torch.set_num_threads(min(CPU_THREADS, os.cpu_count() or 1))
    traj = synthetic.make_trajectories(state_dim=d, T=steps, N=n, seed=31337, image_blackout_ratio=wl.get("blackout", 0.0))
    if wl.get("blackout", 0.0) > 0:
        traj["image"][1, 0] = 0.0  # at least one blacked-out frame in the checked step
    eps0, eps, us = synthetic.draw_filter_noise(T=steps, N=n, M=M, state_dim=d, seed=31338)
    sd = {k: v.detach().cpu() for k, v in f.state_dict().items()}
    want, _, _ = oracle_pf_run(wl["cls"], sd, traj, eps0, eps, us, M, keep_beliefs=False)
    got = run_pf(f, to_device(traj, device), (eps0.to(device), [e.to(device) for e in eps], [u.to(device) for u in us]), M).cpu()
    scale = max(1.0, float(want.abs().max()))
    return {"max_rel_err_posterior_mean_step1_vs_oracle": float((got[0] - want[0]).abs().max()) / scale,
            "sample": f"{n} trajectories x {M} particles, same weights / inputs / noise; CPU oracle"}
