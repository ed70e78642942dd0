"""The HIP engine's recursion (``multimodalfilter_amd.filters``) under known answers, and the
calibrated headline workload under teacher forcing.

``torchfilter`` -- where the recursion T1-T3 lives -- is absent from ``/root/reference``, so
nothing of the reference can pin it.  ``tests/test_oracle_known_answers.py`` pins the oracle's
restatement analytically; this file holds the ENGINE to the same answers, with *forward-only
user models* (a model that implements nothing but the reference's ``forward`` contract,
``/root/reference/crossmodal/door_models/dynamics.py:37-42``, ``door_models/pf.py:63-65``,
``door_models/kf.py:81-83``): the literal "drop in unchanged" path
(``filters.ParticleFilter._propagate`` / ``_measure`` generic branches,
``VirtualSensorExtendedKalmanFilter._predict_pieces`` generic branch).

Tolerances: 1e-4 relative on posterior means / covariances (``north_star``); Monte-Carlo
bounds are stated where they apply.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import models as om
from oracle import tf as otf
from oracle.tf.base import ReplayNoise as OReplay

from _tol import REL_TOL, rel_err


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    return torch.device("cuda:0")


def _system(d=3, seed=0, r_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    A = torch.eye(d) * 0.9 + 0.05 * torch.randn(d, d, generator=g)
    B = 0.1 * torch.randn(d, 7, generator=g)
    L = torch.diag(torch.tensor([0.2, 0.1, 0.15][:d]))
    Rt = torch.diag(torch.tensor([0.3, 0.25, 0.2][:d])) * r_scale
    return A, B, L, Rt


def _kalman_closed_form(A, B, L, Rt, mu, S, us, zs):
    A, B, L, Rt, mu, S, us, zs = (t.double() for t in (A, B, L, Rt, mu, S, us, zs))
    Q, R = L @ L.T, Rt @ Rt.T
    out = []
    for u, z in zip(us, zs):
        mu = mu @ A.T + u @ B.T
        S = A @ S @ A.T + Q
        K = S @ torch.inverse(S + R)
        mu = mu + (z - mu) @ K.T
        S = (torch.eye(len(Q), dtype=torch.float64) - K) @ S
        out.append((mu.clone(), S.clone()))
    return out


def _user_models(base, A, B, L, Rt, dev):
    """Forward-only models against ``base`` = the engine's or the oracle's ``base`` module."""

    class LinearDynamics(base.DynamicsModel):
        def __init__(self):
            super().__init__(state_dim=A.shape[0])
            self.A, self.B, self.L = A.to(dev), B.to(dev), L.to(dev)

        def forward(self, *, initial_states, controls):
            R, d = initial_states.shape
            return initial_states @ self.A.T + controls @ self.B.T, self.L[None].expand(R, d, d)

    class DirectSensor(base.VirtualSensorModel):
        def __init__(self):
            super().__init__(state_dim=A.shape[0])
            self.Rtril = Rt.to(dev)

        def forward(self, *, observations):
            N = observations["z"].shape[0]
            return observations["z"], self.Rtril[None].expand(N, *self.Rtril.shape)

    class GaussianLik(base.ParticleFilterMeasurementModel):
        def __init__(self):
            super().__init__(state_dim=A.shape[0])
            self.Rinv = torch.inverse(Rt @ Rt.T).to(dev)

        def forward(self, *, states, observations):
            e = observations["z"][:, None, :] - states
            return -0.5 * torch.einsum("nmi,ij,nmj->nm", e, self.Rinv, e)

    return LinearDynamics, DirectSensor, GaussianLik


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("loop", [True, False])
def test_engine_ekf_equals_kalman_closed_form(d, loop):
    """K3 behind ``VirtualSensorExtendedKalmanFilter`` with forward-only user models (autograd
    Jacobian default): means at every step and the final covariance equal the Kalman filter's
    closed form (fp64) and the oracle's recursion to 1e-4."""
    import multimodalfilter_amd as mmf

    dev = _dev()
    A, B, L, Rt = _system(d)
    N, T = 37, 6
    g = torch.Generator().manual_seed(1)
    us = torch.randn(T, N, 7, generator=g)
    zs = torch.randn(T, N, d, generator=g)
    mu0 = torch.randn(N, d, generator=g)
    cov0 = (0.1 * torch.eye(d))[None].expand(N, d, d)

    Dyn, Sensor, _ = _user_models(mmf.base, A, B, L, Rt, dev)
    f = mmf.filters.VirtualSensorExtendedKalmanFilter(dynamics_model=Dyn(), virtual_sensor_model=Sensor())
    f.eval()
    f.initialize_beliefs(mean=mu0.to(dev), covariance=cov0.to(dev))
    if loop:
        est = f.forward_loop(observations={"z": zs.to(dev)}, controls=us.to(dev))
    else:
        est = torch.stack([f(observations={"z": zs[t].to(dev)}, controls=us[t].to(dev)) for t in range(T)])
    est = est.cpu()

    ODyn, OSensor, _ = _user_models(otf.base, A, B, L, Rt, "cpu")
    o = otf.filters.VirtualSensorExtendedKalmanFilter(dynamics_model=ODyn(), virtual_sensor_model=OSensor())
    o.initialize_beliefs(mean=mu0, covariance=cov0)
    want_o = o.forward_loop(observations={"z": zs}, controls=us)

    assert rel_err(est, want_o, dims=1) < REL_TOL, rel_err(est, want_o, dims=1)
    cov = f._belief_covariance.cpu()
    assert rel_err(cov, o._belief_covariance) < REL_TOL, rel_err(cov, o._belief_covariance)
    for n in range(N):
        want = _kalman_closed_form(A, B, L, Rt, mu0[n], 0.1 * torch.eye(d), us[:, n], zs[:, n])
        for t in range(T):
            assert rel_err(est[t, n], want[t][0], dims=1) < REL_TOL, (n, t, rel_err(est[t, n], want[t][0], dims=1))
        assert rel_err(cov[n], want[-1][1]) < REL_TOL, (n, rel_err(cov[n], want[-1][1]))


@pytest.mark.parametrize("mode", ["systematic", "multinomial"])
def test_engine_pf_soft_resampling_tracks_oracle_and_kalman(mode):
    """``ParticleFilter(soft_resample_alpha=0.5)`` (upstream option; ``mmf_pf_reweight_resample_soft``)
    with forward-only user models: teacher-forced, every step's posterior mean within 1e-4, ancestors
    equal up to 1e-3 of them and the survivors' importance weights within 1e-4 of the oracle's;
    free-running, both filters stay within Monte-Carlo distance of the Kalman closed form.  In train
    mode with ``resample=True`` the soft draw is differentiable (gradients reach the dynamics)."""
    import multimodalfilter_amd as mmf

    dev = _dev()
    d = 3
    A, B, L, Rt = _system(d, r_scale=3.0)
    N, T, M = 3, 4, 8192
    g = torch.Generator().manual_seed(5)
    us = torch.randn(T, N, 7, generator=g)
    zs = 0.3 * torch.randn(T, N, d, generator=g)
    mu0 = 0.2 * torch.randn(N, d, generator=g)
    cov0 = (0.1 * torch.eye(d))[None].expand(N, d, d)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    uu = [torch.rand((N,) if mode == "systematic" else (N, M), generator=g) for _ in range(T)]

    ODyn, _, OLik = _user_models(otf.base, A, B, L, Rt, "cpu")
    o = otf.filters.ParticleFilter(dynamics_model=ODyn(), measurement_model=OLik(), num_particles=M,
                                   resample_mode=mode, soft_resample_alpha=0.5)
    o.eval()
    o.noise = OReplay([eps0] + eps, uu)
    o.initialize_beliefs(mean=mu0, covariance=cov0)
    want, want_idx, want_lw, beliefs = [], [], [], []
    for t in range(T):
        beliefs.append((o.particle_states, o.particle_log_weights))
        want.append(o(observations={"z": zs[t]}, controls=us[t]))
        want_idx.append(o.last_resample_indices)
        want_lw.append(o.particle_log_weights)
    want = torch.stack(want)
    assert float(want_lw[0].std()) > 1e-3  # soft resampling leaves non-uniform weights

    Dyn, _, Lik = _user_models(mmf.base, A, B, L, Rt, dev)
    f = mmf.filters.ParticleFilter(dynamics_model=Dyn(), measurement_model=Lik(), num_particles=M,
                                   resample_mode=mode, soft_resample_alpha=0.5)
    f.eval()
    f.record_indices = True
    f.noise = mmf.ReplayNoise([eps0], [])
    f.initialize_beliefs(mean=mu0.to(dev), covariance=cov0.to(dev))
    differ = 0
    for t in range(T):
        f.particle_states = beliefs[t][0].to(dev).contiguous()
        f.particle_log_weights = beliefs[t][1].to(dev).contiguous()
        f._spare_states = None
        f.noise = mmf.ReplayNoise([eps[t]], [uu[t]])
        est = f(observations={"z": zs[t].to(dev)}, controls=us[t].to(dev)).cpu()
        assert rel_err(est, want[t], dims=1) < REL_TOL, (t, rel_err(est, want[t], dims=1))
        same = f.last_resample_indices.cpu().long() == want_idx[t]
        differ += int((~same).sum())
        lw = f.particle_log_weights.cpu()
        assert float((torch.logsumexp(lw, dim=1)).abs().max()) < 1e-4
        assert float((lw - want_lw[t])[same].abs().max()) < 1e-4, t
    assert differ <= 1e-3 * T * N * M, f"{differ} of {T * N * M} resample indices differ"

    # free-running through forward_loop (the Python step loop: the native loop resamples with the plain K1)
    f.noise = mmf.ReplayNoise([eps0] + eps, uu)
    f.initialize_beliefs(mean=mu0.to(dev), covariance=cov0.to(dev))
    got = f.forward_loop(observations={"z": zs.to(dev)}, controls=us.to(dev)).cpu()
    for n in range(N):
        kal = _kalman_closed_form(A, B, L, Rt, mu0[n], 0.1 * torch.eye(d), us[:, n], zs[:, n])
        for t in range(T):
            assert float((got[t, n].double() - kal[t][0]).abs().max()) < 0.06, (n, t)
            assert float((want[t, n].double() - kal[t][0]).abs().max()) < 0.06, (n, t)

    # differentiable soft resampling (training backend, resample forced on)
    from multimodalfilter_amd import engine

    class Learnable(Dyn):
        def __init__(self):
            super().__init__()
            self.gain = torch.nn.Parameter(torch.ones(()))

        def forward(self, *, initial_states, controls):
            y, tril = super().forward(initial_states=initial_states, controls=controls)
            return self.gain * y, tril

    engine.set_training_backend("autograd")
    try:
        ft = mmf.filters.ParticleFilter(dynamics_model=Learnable(), measurement_model=Lik(), num_particles=256,
                                        resample=True, resample_mode=mode, soft_resample_alpha=0.5).to(dev)
        ft.train()
        ft.noise = mmf.NoiseSource(1)
        ft.initialize_beliefs(mean=mu0.to(dev), covariance=cov0.to(dev))
        pred = ft.forward_loop(observations={"z": zs.to(dev)}, controls=us.to(dev))
        assert ft.particle_log_weights.requires_grad and float(ft.particle_log_weights.std()) > 1e-3
        pred.square().mean().backward()
        assert ft.dynamics_model.gain.grad is not None and float(ft.dynamics_model.gain.grad.abs()) > 0
    finally:
        engine.set_training_backend(None)


@pytest.mark.parametrize("mode", ["systematic", "multinomial"])
def test_engine_pf_with_user_models_tracks_oracle_and_kalman(mode):
    """K1 behind ``ParticleFilter`` with forward-only user models on identical pre-drawn noise.
    Teacher-forced (engine re-synchronised to the oracle's belief before each of 4 resampling
    steps): posterior means within 1e-4 and resample indices equal except where a last-ulp
    difference of the USER model's arithmetic moves a position across a CDF boundary (bound:
    1e-3 of them).  Free-running: first step within 1e-4, every step of engine and oracle within
    Monte-Carlo distance (0.04 ~ 4 sigma at M = 16,384) of the Kalman closed form."""
    import multimodalfilter_amd as mmf

    dev = _dev()
    d = 3
    A, B, L, Rt = _system(d, r_scale=3.0)  # ESS/M ~ 0.3: the Monte-Carlo error stays well inside the bound
    N, T, M = 3, 4, 16384
    g = torch.Generator().manual_seed(2)
    us = torch.randn(T, N, 7, generator=g)
    zs = 0.3 * torch.randn(T, N, d, generator=g)
    mu0 = 0.2 * torch.randn(N, d, generator=g)
    cov0 = (0.1 * torch.eye(d))[None].expand(N, d, d)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    uu = [torch.rand((N,) if mode == "systematic" else (N, M), generator=g) for _ in range(T)]

    ODyn, _, OLik = _user_models(otf.base, A, B, L, Rt, "cpu")
    o = otf.filters.ParticleFilter(dynamics_model=ODyn(), measurement_model=OLik(), num_particles=M,
                                   resample_mode=mode)
    o.eval()
    o.noise = OReplay([eps0] + eps, uu)
    o.initialize_beliefs(mean=mu0, covariance=cov0)
    want, want_idx, beliefs = [], [], []
    for t in range(T):
        beliefs.append((o.particle_states, o.particle_log_weights))
        want.append(o(observations={"z": zs[t]}, controls=us[t]))
        want_idx.append(o.last_resample_indices)
    want = torch.stack(want)

    Dyn, _, Lik = _user_models(mmf.base, A, B, L, Rt, dev)
    f = mmf.filters.ParticleFilter(dynamics_model=Dyn(), measurement_model=Lik(), num_particles=M,
                                   resample_mode=mode)
    f.eval()
    f.record_indices = True
    f.noise = mmf.ReplayNoise([eps0] + eps, uu)
    f.initialize_beliefs(mean=mu0.to(dev), covariance=cov0.to(dev))
    # (a) teacher-forced: before every step the engine holds the belief the oracle held.  This
    # likelihood is sharply peaked, and systematic resampling walks a CUMULATIVE sum: one ancestor
    # that differs (a last-ulp difference between the GPU's and the CPU's torch arithmetic of the
    # user model) shifts every later CDF boundary of the next step, so free-running particle sets
    # decorrelate within two steps (observed: 1, 564, 11,952 differing indices at steps 0, 1, 2)
    differ = 0
    for t in range(T):
        f.particle_states = beliefs[t][0].to(dev).contiguous()
        f.particle_log_weights = beliefs[t][1].to(dev).contiguous()
        f._spare_states = None
        f.noise = mmf.ReplayNoise([eps[t]], [uu[t]])
        est = f(observations={"z": zs[t].to(dev)}, controls=us[t].to(dev)).cpu()
        assert rel_err(est, want[t], dims=1) < REL_TOL, (t, rel_err(est, want[t], dims=1))
        differ += int((f.last_resample_indices.cpu().long() != want_idx[t]).sum())
    assert differ <= 1e-3 * T * N * M, f"{differ} of {T * N * M} resample indices differ"

    # (b) free-running (step-by-step and through forward_loop = the Python step loop on user
    # models): both filters are Monte-Carlo estimates of the same Kalman posterior
    for use_loop in (False, True):
        f.noise = mmf.ReplayNoise([eps0] + eps, uu)
        f.initialize_beliefs(mean=mu0.to(dev), covariance=cov0.to(dev))
        if use_loop:
            got = f.forward_loop(observations={"z": zs.to(dev)}, controls=us.to(dev)).cpu()
        else:
            got = torch.stack([f(observations={"z": zs[t].to(dev)}, controls=us[t].to(dev)).cpu() for t in range(T)])
        assert rel_err(got[0], want[0], dims=1) < REL_TOL, rel_err(got[0], want[0], dims=1)  # no resampling history yet
        for n in range(N):
            kf = _kalman_closed_form(A, B, L, Rt, mu0[n], 0.1 * torch.eye(d), us[:, n], zs[:, n])
            for t in range(T):
                assert float((got[t, n].double() - kf[t][0]).abs().max()) < 0.04, (use_loop, n, t)
                assert float((want[t, n].double() - kf[t][0]).abs().max()) < 0.04, ("oracle", n, t)


def test_engine_pf_no_resample_matches_importance_sampling():
    """``resample=False``: the particle filter is sequential importance sampling; log-weights
    stay normalised (logsumexp = 0 to 1e-5) and the estimate equals the weighted mean of the
    particles it reports."""
    import multimodalfilter_amd as mmf

    dev = _dev()
    d = 2
    A, B, L, Rt = _system(d)
    Dyn, _, Lik = _user_models(mmf.base, A, B, L, Rt, dev)
    N, M, T = 5, 1000, 3
    f = mmf.filters.ParticleFilter(dynamics_model=Dyn(), measurement_model=Lik(), num_particles=M, resample=False)
    f.eval()
    f.noise = mmf.NoiseSource(5)
    g = torch.Generator().manual_seed(3)
    f.initialize_beliefs(mean=torch.zeros(N, d, device=dev),
                         covariance=(0.1 * torch.eye(d, device=dev))[None].expand(N, d, d))
    for t in range(T):
        est = f(observations={"z": 0.3 * torch.randn(N, d, generator=g).to(dev)},
                controls=torch.randn(N, 7, generator=g).to(dev))
        lw = f.particle_log_weights
        assert float(torch.logsumexp(lw, dim=1).abs().max()) < 1e-5
        mean = (lw.exp()[:, :, None] * f.particle_states).sum(1)
        assert float((mean - est).abs().max()) < 1e-5


# ----------------------------------------------------------------------------- headline workload
@pytest.fixture(scope="module")
def calibrated_door_case():
    """Door crossmodal PF at the bench's calibration (heads scaled to ESS/M ~ 0.25, dynamics
    stabilised), 32 x 4096 particles x 12 steps: the oracle's estimates and per-step beliefs."""
    import bench
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import synthetic

    dev = _dev()
    N, M, d, T = 32, 4096, 3, 12
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    synthetic.stabilise_dynamics(f)
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=4242)
    cal_states = traj["states"][0].to(dev)[:, None, :] + 0.3 * torch.randn((N, 256, d), device=dev)
    synthetic.calibrate_measurement_heads(
        f, {k: traj[k][0].to(dev) for k in ("image", "gripper_pos", "gripper_sensors")}, cal_states)
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=4243)
    sd = {k: v.detach().cpu() for k, v in f.state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    want, _, beliefs = bench.oracle_pf_run("DoorCrossmodalParticleFilter", sd, traj, eps0, eps, us, M)
    # the calibration is what makes the case hard: check it took (peaked at first, never degenerate;
    # later steps flatten as the particle cloud contracts around the stabilised dynamics)
    ess = [b[3] for b in beliefs]
    assert ess[0] < 0.5 and min(ess) > 0.03, ess
    return dict(f=f, traj=traj, eps0=eps0, eps=eps, us=us, want=want, beliefs=beliefs, M=M, N=N, T=T, d=d)


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_calibrated_headline_workload_teacher_forced(calibrated_door_case, precision):
    """The 1e-4 bar on the workload the bench reports (peaked weights, ESS/M ~ 0.25): with the
    engine re-synchronised to the oracle's belief before every step, every step's posterior mean
    is within 1e-4 relative in both arithmetic modes, and the resampler draws the same ancestors
    except at CDF boundaries (measured 2.7e-4 .. 4.7e-4 of them across boxes -- the oracle's
    host-CPU kernels move with the box; bound: 1e-3).  Every differing ancestor is CERTIFIED, not
    just counted (``bench.teacher_forced_parity`` / ``oracle.resample.certify_mismatches``): the
    engine's ancestors equal the integer resampler applied to the engine's own log-weights (K1 exact
    on what it was given: zero exceptions), and each mismatch against the oracle lies within the L1
    distance of the two fixed-point weight vectors of the boundary it crossed (zero unexplained)
    with the two weight vectors themselves within 1e-4 of the total weight of each other (L1; measured
    1.2e-5: the oracle normalises its log-weights before resampling, one more rounding at |logw| ~ 10)."""
    import bench
    from multimodalfilter_amd import engine

    c = calibrated_door_case
    old = engine.DEFAULT_PRECISION
    engine.set_default_precision(precision)
    try:
        r = bench.teacher_forced_parity(c["f"], c["traj"], c["eps"], c["us"], c["beliefs"], c["want"], c["M"])
    finally:
        engine.set_default_precision(old)
    print(precision, r)
    assert r["max_rel_err_posterior_mean"] < REL_TOL, r["max_rel_err_posterior_mean_per_step"]
    assert r["resample_index_mismatch_fraction"] < 1e-3, r["resample_index_mismatches_per_step"]
    cert = r["mismatch_certificate"]
    assert cert["k1_inexact_on_own_weights"] == 0, cert
    assert cert["unexplained"] == 0, cert
    assert cert["max_D_over_Q"] < 1e-4, cert


@pytest.fixture(scope="module")
def tracking_door_case():
    """Door crossmodal PF at the bench's ROUND-4 calibration: measurement heads scaled on the TRACKING filter
    (``bench.calibrate_to_band``: ``BURN_IN`` steps from the 0.1 I initial belief, then the steady-state ESS/M
    moved to ~0.25), 32 x 4096 particles; the oracle's run over 8 burn-in + 12 more steps."""
    import bench
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import synthetic

    dev = _dev()
    N, M, d, burn, T = 32, 4096, 3, 8, 20
    wl = dict(bench.WORKLOADS["door_pf"])
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    synthetic.stabilise_dynamics(f)
    f.num_particles = M
    trace = bench.calibrate_to_band(f, wl, dev, d, M)
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=4242)
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=4243)
    sd = {k: v.detach().cpu() for k, v in f.state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    want, _, beliefs = bench.oracle_pf_run("DoorCrossmodalParticleFilter", sd, traj, eps0, eps, us, M)
    return dict(f=f, wl=wl, trace=trace, traj=traj, eps0=eps0, eps=eps, us=us, want=want, beliefs=beliefs, M=M, N=N, T=T, d=d, burn=burn)


def test_tracking_calibration_holds_the_ess_band(tracking_door_case):
    """benchmark weights must be non-degenerate, ESS/M in [0.05, 0.5] -- over ALL timed steps (round 3
    calibrated at t = 0 and drifted to 0.9).  On a trajectory and noise the calibration never saw, after the
    burn-in every one of 48 steps' batch-mean ESS/M (the ENGINE's own weights) is inside the band, and so is the
    oracle's on its own sample."""
    import bench
    from multimodalfilter_amd import synthetic

    c = tracking_door_case
    dev = _dev()
    B, K = bench.BURN_IN, 48
    print("calibration trace (factor, steady-state ESS/M):", c["trace"])
    assert c["trace"][-1][0] == "all heads" and 0.2 <= c["trace"][-1][2] <= 0.31, c["trace"]
    traj = bench.to_device(synthetic.make_trajectories(state_dim=c["d"], T=B + K, N=c["N"], seed=1234), dev)
    run = bench.FilterRun(c["f"], traj, bench.device_noise(B + K, c["N"], c["M"], c["d"], 99, dev), particles=c["M"])
    burn_ess, ess = bench.engine_ess(run, [(0, B), (B, B + K)])
    summary = bench.ess_summary(ess)
    print(summary, [round(float(x), 3) for x in burn_ess.mean(1)])
    assert summary["steps_with_batch_mean_in_band"] == K, summary
    assert summary["cells_in_band_fraction"] > 0.8, summary
    oracle_ess = [b[3] for b in c["beliefs"]][c["burn"]:]
    assert min(oracle_ess) > 0.05 and max(oracle_ess) < 0.5, oracle_ess


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_tracking_regime_teacher_forced_certificate(tracking_door_case, precision):
    """The mismatch certificate RE-TAKEN in the weight regime the bench now times (ESS/M ~ 0.25 throughout, not the
    0.9 round 3 drifted to).  Holding the band on a CONTRACTED cloud takes heads ~80x the random-init scale, so
    log-likelihoods are O(10..100) and the last-ulp differences of two fp32 summation orders (engine vs torch oracle)
    move the fixed-point weights by 1e-4 .. 3e-4 of their total (``max_D_over_Q``; round 3's flat regime: 2e-5) and
    2e-3 .. 3e-3 of the ancestors across CDF boundaries (measured f32 2.7e-3; hops cross runs of zero-weight
    particles, hence ``max_hop`` in the tens).  What is asserted: teacher-forced on the 12 steps behind the oracle's
    burn-in the posterior means agree within 1e-4 (measured <= 2.6e-5), K1 is exact on its own input, and EVERY
    differing ancestor lies inside its certificate band (zero unexplained)."""
    import bench
    from multimodalfilter_amd import engine

    c = tracking_door_case
    old = engine.DEFAULT_PRECISION
    engine.set_default_precision(precision)
    try:
        r = bench.teacher_forced_parity(c["f"], c["traj"], c["eps"], c["us"], c["beliefs"], c["want"], c["M"], start=c["burn"])
    finally:
        engine.set_default_precision(old)
    print(precision, r)
    assert len(r["max_rel_err_posterior_mean_per_step"]) == c["T"] - c["burn"]
    assert r["max_rel_err_posterior_mean"] < REL_TOL, r["max_rel_err_posterior_mean_per_step"]
    assert r["resample_index_mismatch_fraction"] < 1e-2, r["resample_index_mismatches_per_step"]
    cert = r["mismatch_certificate"]
    assert cert["k1_inexact_on_own_weights"] == 0, cert
    assert cert["unexplained"] == 0, cert
    assert cert["max_D_over_Q"] < 1e-3, cert


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_full_size_native_loop_ancestors_equal_integer_resampler_on_own_weights(precision):
    """K1 inside the native step loop at the headline size (256 x 4096, the bench's calibration:
    peaked weights, ESS/M ~ 0.25), free-running for 10 steps: the ancestors of EVERY step equal
    ``oracle.resample.resample_indices`` applied to the log-weights the engine itself produced --
    step 0 from the belief's log-weights, later steps from the uniform ``-log M`` the native loop
    never materialises (the null-``logw`` shortcuts of ``mmf_pf_reweight_resample``)."""
    import bench
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic
    from oracle import resample as ors

    dev = _dev()
    N, M, d, T = 256, 4096, 3, 10
    old = engine.DEFAULT_PRECISION
    engine.set_default_precision(precision)
    try:
        torch.manual_seed(0)
        f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
        synthetic.stabilise_dynamics(f)
        traj = bench.to_device(synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=777), dev)
        cal_states = traj["states"][0][:, None, :] + 0.3 * torch.randn((N, 256, d), device=dev)
        synthetic.calibrate_measurement_heads(
            f, {k: traj[k][0] for k in ("image", "gripper_pos", "gripper_sensors")}, cal_states)
        eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=778)
        f.record_indices = True
        bench.run_pf(f, traj, (eps0.to(dev), torch.stack(eps).to(dev), torch.stack(us).to(dev)), M)
    finally:
        engine.set_default_precision(old)
    idx = f.last_resample_indices.cpu().numpy()          # (T - 1, N, M): run_pf filters states[1:]
    ll = f.last_log_likelihoods.cpu().numpy()
    steps = idx.shape[0]
    assert idx.shape == (steps, N, M) and ll.shape == idx.shape and steps >= 8
    lw0 = f.last_log_weights_in.cpu().numpy()
    uniform = np.full((N, M), np.float32(-math.log(M)), dtype=np.float32)
    ess = []
    for t in range(steps):
        tot = ((lw0 if t == 0 else uniform) + ll[t]).astype(np.float32)
        want = ors.resample_indices(tot, us[t].numpy(), "systematic")
        bad = int((want != idx[t]).sum())
        assert bad == 0, f"step {t}: {bad} ancestors differ from the integer resampler on the engine's own weights"
        w = ors.quantise(tot)[1].astype(np.float64)
        ess.append(float(((w.sum(1) ** 2) / (w * w).sum(1)).mean() / M))
    print(precision, "ESS/M per step:", [round(e, 3) for e in ess])
    assert min(ess) > 0.02 and ess[0] < 0.6, ess   # peaked but not degenerate: the case is not trivial


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_calibrated_headline_workload_free_running(calibrated_door_case, precision):
    """Left alone for 12 steps the two filters stay statistically indistinguishable: the first
    step (no resampling history) is within 1e-4.  Afterwards one flipped ancestor changes a weight,
    which shifts every later boundary of the next step's cumulative sum, so within a few steps the
    two particle sets are different draws from the same posterior: single posterior means differ
    by up to about one posterior standard deviation on a single trajectory (printed, not asserted:
    it is a property of the chaos, not of the kernels), while the evaluation statistic the
    reference reports -- RMSE against the truth, ``eval_helpers.py:149-160`` -- agrees to under
    1 %.  The 1e-4 bar is met where it can be: teacher-forced, above."""
    import bench
    from multimodalfilter_amd import engine

    c = calibrated_door_case
    dev = _dev()
    old = engine.DEFAULT_PRECISION
    engine.set_default_precision(precision)
    try:
        got = bench.run_pf(c["f"], bench.to_device(c["traj"], dev),
                           (c["eps0"].to(dev), [e.to(dev) for e in c["eps"]], [u.to(dev) for u in c["us"]]),
                           c["M"]).cpu()
    finally:
        engine.set_default_precision(old)
    want, truth = c["want"], c["traj"]["states"][1:]
    assert rel_err(got[0], want[0], dims=1) < REL_TOL, rel_err(got[0], want[0], dims=1)
    # the two runs as draws of the same estimator: their distance against the posterior's own
    # spread (std of the particle cloud the estimate is a weighted mean of)
    worst = 0.0
    for t in range(1, c["T"]):
        spread = c["beliefs"][t][0].std(dim=1)                      # (N, d), belief before step t + 1
        ratio = float(((got[t] - want[t]).abs() / (spread + 1e-6)).max())
        worst = max(worst, ratio)
    print(precision, "largest |engine - oracle| / posterior std over the free run:", worst)
    rm_e = ((got - truth) ** 2).mean((0, 1)).sqrt()
    rm_o = ((want - truth) ** 2).mean((0, 1)).sqrt()
    assert float(((rm_e - rm_o).abs() / rm_o).max()) < 1e-2


def test_particle_count_adaptation_matches_oracle():
    """The reference flips ``num_particles`` 30 <-> 300 in ``train()``
    (``/root/reference/crossmodal/door_models/pf.py:24-27``): a non-resampling step that finds a
    belief of another size adapts it as upstream torchfilter does (copies, then a sample without
    replacement).  Engine and oracle, same explicit uniforms: same particles, same estimates."""
    import multimodalfilter_amd as mmf

    dev = _dev()
    d = 3
    A, B, L, Rt = _system(d)
    N, M0 = 4, 300
    g = torch.Generator().manual_seed(9)
    eps0 = torch.randn((N, M0, d), generator=g)
    sizes = [300, 30, 30, 75]
    eps = [torch.randn((N, m, d), generator=g) for m in sizes]
    perms = [torch.rand((300,), generator=g), torch.rand((30,), generator=g)]
    zs = 0.3 * torch.randn(len(sizes), N, d, generator=g)
    us = torch.randn(len(sizes), N, 7, generator=g)
    cov0 = (0.1 * torch.eye(d))[None].expand(N, d, d)

    def run(base, filters, replay, device):
        Dyn, _, Lik = _user_models(base, A, B, L, Rt, device)
        f = filters.ParticleFilter(dynamics_model=Dyn(), measurement_model=Lik(), num_particles=M0, resample=False)
        f.eval()
        f.noise = replay([eps0] + eps, list(perms))
        f.initialize_beliefs(mean=torch.zeros(N, d).to(device), covariance=cov0.to(device))
        out = []
        for t, m in enumerate(sizes):
            f.num_particles = m
            out.append(f(observations={"z": zs[t].to(device)}, controls=us[t].to(device)).cpu())
            assert f.particle_states.shape == (N, m, d)
        return torch.stack(out), f.particle_states.cpu(), f.particle_log_weights.cpu()

    want, S_o, W_o = run(otf.base, otf.filters, OReplay, "cpu")
    got, S_e, W_e = run(mmf.base, mmf.filters, mmf.ReplayNoise, dev)
    assert rel_err(got, want, dims=1) < REL_TOL, rel_err(got, want, dims=1)
    assert rel_err(S_e, S_o, dims=1) < REL_TOL, rel_err(S_e, S_o, dims=1)   # every particle's state vector
    assert float((W_e - W_o).abs().max()) < 1e-4


def test_f16x3_error_against_fp64_within_2x_of_exact_f32_products_at_full_size():
    """The split-f16 arithmetic (default) against the exact-fp32-product mode, both measured
    against an fp64 evaluation of the same networks on the same fp32 inputs at BASELINE's size
    (256 x 4096 rows): dynamics (un-stabilised random init) and both measurement networks of the
    door crossmodal PF.  Both modes sit two orders below the 1e-4 bar, and f16x3's worst error is
    within 2x of the f32 mode's on every network -- the condition under which ``bench.py`` reports
    the f16x3 throughput as the unqualified number."""
    import bench
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import synthetic

    dev = _dev()
    N, M, d = 256, 4096, 3
    wl = dict(bench.WORKLOADS["door_pf"])
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=2, N=N, seed=11).items()}
    cal_states = traj["states"][0][:, None, :] + 0.3 * torch.randn((N, 256, d), device=dev)
    synthetic.calibrate_measurement_heads(f, {k: traj[k][0] for k in ("image", "gripper_pos", "gripper_sensors")}, cal_states)
    r = bench.precision_errors(wl, f, traj, N, M)
    print(r)
    for mode in ("f32", "f16x3"):
        for net, e in r[mode].items():
            assert e["max_rel"] < 1e-5, (mode, net, e)
    assert max(r["f16x3_over_f32_max_err"].values()) <= 2.0, r["f16x3_over_f32_max_err"]


@pytest.mark.parametrize("scale", [1e-3, 2.0, 30.0])
def test_f16x3_rule_and_failure_mode_at_other_weight_scales(scale):
    """Round-3 verdict, item 9: the <= 2x rule was only ever measured on random-init weights.  Here every layer of
    the per-particle networks is scaled by ``scale`` (1e-3: the f16 "lo" halves of the weights sink into the
    subnormals, the regime of small trained weights; 2: activations grow 2^9-fold through the dynamics network and
    stay in range; 30: they leave the f16 range).  In range the rule must hold; out of range the mode's failure has
    to be FINITE and LOUD: no NaN / inf in any output, and ``engine.check_range`` raises."""
    import bench
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, engine, synthetic

    dev = _dev()
    N, M, d = 32, 4096, 3
    wl = dict(bench.WORKLOADS["door_pf"])
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    nets = [f.dynamics_model._net] + [m._net for m in f.measurement_model.measurement_models]
    with torch.no_grad():
        for net in nets:
            for t in net._sources():
                if t.dim() == 2 and t.shape[0] == 64 and t.shape[1] >= 64:  # the 64 x 64 layers and the join layer
                    t.mul_(scale)
    traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=2, N=N, seed=11).items()}
    engine.clear_range(dev)
    r = bench.precision_errors(wl, f, traj, N, M)
    print(scale, r)
    flagged = False
    try:
        engine.check_range(dev)
    except _abi.MmfError:
        flagged = True
    if scale <= 2.0:
        assert not flagged
        # the <= 2x rule, with one floor: an error below 1.2e-7 of the output scale is within one fp32 ulp (6e-8 ..
        # 1.2e-7) of the value the REFERENCE itself would have to round -- there a ratio of two such numbers says
        # nothing.  Measured at scale 1e-3 (weights of 6e-5: their f16 "lo" halves underflow to zero, K2 has no
        # K4-style 2^8 pre-scale -- it would cost one VALU per activation in the hot loop): 7.3e-8 (f16x3) against
        # 3.5e-8 (f32), ratio 2.07 on one network, 1.15 / 1.33 on the others.
        for net, ratio in r["f16x3_over_f32_max_err"].items():
            assert ratio <= 2.0 or r["f16x3"][net]["max_rel"] <= 1.2e-7, (net, ratio, r["f16x3"][net])
    else:
        assert flagged, "activations of 30^9 must raise the range flag"
    # whatever the range: the f16x3 outputs themselves are finite (saturated, never NaN)
    old = engine.DEFAULT_PRECISION
    engine.set_default_precision("f16x3")
    try:
        states = traj["states"][1][:, None, :] + 0.3 * torch.randn((N, M, d), device=dev)
        obs = {k: traj[k][1] for k in ("image", "gripper_pos", "gripper_sensors")}
        with torch.no_grad():
            pred = f.dynamics_model(initial_states=states.reshape(N * M, d),
                                    controls=traj["controls"][1].repeat_interleave(M, dim=0))[0]
            ll = f.measurement_model(states=states, observations=obs)
        assert bool(torch.isfinite(pred).all()) and bool(torch.isfinite(ll).all())
    finally:
        engine.set_default_precision(old)
        engine.clear_range(dev)


def test_f16x3_out_of_range_step_raises_at_the_step_and_stays_finite():
    """A bare ``forward()`` (no ``forward_loop`` around it) checks the range flag itself: a belief far outside
    the f16 range raises at the step that saw it, and what the kernels wrote is finite -- no NaN reaches a caller."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, engine, synthetic

    dev = _dev()
    N, M, d = 4, 512, 3
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    f.num_particles = M
    traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=2, N=N, seed=3).items()}
    cov = (torch.eye(d, device=dev) * 0.1)[None].expand(N, d, d)
    obs = {k: traj[k][1] for k in ("image", "gripper_pos", "gripper_sensors")}
    old = engine.DEFAULT_PRECISION
    engine.set_default_precision("f16x3")
    try:
        f.noise = mmf.NoiseSource(1)
        f.initialize_beliefs(mean=traj["states"][0] + 4e5, covariance=cov)   # first-layer activations ~ 1e5 > 65504
        with pytest.raises(_abi.MmfError):
            f(observations=obs, controls=traj["controls"][1])
        assert bool(torch.isfinite(f.particle_states).all()) and bool(torch.isfinite(f.particle_log_weights).all())
        # the same filter keeps working on a sane belief afterwards (the flag was consumed)
        f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
        est = f(observations=obs, controls=traj["controls"][1])
        assert bool(torch.isfinite(est).all())
        # a NaN state is reported too (the first ReLU keeps it for the operand split to see)
        bad = traj["states"][0].clone()
        bad[1, 0] = float("nan")
        f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
        f.particle_states[1, 3, 0] = float("nan")
        with pytest.raises(_abi.MmfError):
            f(observations=obs, controls=traj["controls"][1])
    finally:
        engine.set_default_precision(old)
        engine.clear_range(dev)


@pytest.mark.parametrize("strategy", ["julier", "merwe"])
def test_engine_unscented_filter_known_answers(strategy):
    """``VirtualSensorUnscentedKalmanFilter`` (sigma points and moments in HIP, K2 for the points,
    K3 for the update).  (a) forward-only LINEAR user models: equals the Kalman closed form (the
    unscented transform is exact there).  (b) the door task's networks: means and covariances
    within 1e-4 of the oracle's general-form UKF over 5 steps, loop == step-by-step."""
    import multimodalfilter_amd as mmf

    dev = _dev()
    make = (lambda ns: ns.JulierSigmaPointStrategy()) if strategy == "julier" else \
        (lambda ns: ns.MerweSigmaPointStrategy(alpha=0.5, beta=2.0))
    # (a)
    d = 3
    A, B, L, Rt = _system(d)
    N, T = 19, 5
    g = torch.Generator().manual_seed(4)
    us = torch.randn(T, N, 7, generator=g)
    zs = torch.randn(T, N, d, generator=g)
    mu0 = torch.randn(N, d, generator=g)
    cov0 = (0.1 * torch.eye(d))[None].expand(N, d, d)
    Dyn, Sensor, _ = _user_models(mmf.base, A, B, L, Rt, dev)
    f = mmf.filters.VirtualSensorUnscentedKalmanFilter(dynamics_model=Dyn(), virtual_sensor_model=Sensor(),
                                                       sigma_point_strategy=make(mmf.filters))
    f.eval()
    f.initialize_beliefs(mean=mu0.to(dev), covariance=cov0.to(dev))
    est = f.forward_loop(observations={"z": zs.to(dev)}, controls=us.to(dev)).cpu()
    cov = f._belief_covariance.cpu()
    for n in range(N):
        want = _kalman_closed_form(A, B, L, Rt, mu0[n], 0.1 * torch.eye(d), us[:, n], zs[:, n])
        for t in range(T):
            assert rel_err(est[t, n], want[t][0], dims=1) < REL_TOL, (n, t, rel_err(est[t, n], want[t][0], dims=1))
        assert rel_err(cov[n], want[-1][1]) < REL_TOL, (n, rel_err(cov[n], want[-1][1]))

    # (b)
    task = om.TASKS["door"]
    N, T = 33, 5
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1),
           "gripper_pos": torch.randn((T, N, 3), generator=g), "gripper_sensors": torch.randn((T, N, 7), generator=g)}
    ctrl = torch.randn((T, N, 7), generator=g)
    x0 = torch.randn((N, d), generator=g)
    cov0 = (0.1 * torch.eye(d))[None].expand(N, d, d)
    base_o = om.build("DoorKalmanFilter")
    base_o.load_state_dict(om.seeded_state_dict(base_o, seed=14, gain=1.0))
    o = otf.filters.VirtualSensorUnscentedKalmanFilter(dynamics_model=base_o.dynamics_model,
                                                       virtual_sensor_model=base_o.virtual_sensor_model,
                                                       sigma_point_strategy=make(otf.filters))
    o.eval()
    with torch.no_grad():
        o.initialize_beliefs(mean=x0, covariance=cov0)
        want = o.forward_loop(observations=obs, controls=ctrl)
    base_e = mmf.door_models.DoorKalmanFilter()
    base_e.load_state_dict(base_o.state_dict())
    e = mmf.filters.VirtualSensorUnscentedKalmanFilter(dynamics_model=base_e.dynamics_model,
                                                       virtual_sensor_model=base_e.virtual_sensor_model,
                                                       sigma_point_strategy=make(mmf.filters)).to(dev).eval()
    e.initialize_beliefs(mean=x0.to(dev), covariance=cov0.to(dev))
    odev = {k: v.to(dev) for k, v in obs.items()}
    loop = e.forward_loop(observations=odev, controls=ctrl.to(dev))
    cov_loop = e._belief_covariance.clone()
    assert rel_err(loop.cpu(), want, dims=1) < REL_TOL, rel_err(loop.cpu(), want, dims=1)
    assert rel_err(cov_loop, o._belief_covariance) < REL_TOL, rel_err(cov_loop, o._belief_covariance)
    e.initialize_beliefs(mean=x0.to(dev), covariance=cov0.to(dev))
    step = torch.stack([e(observations={k: v[t] for k, v in odev.items()}, controls=ctrl[t].to(dev)) for t in range(T)])
    assert torch.equal(step, loop) and torch.equal(e._belief_covariance, cov_loop)
    # a covariance that is not positive definite is refused, as the Cholesky upstream would
    e.initialize_beliefs(mean=x0.to(dev), covariance=-cov0.to(dev))
    with pytest.raises(ValueError):
        e(observations={k: v[0] for k, v in odev.items()}, controls=ctrl[0].to(dev))
