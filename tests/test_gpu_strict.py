"""Row N1 of the scope table: ``north_star``'s "bit-exact resample indices under fixed seed ...
posterior RMSE matching reference" on the WHOLE filter as the reference drives it
(``/root/reference/crossmodal/eval_helpers.py:125-160``: ``initialize_beliefs`` + ``forward_loop``).

In the exact-fp32 mode (``engine.set_default_precision("f32")``) every number the engine produces is
reproducible bit for bit by ``oracle/strict`` (a CPU restatement of the kernels' fmaf chains, checked
against the torch oracle to 2e-6 in ``tests/test_strict_cpu.py``).  Stage by stage -- K7 encoders,
K4 image encoders, K2 dynamics / measurement incl. the crossmodal logsumexp, K1's estimate -- and
then the free-running filter at the bench's calibration: ZERO differing ancestors over the whole run,
identical estimates, identical final particle set.  Comparisons are ``==`` on fp32 values."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import models as om
from oracle import resample as ors
from oracle import strict


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")
    return torch.device("cuda:0")


@pytest.fixture()
def f32_mode():
    from multimodalfilter_amd import engine

    old = engine.DEFAULT_PRECISION
    engine.set_default_precision("f32")
    yield
    engine.set_default_precision(old)


def _pair(cls, seed=5, calibrate=False, N_cal=8):
    """(engine filter on the GPU, oracle filter on the CPU) with identical weights."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import synthetic

    dev = _dev()
    task = "door" if cls.startswith("Door") else "push"
    o = om.build(cls)
    o.load_state_dict(om.seeded_state_dict(o, seed=seed, gain=1.0))
    o.eval()
    e = mmf.model_types(task)[cls]()
    e.load_state_dict(o.state_dict())
    e.to(dev).eval()
    if calibrate:
        synthetic.stabilise_dynamics(e)
        d = e.state_dim
        traj = synthetic.make_trajectories(state_dim=d, T=1, N=N_cal, seed=99)
        g = torch.Generator().manual_seed(98)
        cal = traj["states"][0][:, None, :] + 0.3 * torch.randn((N_cal, 256, d), generator=g)
        synthetic.calibrate_measurement_heads(
            e, {k: traj[k][0].to(dev) for k in ("image", "gripper_pos", "gripper_sensors")}, cal.to(dev))
        o.load_state_dict({k: v.detach().cpu() for k, v in e.state_dict().items()})
    return e, o


def _eq(got, want, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = int((got != want).sum())
    if bad:
        err = float(np.nanmax(np.abs(got.astype(np.float64) - want.astype(np.float64))))
        raise AssertionError(f"{what}: {bad} of {got.size} values differ (max abs {err:.3e})")


@pytest.mark.parametrize("cls", ["DoorCrossmodalParticleFilter", "PushCrossmodalParticleFilter"])
def test_strict_stages_bit_exact(f32_mode, cls):
    from multimodalfilter_amd import engine

    dev = _dev()
    e, o = _pair(cls)
    s = strict.StrictParticleFilter(o)
    d = e.state_dim
    N, M = 7, 200        # ragged: 1400 rows = 43.75 tiles of 32
    g = torch.Generator().manual_seed(11)
    obs = {"image": (torch.randn((N, 32, 32), generator=g) * 0.5).clamp(-1, 1),
           "gripper_pos": torch.randn((N, 3), generator=g), "gripper_sensors": torch.randn((N, 7), generator=g)}
    obs["image"][2] = 0.0
    ctrl = torch.randn((N, 7), generator=g)
    states = torch.randn((N, M, d), generator=g)
    eps = torch.randn((N, M, d), generator=g)
    odev = {k: v.to(dev) for k, v in obs.items()}

    # K7: control encoder + hoisted control half of the join layer
    dyn = e.dynamics_model
    cb = dyn.encode_controls(ctrl.to(dev))["bias"]
    want_cb = s.control_bias(ctrl)
    _eq(cb, want_cb, "K7 control bias")
    # K4 (f32 path): every image encoder of the measurement model
    meas = e.measurement_model
    encs = [m.observation_image_layers for m in list(meas.measurement_models) + [meas.crossmodal_weight_model]
            if "image" in m.modalities]
    o_encs = [m.observation_image_layers for m in list(o.measurement_model.measurement_models) + [o.measurement_model.crossmodal_weight_model]
              if "image" in m.modalities]
    feats = engine.encode_images(encs, odev["image"])
    for k, (fe, oe) in enumerate(zip(feats, o_encs)):
        _eq(fe, strict.image_encoder(oe, obs["image"]), f"K4 image encoder {k}")
    # K7 + K4: hoisted observation halves and modality log-weights
    ctx = meas.encode_observations(odev)
    biases, beta = s.measurement_terms(obs)
    for i in range(2):
        _eq(ctx[f"m{i}.bias"], biases[i], f"K7 measurement bias {i}")
    _eq(ctx["modality_log_weights"], beta, "K7 modality log-weights")
    # K2 dynamics incl. sigmoid gate and noise
    tril = dyn.scale_tril()
    prop = engine.run_dynamics(dyn._net, states.to(dev), cb, eps.to(dev), tril)
    want_prop = strict.dynamics_step(s.dyn_net, states, want_cb, eps, tril.cpu())
    _eq(prop, want_prop, "K2 dynamics")
    # K2 measurement incl. the crossmodal logsumexp
    ll = meas.forward_encoded(prop, ctx)
    want_ll = np.empty((N, M), dtype=np.float32)
    for i in range(2):
        strict.measure_step(s.meas_nets[i], want_prop, biases[i], want_ll, mod_logw=beta.reshape(-1)[i:], stride=2,
                            combine=i > 0)
    _eq(ll, want_ll, "K2 measurement")


@pytest.mark.parametrize("M", [30, 300, 1000, 4096, 5000])
def test_strict_k1_estimate_bit_exact(M):
    from multimodalfilter_amd import _abi

    dev = _dev()
    N, d = 5, 3
    g = torch.Generator().manual_seed(M)
    ll = torch.randn((N, M), generator=g) * 2
    lw = torch.full((N, M), float(np.float32(-math.log(M))))
    x = torch.randn((N, M, d), generator=g)
    u = torch.rand((N,), generator=g)
    est = torch.empty((N, d), device=dev)
    out = torch.empty((N, M, d), device=dev)
    lwo = torch.empty((N, M), device=dev)
    idx = torch.empty((N, M), dtype=torch.int32, device=dev)
    _abi.pf_reweight_resample(ll.to(dev), lw.to(dev), x.to(dev), u.to(dev), est, out, lwo, idx, 1, 1.0)
    tot = (lw + ll).numpy()
    _eq(est, strict.estimate(tot, x), "K1 estimate")
    _eq(idx, ors.resample_indices(tot, u.numpy(), "systematic"), "K1 ancestors")


@pytest.mark.parametrize("cls,N,M,T", [("DoorCrossmodalParticleFilter", 32, 4096, 12),
                                       ("DoorCrossmodalParticleFilter", 256, 1024, 4),   # BASELINE config 2 at its own size
                                       ("DoorCrossmodalParticleFilter", 256, 4096, 3),   # the headline shape itself: 3.1 M ancestors
                                       ("PushCrossmodalParticleFilter", 6, 300, 20),
                                       ("DoorUnimodalParticleFilter", 4, 1000, 8)])
def test_strict_free_running_filter_bit_exact(f32_mode, cls, N, M, T):
    """The engine's native ``forward_loop`` and the CPU twin, each left alone for the whole run on the
    same initial particles, noise and uniforms (calibrated heads: peaked weights, so resampling matters):
    every ancestor of every step identical, every estimate identical, the final particle set identical."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import synthetic

    dev = _dev()
    e, o = _pair(cls, calibrate=True)
    d = e.state_dim
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=4242)
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=4243)
    obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
    ctrl = traj["controls"][1:]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)

    e.num_particles = M
    e.record_indices = True
    e.noise = mmf.StackedNoise(eps0.to(dev), torch.stack(eps).to(dev), torch.stack(us).to(dev))
    e.initialize_beliefs(mean=traj["states"][0].to(dev), covariance=cov.to(dev))
    init_states, init_logw = e.particle_states.cpu().numpy().copy(), e.particle_log_weights.cpu().numpy().copy()
    # the initial particle set is an INPUT of the recursion; it agrees with torch's Cholesky sampling
    want0 = traj["states"][0][:, None, :] + math.sqrt(0.1) * eps0
    assert float((torch.from_numpy(init_states) - want0).abs().max()) < 1e-6
    got = e.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev)).cpu().numpy()
    got_idx = e.last_resample_indices.cpu().numpy()
    assert got_idx.shape == (T, N, M)

    s = strict.StrictParticleFilter(o)
    s.set_belief(init_states, init_logw)
    ess, wants = [], []
    for t in range(T):
        want = s.step(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t], eps=eps[t], u=us[t])
        _eq(got_idx[t], s.last_resample_indices, f"step {t}: ancestors")
        _eq(got[t], want, f"step {t}: posterior mean")
        wants.append(want)
        w = ors.quantise(s.last_total_log_weights)[1].astype(np.float64)
        ess.append(float(((w.sum(1) ** 2) / (w * w).sum(1)).mean() / M))
    _eq(e.particle_states, s.states, "final particle set")
    print(cls, "ESS/M per step:", [round(x, 3) for x in ess])
    assert min(ess) < 0.7, ess  # the weights are not flat: resampling had something to decide
    # RMSE against the truth (eval_helpers.py:149-160): identical by construction; stated for the record
    truth = traj["states"][1:].numpy()
    rmse = lambda est: np.sqrt(((est - truth) ** 2).mean((0, 1)))
    assert np.array_equal(rmse(got), rmse(np.stack(wants)))


def test_counter_noise_kernels_match_the_checker():
    """``mmf_philox_normals`` / ``mmf_philox_uniforms`` (and therefore the draws ``mmf_pf_dynamics_philox``
    makes in its epilogue: same header, ``include/mmf_philox.h``) against ``oracle.strict``: every bit."""
    from multimodalfilter_amd import _abi

    dev = _dev()
    for seed, step, traj0, (N, M, d) in [(0, 0, 0, (3, 300, 3)), (2 ** 63 + 12345, 999, 7, (5, 4096, 2)),
                                          (77, 3, 100000, (2, 65, 4))]:
        out = torch.empty((N, M, d), device=dev)
        _abi.philox_normals(seed, step, traj0, out)
        _eq(out, strict.philox_normals(seed, step, N, M, d, traj0), "philox normals")
    u = torch.empty((40, 33), device=dev)
    _abi.philox_uniforms(5, 11, 2, u)
    _eq(u, strict.philox_uniforms(5, 11, 40, 33, traj0=2), "philox uniforms")


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_counter_noise_in_kernel_equals_materialised_tensor(precision):
    """The native loop with ``CounterNoise`` (noise generated inside the dynamics kernel, no
    ``(T, N, M, d)`` tensor) == the same loop fed the checker's materialised draws through
    ``StackedNoise`` == step-by-step evaluation with ``CounterNoise``: estimates and belief, bit for bit."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic

    dev = _dev()
    N, M, d, T, seed = 9, 1000, 3, 6, 4242
    old = engine.DEFAULT_PRECISION
    engine.set_default_precision(precision)
    try:
        torch.manual_seed(0)
        f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
        f.num_particles = M
        traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=5).items()}
        obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
        ctrl = traj["controls"][1:]
        cov = (torch.eye(d, device=dev) * 0.1)[None].expand(N, d, d)

        def run(noise, loop=True):
            f.noise = noise
            f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
            if loop:
                est = f.forward_loop(observations=obs, controls=ctrl)
            else:
                est = torch.stack([f(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t]) for t in range(T)])
            return est, f.particle_states.clone(), f.particle_log_weights.clone()

        a = run(mmf.CounterNoise(seed))
        eps0 = torch.from_numpy(strict.philox_normals(seed, 0, N, M, d)).to(dev)
        eps = torch.from_numpy(np.stack([strict.philox_normals(seed, 1 + t, N, M, d) for t in range(T)])).to(dev)
        us = torch.from_numpy(strict.philox_uniforms(seed, 0, T, N)).to(dev)
        b = run(mmf.StackedNoise(eps0, eps, us))
        c = run(mmf.CounterNoise(seed), loop=False)
    finally:
        engine.set_default_precision(old)
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)
    # sharding: trajectories 4.. of the batch, run alone with traj_offset = 4, draw the same noise
    assert torch.equal(torch.from_numpy(strict.philox_normals(seed, 3, 5, M, d, traj0=4)).to(dev), eps[2][4:9])
