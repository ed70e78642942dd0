"""The HIP engine's model classes against (a) the golden vectors the REFERENCE's crossmodal
package produced (``tests/golden/*.npz``, see ``oracle/capture_golden.py``) and (b) the CPU
oracle on larger seeded inputs.  Tolerance: 1e-4 relative (``north_star``), written below.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import golden_cases as gc
from oracle import models as om
from oracle.tf.base import ReplayNoise

from _tol import PRIOR_COVARIANCE_SCALE, REFERENCE_FP32_GAP, REL_TOL, rel_err, rel_err_finite


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X")


def _product(case_name: str, task: om.TaskSpec):
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import layers

    class K4Sequential(torch.nn.Sequential):
        """The default image stack evaluated by the HIP image-encoder kernels (K4)."""

        def forward(self, x):
            from multimodalfilter_amd import engine

            return engine.encode_images([self], x[:, 0])[0]

    def k4_encoder(spanning=False):
        seq = layers.image_encoder(64, spanning)
        seq.__class__ = K4Sequential
        return seq

    ns = mmf.door_models if task.name == "door" else mmf.push_models
    P = task.name.capitalize()
    g = lambda suffix: getattr(ns, P + suffix)
    mods = {"image": {"image"}, "possens": {"pos", "sensors"}, "all": {"image", "pos", "sensors"}}
    table = {
        "dynamics_ekf": lambda: g("DynamicsModel")(),
        "dynamics_pf_brent": lambda: ns.DoorDynamicsModelBrent(),
        "dynamics_jacobian": lambda: g("DynamicsModel")(),
        "state_encoder": lambda: layers.vector_encoder(task.state_dim, 64),
        "image_encoder": k4_encoder,
        "image_encoder_spanning": lambda: k4_encoder(True),
        "pf_weight_model": lambda: g("CrossmodalWeightModel")(know_image_blackout=False),
        "pf_weight_model_blackout": lambda: g("CrossmodalWeightModel")(know_image_blackout=True),
        "pf_crossmodal_measurement": lambda: g("CrossmodalParticleFilter")().measurement_model,
        "pf_crossmodal_measurement_masks": lambda: g("CrossmodalParticleFilter")().measurement_model,
        "pf_crossmodal_measurement_blackout": lambda: g("CrossmodalParticleFilterSeq5")().measurement_model,
        "pf_unimodal_measurement": lambda: g("UnimodalParticleFilter")().measurement_model,
        "kf_weight_model": lambda: g("CrossmodalKalmanFilterWeightModel")(state_dim=task.state_dim),
        "crossmodal_virtual_sensor": lambda: g("MeasurementCrossmodalKalmanFilter")().virtual_sensor_model,
        "unimodal_virtual_sensor": lambda: g("MeasurementUnimodalKalmanFilter")().virtual_sensor_model,
        "filter_pf_single": lambda: g("ParticleFilter")(),
        "filter_pf_crossmodal": lambda: g("CrossmodalParticleFilter")(),
        "filter_pf_unimodal": lambda: g("UnimodalParticleFilter")(),
        "filter_pf_crossmodal_seq5": lambda: g("CrossmodalParticleFilterSeq5")(),
        "filter_kf": lambda: g("KalmanFilter")(),
        "filter_kf_crossmodal": lambda: g("CrossmodalKalmanFilter")(),
        "filter_kf_crossmodal_blackout": lambda: g("CrossmodalKalmanFilter")(know_image_blackout=True),
        "filter_kf_crossmodal_masked": lambda: g("CrossmodalKalmanFilter")(),
        "filter_kf_crossmodal_measinit": lambda: g("CrossmodalKalmanFilter")(),
        "filter_kf_unimodal": lambda: g("UnimodalKalmanFilter")(),
        "filter_kf_unimodal_masked": lambda: g("UnimodalKalmanFilter")(),
        "filter_kf_meas_crossmodal": lambda: g("MeasurementCrossmodalKalmanFilter")(),
        "filter_kf_meas_unimodal": lambda: g("MeasurementUnimodalKalmanFilter")(),
    }
    table["virtual_sensor_fixed_noise"] = lambda: g("VirtualSensorModel")()
    for tag, m in mods.items():
        table[f"pf_measurement_{tag}"] = (lambda m: lambda: g("MeasurementModel")(modalities=set(m)))(m)
        table[f"virtual_sensor_{tag}"] = (lambda m: lambda: g("VirtualSensorModel")(modalities=set(m)))(m)
    return table[case_name]()


_PARAMS = [(c, t, n, m) for c in gc.CASES for t in c.tasks for (n, m) in c.shapes]


@pytest.fixture(scope="module")
def golden(golden_dir):
    return {t: np.load(os.path.join(golden_dir, f"{t}.npz")) for t in ("door", "push")}


@pytest.fixture()
def on_gpu():
    _need_gpu()
    gc.DEVICE = "cuda"
    yield
    gc.DEVICE = "cpu"


@pytest.mark.parametrize("case,tname,n,m", _PARAMS,
                         ids=[gc.case_key(c, t, n, m) for c, t, n, m in _PARAMS])
def test_engine_matches_reference_vectors(golden, on_gpu, case, tname, n, m):
    task = om.TASKS[tname]
    z = golden[tname]
    inp = {k[len("input/"):]: z[k] for k in z.files if k.startswith("input/")}
    model = _product(case.name, task)
    out = gc.run_case(case, model, task, inp, n, m)
    prefix = gc.case_key(case, tname, n, m) + "/"
    expected = {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}
    assert set(out) == set(expected) and expected
    for k, want in expected.items():
        got = out[k]
        assert got.shape == want.shape, k
        np.testing.assert_array_equal(np.isneginf(got), np.isneginf(want), err_msg=k)
        if prefix + k in REFERENCE_FP32_GAP:   # the reference's own fp32 arithmetic is not 1e-4 accurate here (tests/_tol.py)
            gap = REFERENCE_FP32_GAP[prefix + k]
            if gap is None:
                assert float(np.abs(got - want).max()) < REL_TOL * PRIOR_COVARIANCE_SCALE, k
            else:
                assert rel_err_finite(got, want) < 4 * gap, f"{k}: rel err {rel_err_finite(got, want):.3e} against the reference's own {gap:.2e}"
            continue
        err = rel_err_finite(got, want)   # per vector / per matrix, 1e-4 of its own Frobenius norm (tests/_tol.py)
        assert err < REL_TOL, f"{k}: rel err {err:.3e}"


@pytest.mark.parametrize("tname,kind,cls", [
    ("door", "crossmodal", "DoorCrossmodalParticleFilter"),
    ("push", "crossmodal", "PushCrossmodalParticleFilter"),
    ("door", "unimodal", "DoorUnimodalParticleFilter"),
    ("push", "single", "PushParticleFilter"),
])
@pytest.mark.parametrize("mode", ["systematic", "multinomial"])
def test_particle_filter_tracks_oracle(tname, kind, cls, mode):
    """Whole PF recursion, N=6, M=1024, T=5, identical pre-drawn randomness: teacher-forced
    posterior means within 1e-4 relative and resample indices equal at every step; the engine's
    three execution paths (step-by-step, forward_loop, native C loop) agree bit for bit."""
    _need_gpu()
    import multimodalfilter_amd as mmf

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d, N, M, T = task.state_dim, 6, 1024, 5
    g = torch.Generator().manual_seed(11)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1),
           "gripper_pos": torch.randn((T, N, 3), generator=g),
           "gripper_sensors": torch.randn((T, N, 7), generator=g)}
    ctrl = torch.randn((T, N, 7), generator=g)
    x0 = torch.randn((N, d), generator=g)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    us = [torch.rand((N,) if mode == "systematic" else (N, M), generator=g) for _ in range(T)]

    oracle = om.ParticleFilter(task, kind, resample_mode=mode)
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=3, gain=1.0))
    oracle.eval()
    oracle.num_particles = M
    oracle.noise = ReplayNoise([eps0] + eps, us)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    want, want_idx, beliefs, drawn_from = [], [], [], []
    resample = oracle._resample

    def resample_and_record():  # the normalised log-weights the oracle's resampler drew from (for the certificate)
        drawn_from.append(oracle.particle_log_weights.clone())
        resample()

    oracle._resample = resample_and_record
    with torch.no_grad():
        oracle.initialize_beliefs(mean=x0, covariance=cov)
        for t in range(T):
            beliefs.append((oracle.particle_states, oracle.particle_log_weights))
            want.append(oracle(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t]))
            want_idx.append(oracle.last_resample_indices.clone())

    engine = mmf.model_types(tname)[cls]()
    engine.load_state_dict(oracle.state_dict())
    engine.to(dev)
    engine.eval()
    engine.num_particles = M
    engine.resample_mode = mode
    engine.record_indices = True
    engine.noise = mmf.ReplayNoise([eps0], [])
    engine.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    assert float((engine.particle_states.cpu() - beliefs[0][0]).abs().max()) < 1e-5
    # Teacher-forced: the engine steps from the belief the oracle held.  Whether a position lands on
    # the other side of a CDF boundary hinges on the last ulp of a log-likelihood, i.e. on the host
    # CPU's torch kernels as much as on the GPU's.  Round 5: no allowance by COUNT any more -- under systematic
    # resampling every differing ancestor must be CERTIFIED (oracle.resample.certify_mismatches: K1 exact on the
    # engine's own weights, and the mismatch within the L1 distance of the two fixed-point weight vectors of the CDF
    # boundary it crossed); the multinomial mode (one uniform per output particle: no certificate) keeps the 1e-3
    # count.  Means to 1e-4.  Bit-exactness of K1 itself is test_k1_indices_bit_exact.
    from oracle import resample as ors

    differ = 0
    for t in range(T):
        engine.particle_states = beliefs[t][0].to(dev).contiguous()
        engine.particle_log_weights = beliefs[t][1].to(dev).contiguous()
        engine._spare_states = None
        engine.noise = mmf.ReplayNoise([eps[t]], [us[t]])
        est = engine(observations={k: v[t].to(dev) for k, v in obs.items()}, controls=ctrl[t].to(dev))
        assert rel_err(est.cpu(), want[t], dims=1) < REL_TOL, f"step {t}: {rel_err(est.cpu(), want[t], dims=1):.2e}"
        got_idx = engine.last_resample_indices.cpu().numpy()
        differ += int((got_idx.astype("int64") != want_idx[t].numpy()).sum())
        if mode == "systematic":
            lw_e = (engine.last_log_weights_in + engine.last_log_likelihoods).cpu().numpy()  # one fp32 add, as K1 does
            u_t = us[t].numpy()
            assert int((ors.resample_indices(lw_e, u_t, mode) != got_idx).sum()) == 0, f"step {t}: K1 inexact on its own weights"
            c = ors.certify_mismatches(drawn_from[t].numpy(), lw_e, u_t, want_idx[t].numpy(), got_idx)
            assert c["unexplained"] == 0, (t, c)
    assert differ <= (1e-2 if mode == "systematic" else 1e-3) * T * N * M, f"{differ} of {T * N * M} resample indices differ"

    # free-running engine, step by step and through forward_loop (observation encoders batched
    # over T*N): identical to each other bit for bit, and -- with these flat random-init
    # likelihoods, where a flipped ancestor moves an estimate by ~spread / M -- within 1e-2 of the
    # oracle's free run (whether a flip happens at all depends on the box's host CPU)
    engine.noise = mmf.ReplayNoise([eps0] + eps, us)
    engine.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    steps = torch.stack([engine(observations={k: v[t].to(dev) for k, v in obs.items()}, controls=ctrl[t].to(dev))
                         for t in range(T)])
    engine.noise = mmf.ReplayNoise([eps0] + eps, us)
    engine.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    engine.use_native_loop = False
    loop = engine.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
    engine.use_native_loop = True
    assert torch.equal(loop, steps)
    torch.testing.assert_close(loop.cpu(), torch.stack(want), rtol=1e-2, atol=1e-2)

    # ... and so does the native step loop (mmf_pf_forward_loop: record_indices off, zero-copy
    # noise blocks), bit for bit against the step-by-step engine path, including the belief
    engine.record_indices = False
    states_ref, logw_ref = engine.particle_states.clone(), engine.particle_log_weights.clone()
    engine.noise = mmf.StackedNoise(eps0.to(dev), torch.stack(eps).to(dev), torch.stack(us).to(dev))
    engine.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    calls = []
    from multimodalfilter_amd import _abi
    real = _abi.pf_forward_loop
    _abi.pf_forward_loop = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        native = engine.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
    finally:
        _abi.pf_forward_loop = real
    assert calls, "fused models must take the native step loop"
    assert torch.equal(native, loop)
    assert torch.equal(engine.particle_states, states_ref)
    assert torch.equal(engine.particle_log_weights, logw_ref)


@pytest.mark.parametrize("tname,cls,okw", [
    ("door", "DoorKalmanFilter", {}),
    ("door", "DoorCrossmodalKalmanFilter", {}),
    ("push", "PushCrossmodalKalmanFilter", {"feedback": "belief", "fix_weight_layout": True}),
    ("push", "PushUnimodalKalmanFilter", {}),
    ("door", "DoorUnimodalKalmanFilter", {}),     # BASELINE config 1 itself: door unimodal EKF, batch 32
])
def test_kalman_filters_track_oracle(tname, cls, okw):
    """EKF recursions, N=32 (config 1's batch), T=6: means and covariances within 1e-4."""
    _need_gpu()
    import multimodalfilter_amd as mmf

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d, N, T = task.state_dim, 32, 6
    g = torch.Generator().manual_seed(5)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1),
           "gripper_pos": torch.randn((T, N, 3), generator=g),
           "gripper_sensors": torch.randn((T, N, 7), generator=g)}
    ctrl = torch.randn((T, N, 7), generator=g)
    x0 = torch.randn((N, d), generator=g)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    oracle = om.build(cls, **okw)
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=4, gain=1.0))
    oracle.eval()
    with torch.no_grad():
        oracle.initialize_beliefs(mean=x0, covariance=cov)
        want = oracle.forward_loop(observations=obs, controls=ctrl)
    engine = mmf.model_types(tname)[cls](**okw)
    engine.load_state_dict(oracle.state_dict())
    engine.to(dev).eval()
    engine.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    got = engine.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
    # element-wise relative (tests/_tol.py): every mean and covariance entry against its own magnitude
    assert rel_err(got, want) < REL_TOL, rel_err(got, want)
    subs_o = list(oracle.filter_models) if hasattr(oracle, "filter_models") else [oracle]
    subs_e = list(engine.filter_models) if hasattr(engine, "filter_models") else [engine]
    for fo, fe in zip(subs_o, subs_e):
        assert rel_err(fe._belief_covariance, fo._belief_covariance) < REL_TOL, rel_err(fe._belief_covariance, fo._belief_covariance)


@pytest.mark.parametrize("resample,T,alpha,method", [
    (False, 3, 1.0, "weighted_average"), (False, 4, 1.0, "weighted_average"), (True, 1, 1.0, "weighted_average"),
    (True, 4, 1.0, "weighted_average"),
    # round 4: torchfilter's other two options run inside the native loop too (they kept the Python loop before)
    (True, 4, 0.5, "weighted_average"), (True, 3, 0.7, "argmax"), (True, 4, 1.0, "argmax"), (False, 3, 1.0, "argmax")])
def test_native_step_loop_equals_stepwise(resample, T, alpha, method):
    """``mmf_pf_forward_loop`` (C host loop) against T separate ``forward`` calls on the same
    pre-drawn randomness: estimates and the final belief are identical bits, whichever of the
    ping-pong buffers the belief ends in (odd / even T, with and without resampling; plain and soft
    resampling; weighted-average and arg-max estimates)."""
    _need_gpu()
    import multimodalfilter_amd as mmf

    dev = torch.device("cuda:0")
    d, N, M = 3, 5, 300
    g = torch.Generator().manual_seed(23)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1).to(dev),
           "gripper_pos": torch.randn((T, N, 3), generator=g).to(dev),
           "gripper_sensors": torch.randn((T, N, 7), generator=g).to(dev)}
    ctrl = torch.randn((T, N, 7), generator=g).to(dev)
    x0 = torch.randn((N, d), generator=g).to(dev)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d).to(dev)
    eps0 = torch.randn((N, M, d), generator=g).to(dev)
    eps = torch.randn((T, N, M, d), generator=g).to(dev)
    us = torch.rand((T, N), generator=g).to(dev)
    from multimodalfilter_amd import _abi

    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    f.num_particles = M
    f.resample = resample
    f.soft_resample_alpha, f.estimation_method = alpha, method

    f.noise = mmf.StackedNoise(eps0, eps, us)
    f.initialize_beliefs(mean=x0, covariance=cov)
    step = torch.stack([f(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t]) for t in range(T)])
    s_ref, w_ref = f.particle_states.clone(), f.particle_log_weights.clone()

    calls = []
    real = _abi.pf_forward_loop
    _abi.pf_forward_loop = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        f.noise = mmf.StackedNoise(eps0, eps, us)
        f.initialize_beliefs(mean=x0, covariance=cov)
        loop = f.forward_loop(observations=obs, controls=ctrl)
    finally:
        _abi.pf_forward_loop = real
    assert calls, "the fused particle filter must take the native step loop"
    assert torch.equal(loop, step)
    assert torch.equal(f.particle_states, s_ref) and torch.equal(f.particle_log_weights, w_ref)
    # the belief stays usable for further single steps
    f.noise = mmf.StackedNoise(None, eps[:1], us[:1])
    f(observations={k: v[0] for k, v in obs.items()}, controls=ctrl[0])


def test_persistent_loop_that_gives_up_is_rerun_as_a_loop_of_launches():
    """The persistent launch needs all of its workgroups resident at once; when a hand-off's bounded spin runs out
    (another process on the GPU) the kernel aborts and raises bit 2 of the range flag.  ``ParticleFilter`` then restores
    the belief it saved, re-runs the call as a loop of launches, switches the persistent form off for the process and
    warns once.  Simulated here: the first (persistent) C call is followed by a scrambled belief and the abort bit --
    estimates and final belief must equal the launch path's, bit for bit, and the next call must not try again."""
    _need_gpu()
    import warnings
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, engine, synthetic

    dev = torch.device("cuda:0")
    N, M, T, d = 8, 300, 5, 3
    old_persist, old_warned = engine.PF_PERSISTENT, engine._PERSISTENT_WARNED
    try:
        torch.manual_seed(3)
        f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
        f.num_particles = M
        traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=T + 1, N=N, seed=23).items()}
        obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
        ctrl = traj["controls"][1:]
        cov = (torch.eye(d, device=dev) * 0.1)[None].expand(N, d, d)
        g = torch.Generator(device=dev).manual_seed(5)
        eps0 = torch.randn((N, M, d), generator=g, device=dev)
        eps = torch.randn((2 * T, N, M, d), generator=g, device=dev)
        us = torch.rand((2 * T, N), generator=g, device=dev)

        def run(persistent, sabotage):
            engine.PF_PERSISTENT, engine._PERSISTENT_WARNED = persistent, False
            taken = []
            real = _abi.pf_forward_loop

            def spy(a, like, *r, **k):
                taken.append(int(a.persistent))
                loc = real(a, like, *r, **k)
                if sabotage and a.persistent:  # what an aborted launch leaves behind: garbage and the abort bit
                    f.particle_states.fill_(float("nan"))
                    f.particle_log_weights.fill_(0.0)
                    engine.range_flag(dev).bitwise_or_(4)
                return loc

            _abi.pf_forward_loop = spy
            try:
                f.noise = mmf.StackedNoise(eps0, eps, us)
                f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
                with warnings.catch_warnings(record=True) as caught:
                    warnings.simplefilter("always")
                    a = f.forward_loop(observations=obs, controls=ctrl)
                    b = f.forward_loop(observations={k: v[:2] for k, v in obs.items()}, controls=ctrl[:2])
                return taken, [str(w.message) for w in caught], a, b, f.particle_states.clone(), f.particle_log_weights.clone()
            finally:
                _abi.pf_forward_loop = real

        ref = run(False, False)
        got = run(True, True)
        assert ref[0] == [0, 0]
        assert got[0] == [1, 0, 0], got[0]          # persistent, its re-run, and a second call that does not try again
        assert len(got[1]) == 1 and "gave up" in got[1][0]
        assert engine.PF_PERSISTENT is False
        for x, y in zip(ref[2:], got[2:]):
            assert torch.equal(x, y)
        engine.check_range(dev)                      # the abort bit was consumed: nothing left to raise
    finally:
        engine.PF_PERSISTENT, engine._PERSISTENT_WARNED = old_persist, old_warned


@pytest.mark.parametrize("cls,N,M,T,precision,noise", [
    ("DoorCrossmodalParticleFilter", 32, 300, 9, "f16x3", "tensor"),    # the reference's evaluation size (door_models/pf.py:24-27)
    ("DoorCrossmodalParticleFilter", 32, 300, 6, "f32", "tensor"),      # the bit-reproducible mode
    ("PushCrossmodalParticleFilter", 7, 300, 5, "f16x3", "philox"),     # d = 2, counter-based noise
    ("DoorUnimodalParticleFilter", 3, 1000, 4, "f16x3", "tensor"),      # two modalities, no weight model
    ("DoorParticleFilter", 5, 77, 5, "f16x3", "tensor"),                # ONE network; ragged tiles (77 = 2 x 32 + 13)
    ("DoorCrossmodalParticleFilterSeq5", 6, 2048, 3, "f16x3", "philox"),  # the largest single-chunk M of a 512-thread K1; blacked-out frames (-inf modality weights)
    # round 6: several rounds of tiles per wave, and K1's 512 threads standing for the launch path's 1024 (two chunks)
    # (eligible: <= 5 rounds per wave, and >= 24 trajectories once M > 2048 -- csrc/pf_persistent.inc, persistent_plan)
    ("DoorCrossmodalParticleFilter", 32, 2048, 4, "f16x3", "tensor"),   # four rounds of tiles per wave and step
    ("DoorCrossmodalParticleFilter", 24, 3000, 3, "f32", "philox"),     # K1: ragged second chunk (768-thread partition), bit-reproducible mode
    ("PushCrossmodalParticleFilter", 24, 3328, 3, "f16x3", "philox"),   # d = 2; K1: 832-thread partition
])
def test_persistent_step_loop_equals_loop_of_launches(cls, N, M, T, precision, noise):
    """``mmf_pf_forward_loop`` with ``MmfPfLoopArgs.persistent`` (ONE launch for all T steps: role-specialised
    workgroups, hand-offs through L2 -- csrc/pf_persistent.inc) against the same loop as launches: estimates of every
    step and the final belief are identical BITS, in both arithmetic modes, for both noise sources, from a
    non-uniform incoming belief (the first call follows a step without resampling), and twice in a row (the
    hand-off words are re-initialised per call)."""
    _need_gpu()
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, engine, synthetic

    dev = torch.device("cuda:0")
    tname = "door" if cls.startswith("Door") else "push"
    d = om.TASKS[tname].state_dim
    old_prec, old_persist = engine.DEFAULT_PRECISION, engine.PF_PERSISTENT
    engine.set_default_precision(precision)
    try:
        torch.manual_seed(3)
        f = mmf.model_types(tname)[cls]().to(dev).eval()
        f.num_particles = M
        traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(
            state_dim=d, T=T + 1, N=N, seed=17, image_blackout_ratio=0.4 if cls.endswith("Seq5") else 0.0).items()}
        obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
        ctrl = traj["controls"][1:]
        cov = (torch.eye(d, device=dev) * 0.1)[None].expand(N, d, d)
        g = torch.Generator(device=dev).manual_seed(5)
        eps0 = torch.randn((N, M, d), generator=g, device=dev)
        eps = torch.randn((T + 3, N, M, d), generator=g, device=dev)   # 1 + T + 2 steps are drawn below
        us = torch.rand((T + 3, N), generator=g, device=dev)

        def run(persistent):
            engine.PF_PERSISTENT = persistent
            taken = []
            real = _abi.pf_forward_loop
            _abi.pf_forward_loop = lambda a, *r, **k: (taken.append(int(a.persistent)), real(a, *r, **k))[1]
            try:
                f.noise = mmf.CounterNoise(99) if noise == "philox" else mmf.StackedNoise(eps0, eps, us)
                f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
                # one step WITHOUT resampling first: the loop then starts from non-uniform log-weights
                f.resample = False
                first = f.forward_loop(observations={k: v[:1] for k, v in obs.items()}, controls=ctrl[:1])
                f.resample = None
                a = f.forward_loop(observations={k: v[1:] for k, v in obs.items()}, controls=ctrl[1:])
                s1, w1 = f.particle_states.clone(), f.particle_log_weights.clone()
                b = f.forward_loop(observations={k: v[1:3] for k, v in obs.items()}, controls=ctrl[1:3])  # again: even T
                return taken, first, a, s1, w1, b, f.particle_states.clone(), f.particle_log_weights.clone()
            finally:
                _abi.pf_forward_loop = real

        ref = run(False)
        got = run(True)
        assert ref[0] == [0, 0, 0] and got[0] == [0, 1, 1], (ref[0], got[0])  # (the no-resampling step keeps the launches)
        for x, y in zip(ref[1:], got[1:]):
            assert torch.equal(x, y)
        assert bool(torch.isfinite(got[2]).all())
    finally:
        engine.set_default_precision(old_prec)
        engine.PF_PERSISTENT = old_persist


@pytest.mark.parametrize("cls,kw,masked", [
    ("DoorKalmanFilter", {}, False),
    ("PushKalmanFilter", {}, False),
    ("DoorCrossmodalKalmanFilter", {}, False),
    ("DoorCrossmodalKalmanFilter", {"feedback": "belief", "fix_weight_layout": True}, False),
    ("PushCrossmodalKalmanFilter", {}, True),
    ("DoorUnimodalKalmanFilter", {}, False),
    ("PushUnimodalKalmanFilter", {}, True),
    # the blackout override (door_models/crossmodal_kf.py:43-98): the batch-global branch is a device word per step
    ("DoorCrossmodalKalmanFilter", {"know_image_blackout": True}, False),
    ("DoorCrossmodalKalmanFilter", {"know_image_blackout": True, "feedback": "belief"}, False),
    ("PushCrossmodalKalmanFilter", {"know_image_blackout": True, "feedback": "belief", "fix_weight_layout": True}, False),
    ("DoorCrossmodalKalmanFilter", {"know_image_blackout": True}, True),
])
def test_native_ekf_loop_equals_stepwise(cls, kw, masked):
    """``mmf_ekf_forward_loop`` (C host loop over K5 + K3) against T separate ``forward``
    calls: estimates, sub-filter beliefs and the fused covariance are identical bits.  With
    ``know_image_blackout`` the sequence mixes steps with and without blacked-out frames (steps 1 and 3 have
    some, the others none), so both branches of the reference's batch-global test are taken inside one loop."""
    _need_gpu()
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi

    dev = torch.device("cuda:0")
    tname = "door" if cls.startswith("Door") else "push"
    d, N, T = om.TASKS[tname].state_dim, 7, 5
    g = torch.Generator().manual_seed(31)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1).to(dev),
           "gripper_pos": torch.randn((T, N, 3), generator=g).to(dev),
           "gripper_sensors": torch.randn((T, N, 7), generator=g).to(dev)}
    ctrl = torch.randn((T, N, 7), generator=g).to(dev)
    x0 = torch.randn((N, d), generator=g).to(dev)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d).to(dev)
    if kw.get("know_image_blackout"):
        obs["image"][1, 2] = 0.0
        obs["image"][3, 0] = 0.0
        obs["image"][3, 5] = 0.0
    f = mmf.model_types(tname)[cls](**kw).to(dev).eval()
    if masked:
        f.enabled_models = [False, True]

    f.initialize_beliefs(mean=x0, covariance=cov)
    step = torch.stack([f(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t]) for t in range(T)])
    subs = list(f.filter_models) if hasattr(f, "filter_models") else [f]
    beliefs = [(m._belief_mean.clone(), m._belief_covariance.clone()) for m in subs]
    wc = getattr(f, "weighted_covariances", None)
    wc = None if wc is None else wc.clone()

    calls = []
    real = _abi.ekf_forward_loop
    _abi.ekf_forward_loop = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        f.initialize_beliefs(mean=x0, covariance=cov)
        loop = f.forward_loop(observations=obs, controls=ctrl)
    finally:
        _abi.ekf_forward_loop = real
    assert calls, "fused Kalman filters must take the native step loop"
    assert torch.equal(loop, step)
    for m, (mu, S) in zip(subs, beliefs):
        if m._initialized:
            assert torch.equal(m._belief_mean, mu) and torch.equal(m._belief_covariance, S)
    if wc is not None:
        assert torch.equal(f.weighted_covariances, wc)


@pytest.mark.parametrize("cls,kw,N,T,precision", [
    ("DoorCrossmodalKalmanFilter", {}, 37, 6, "f16x3"),                                   # K = 2, ragged last tile (37 = 4 x 8 + 5)
    ("DoorCrossmodalKalmanFilter", {"know_image_blackout": True, "feedback": "belief"}, 1027, 5, "f16x3"),  # many workgroups per role; the gate
    ("DoorCrossmodalKalmanFilter", {"feedback": "belief", "fix_weight_layout": True}, 64, 4, "f32"),        # the bit-reproducible mode
    ("PushCrossmodalKalmanFilter", {}, 200, 4, "f16x3"),                                  # d = 2
    ("DoorUnimodalKalmanFilter", {}, 32, 7, "f16x3"),                                     # BASELINE config 1's shape: information-form fusion
    ("DoorKalmanFilter", {}, 100, 5, "f16x3"),                                            # K = 1, no fusion, no hand-offs
    ("DoorCrossmodalKalmanFilter", {}, 4096 + 5, 3, "f16x3"),                             # two tiles per wave, one after the other
])
def test_persistent_ekf_loop_equals_loop_of_launches(cls, kw, N, T, precision):
    """``mmf_ekf_forward_loop`` with ``MmfEkfLoopArgs.persistent`` (ONE launch for all T steps: a wave owns 8 trajectories
    of one sub-filter, the sub-filters' workgroups meet once per step through L2 -- csrc/ekf_persistent.inc) against the
    same loop as 2 T launches: estimates, every sub-filter's belief and the fused covariance are identical BITS, twice in a
    row (the hand-off words are re-initialised per call)."""
    _need_gpu()
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, engine

    dev = torch.device("cuda:0")
    tname = "door" if cls.startswith("Door") else "push"
    d = om.TASKS[tname].state_dim
    old_prec, old_persist = engine.DEFAULT_PRECISION, engine.EKF_PERSISTENT
    engine.set_default_precision(precision)
    try:
        torch.manual_seed(5)
        g = torch.Generator().manual_seed(41)
        obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1).to(dev),
               "gripper_pos": torch.randn((T, N, 3), generator=g).to(dev),
               "gripper_sensors": torch.randn((T, N, 7), generator=g).to(dev)}
        ctrl = torch.randn((T, N, 7), generator=g).to(dev)
        x0 = torch.randn((N, d), generator=g).to(dev)
        cov = (torch.eye(d) * 0.1)[None].expand(N, d, d).to(dev)
        if kw.get("know_image_blackout"):
            obs["image"][1, 2] = 0.0
            obs["image"][3, N - 1] = 0.0
        f = mmf.model_types(tname)[cls](**kw).to(dev).eval()

        def run(persistent):
            engine.EKF_PERSISTENT = persistent
            taken = []
            real = _abi.ekf_forward_loop
            _abi.ekf_forward_loop = lambda a, *r, **k: (taken.append(int(a.persistent)), real(a, *r, **k))[1]
            try:
                f.initialize_beliefs(mean=x0, covariance=cov)
                a = f.forward_loop(observations=obs, controls=ctrl)
                b = f.forward_loop(observations={k: v[:2] for k, v in obs.items()}, controls=ctrl[:2])  # again, from the belief it left
                subs = list(f.filter_models) if hasattr(f, "filter_models") else [f]
                beliefs = [(m._belief_mean.clone(), m._belief_covariance.clone()) for m in subs if m._initialized]
                wc = getattr(f, "weighted_covariances", None)
                return taken, a, b, beliefs, (None if wc is None else wc.clone())
            finally:
                _abi.ekf_forward_loop = real

        ref = run(False)
        got = run(True)
        assert ref[0] == [0, 0] and got[0] == [1, 1], (ref[0], got[0])
        assert torch.equal(ref[1], got[1]) and torch.equal(ref[2], got[2])
        assert bool(torch.isfinite(got[1]).all())
        for (m0, s0), (m1, s1) in zip(ref[3], got[3]):
            assert torch.equal(m0, m1) and torch.equal(s0, s1)
        if ref[4] is not None:
            assert torch.equal(ref[4], got[4])
    finally:
        engine.set_default_precision(old_prec)
        engine.EKF_PERSISTENT = old_persist


@pytest.mark.parametrize("tname", ["door", "push"])
def test_dynamics_forward_loop_native_rollout(tname):
    """``DynamicsModel.forward_loop`` (``/root/reference/crossmodal/eval_helpers.py:135-137``): the native
    rollout (one K7 launch over the T*N controls + ``mmf_dynamics_forward_loop``) equals T separate
    ``forward`` calls bit for bit and the oracle's rollout to 1e-4."""
    _need_gpu()
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d, N, T = task.state_dim, 37, 9
    g = torch.Generator().manual_seed(3)
    ctrl = torch.randn((T, N, 7), generator=g)
    x0 = torch.randn((N, d), generator=g)
    o = om.DynamicsModel(task)
    o.load_state_dict(om.seeded_state_dict(o, seed=2, gain=1.0))
    o.eval()
    with torch.no_grad():
        want, want_tril = o.forward_loop(initial_states=x0, controls=ctrl)
    e = mmf.model_types(tname)[f"{tname.capitalize()}KalmanFilter"]().dynamics_model
    e.load_state_dict(o.state_dict())
    e.to(dev).eval()
    calls = []
    real = _abi.dynamics_forward_loop
    _abi.dynamics_forward_loop = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        got, tril = e.forward_loop(initial_states=x0.to(dev), controls=ctrl.to(dev))
    finally:
        _abi.dynamics_forward_loop = real
    assert calls, "the built-in dynamics models roll out natively"
    x, steps = x0.to(dev), []
    for t in range(T):
        x, _ = e(initial_states=x, controls=ctrl[t].to(dev))
        steps.append(x)
    assert torch.equal(got, torch.stack(steps))
    assert rel_err(got.cpu(), want, dims=1) < REL_TOL, rel_err(got.cpu(), want, dims=1)
    assert tril.shape == want_tril.shape and float((tril.cpu() - want_tril).abs().max()) < 1e-7


@pytest.mark.parametrize("kind", ["ekf", "ukf", "crossmodal"])
def test_fixed_measurement_noise_in_forward_loops(kind):
    """``noise_R_tril`` (``/root/reference/crossmodal/door_models/kf.py:36-37,111-126``: a fixed ``(N, d)``
    diagonal replacing the learned noise head) through every ``forward_loop`` that evaluates the
    sensor on the ``T*N`` flattened rows -- EKF, UKF, fused crossmodal EKF: identical to ``T``
    separate ``forward`` calls, and the EKF equals the oracle's with the same option set."""
    _need_gpu()
    import multimodalfilter_amd as mmf

    dev = torch.device("cuda:0")
    d, N, T = 3, 6, 5
    g = torch.Generator().manual_seed(77)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1),
           "gripper_pos": torch.randn((T, N, 3), generator=g), "gripper_sensors": torch.randn((T, N, 7), generator=g)}
    ctrl = torch.randn((T, N, 7), generator=g)
    x0 = torch.randn((N, d), generator=g)
    fixed = torch.rand((N, d), generator=g) + 0.1
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    odev = {k: v.to(dev) for k, v in obs.items()}
    if kind == "crossmodal":
        f = mmf.door_models.DoorCrossmodalKalmanFilter().to(dev).eval()
        for m in f.filter_models:
            m.virtual_sensor_model.noise_R_tril = fixed.to(dev)
    else:
        base = mmf.door_models.DoorKalmanFilter().to(dev).eval()
        base.virtual_sensor_model.noise_R_tril = fixed.to(dev)
        f = base if kind == "ekf" else mmf.filters.VirtualSensorUnscentedKalmanFilter(
            dynamics_model=base.dynamics_model, virtual_sensor_model=base.virtual_sensor_model).to(dev).eval()
    f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    step = torch.stack([f(observations={k: v[t] for k, v in odev.items()}, controls=ctrl[t].to(dev)) for t in range(T)])
    f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    loop = f.forward_loop(observations=odev, controls=ctrl.to(dev))
    assert torch.equal(loop, step)
    if kind == "ekf":
        o = om.build("DoorKalmanFilter")
        o.load_state_dict({k: v.cpu() for k, v in f.state_dict().items()})
        o.eval()
        o.virtual_sensor_model.noise_R_tril = fixed
        with torch.no_grad():
            o.initialize_beliefs(mean=x0, covariance=cov)
            want = o.forward_loop(observations=obs, controls=ctrl)
        assert rel_err(loop.cpu(), want, dims=1) < REL_TOL, rel_err(loop.cpu(), want, dims=1)


@pytest.mark.parametrize("tname,cls,N,M", [("door", "DoorCrossmodalParticleFilter", 256, 4096),
                                           ("door", "DoorCrossmodalParticleFilter", 256, 1024),
                                           ("push", "PushCrossmodalParticleFilter", 1024, 4096)])
def test_full_size_particle_filter_properties(tname, cls, N, M):
    """BASELINE.json's headline shape (door crossmodal PF, 256 x 4096 particles), config 2 at its own
    size (256 x 1024) and config 3 (push crossmodal PF, 1024 x 4096, resample-bound), 3 steps: the
    native step loop equals step-by-step evaluation bit for bit; after resampling the log-weights
    are exactly -log M, every estimate lies inside its particles' bounding box before resampling,
    and every resampled particle is one of the propagated ones (ancestor indices in range and
    non-decreasing, as systematic resampling yields)."""
    _need_gpu()
    import math

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import synthetic

    dev = torch.device("cuda:0")
    d, T = om.TASKS[tname].state_dim, 3
    traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=5).items()}
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=6)
    eps0, eps, us = eps0.to(dev), torch.stack(eps).to(dev), torch.stack(us).to(dev)
    torch.manual_seed(0)
    f = mmf.model_types(tname)[cls]().to(dev).eval()
    f.num_particles = M
    obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
    ctrl = traj["controls"][1:]
    cov = (torch.eye(d, device=dev) * 0.1)[None].expand(N, d, d)

    f.noise = mmf.StackedNoise(eps0, eps, us)
    f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
    loop = f.forward_loop(observations=obs, controls=ctrl)
    final_states, final_logw = f.particle_states.clone(), f.particle_log_weights.clone()

    f.noise = mmf.StackedNoise(eps0, eps, us)
    f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
    f.record_indices = True
    for t in range(T):
        before = None
        est = f(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t])
        assert torch.equal(est, loop[t])
        idx = f.last_resample_indices.long()
        assert int(idx.min()) >= 0 and int(idx.max()) < M
        assert bool((idx[:, 1:] >= idx[:, :-1]).all())
        # the resampled set is a gather of the propagated set (the spare buffer still holds it)
        propagated = f._spare_states
        gathered = torch.gather(propagated, 1, idx[:, :, None].expand(N, M, d))
        assert torch.equal(gathered, f.particle_states)
        lo, hi = propagated.amin(dim=1), propagated.amax(dim=1)
        assert bool(((est >= lo - 1e-5) & (est <= hi + 1e-5)).all())
    assert torch.equal(f.particle_states, final_states)
    assert torch.equal(f.particle_log_weights, final_logw)
    assert torch.equal(final_logw, torch.full_like(final_logw, -math.log(M)))


def test_batch_coupled_sub_filter_sensor_is_evaluated_per_step():
    """A sub-filter sensor that couples the rows of a batch (here: batch-mean subtraction) is not
    ``row_wise``: ``forward_loop`` must evaluate it one step at a time, so its results equal
    step-by-step ``forward`` (``_encode_loop`` used to flatten T*N rows unconditionally)."""
    _need_gpu()
    import multimodalfilter_amd as mmf

    class Coupled(mmf.base.VirtualSensorModel):
        def __init__(self, inner):
            super().__init__(state_dim=inner.state_dim)
            self.inner = inner

        def forward(self, *, observations):
            z, r = self.inner(observations=observations)
            return z - z.mean(dim=0, keepdim=True), r

    dev = torch.device("cuda:0")
    d, N, T = 3, 6, 4
    g = torch.Generator().manual_seed(77)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1).to(dev),
           "gripper_pos": torch.randn((T, N, 3), generator=g).to(dev),
           "gripper_sensors": torch.randn((T, N, 7), generator=g).to(dev)}
    ctrl = torch.randn((T, N, 7), generator=g).to(dev)
    x0 = torch.randn((N, d), generator=g).to(dev)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d).to(dev)
    torch.manual_seed(1)
    f = mmf.door_models.DoorCrossmodalKalmanFilter().to(dev).eval()
    f.filter_models[0].virtual_sensor_model = Coupled(f.filter_models[0].virtual_sensor_model)
    f.initialize_beliefs(mean=x0, covariance=cov)
    step = torch.stack([f(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t]) for t in range(T)])
    f.initialize_beliefs(mean=x0, covariance=cov)
    loop = f.forward_loop(observations=obs, controls=ctrl)
    assert torch.equal(loop, step)


def test_full_size_crossmodal_ekf_matches_oracle_on_a_shard():
    """BASELINE config 4's per-GPU shape: door crossmodal EKF, 1024 trajectories, 5 steps.  The
    native loop equals step-by-step evaluation bit for bit; with ``fix_weight_layout`` (the
    shard-invariant weight layout) trajectories are independent, so 24 of the 1024 -- the first 12 and
    the LAST 12, i.e. the head and the tail of the image encoder's 4096-image launch chunks (image 4095 =
    trajectory 1023 of step 3) -- must reproduce the CPU oracle run on those 24 alone: means and fused
    covariances within 1e-4."""
    _need_gpu()
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import synthetic

    dev = torch.device("cuda:0")
    N, T, d, S = 1024, 5, 3, 24
    sel = torch.cat([torch.arange(S // 2), torch.arange(N - S // 2, N)])
    kw = {"feedback": "belief", "fix_weight_layout": True}
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=21)
    oracle = om.build("DoorCrossmodalKalmanFilter", **kw)
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=6, gain=1.0))
    oracle.eval()
    obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
    cov = (torch.eye(d) * 0.1)[None]
    with torch.no_grad():
        oracle.initialize_beliefs(mean=traj["states"][0, sel], covariance=cov.expand(S, d, d))
        want = oracle.forward_loop(observations={k: v[:, sel] for k, v in obs.items()}, controls=traj["controls"][1:, sel])
    f = mmf.door_models.DoorCrossmodalKalmanFilter(**kw)
    f.load_state_dict(oracle.state_dict())
    f.to(dev).eval()
    tdev = {k: v.to(dev) for k, v in traj.items()}
    odev = {k: tdev[k][1:] for k in obs}
    f.initialize_beliefs(mean=tdev["states"][0], covariance=cov.to(dev).expand(N, d, d))
    loop = f.forward_loop(observations=odev, controls=tdev["controls"][1:])
    cov_loop = f.weighted_covariances.clone()
    f.initialize_beliefs(mean=tdev["states"][0], covariance=cov.to(dev).expand(N, d, d))
    step = torch.stack([f(observations={k: v[t] for k, v in odev.items()}, controls=tdev["controls"][1 + t]) for t in range(T)])
    assert torch.equal(loop, step) and torch.equal(cov_loop, f.weighted_covariances)
    assert bool(torch.isfinite(loop).all())
    assert rel_err(loop.cpu()[:, sel], want) < REL_TOL, rel_err(loop.cpu()[:, sel], want)
    wc = oracle.weighted_covariances
    assert rel_err(cov_loop.cpu()[sel], wc) < REL_TOL, rel_err(cov_loop.cpu()[sel], wc)


def test_rccl_executes_the_collectives_one_rank():
    """The exchange steps of the multi-GPU layout (X1 all-gather of per-sequence errors, the max-over-ranks clock,
    X2 the flat gradient all-reduce, the barrier) on the ``nccl`` (= RCCL) backend with DEVICE tensors.  This box
    has one GPU and RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the group has one rank
    and ``MMF_DIST_FORCE_COLLECTIVES=1`` keeps the functions from short-circuiting: what is checked is that the
    RCCL code path of each function runs and returns the right values; the two-rank semantics are covered by
    the gloo tests (``tests/test_distributed_cpu.py``)."""
    _need_gpu()
    import subprocess
    import sys

    code = r"""
import os, torch, torch.distributed as dist
import multimodalfilter_amd as mmf
from multimodalfilter_amd import distributed
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
x = torch.arange(12, dtype=torch.float32, device="cuda:0").reshape(4, 3)
g = distributed.all_gather_rows(x)
assert g.is_cuda and torch.equal(g, x)
g1 = distributed.all_gather_rows(x, total_rows=4)   # equal shards: all_gather_into_tensor, no size exchange
assert g1.is_cuda and torch.equal(g1, x)
assert distributed.max_over_ranks(1.25, torch.device("cuda:0")) == 1.25
distributed.barrier()
lin = torch.nn.Linear(5, 3).cuda()
lin(torch.ones(2, 5, device="cuda:0")).sum().backward()
want = [p.grad.clone() for p in lin.parameters()]
n = distributed.all_reduce_gradients(lin)
assert n == 18 and all(torch.equal(p.grad, w) for p, w in zip(lin.parameters(), want))
ps = list(lin.parameters())
assert ps[0].grad.untyped_storage().data_ptr() == ps[1].grad.untyped_storage().data_ptr()   # views of the one reduced buffer
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL-OK")
"""
    env = dict(os.environ, MMF_DIST_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29741",
               RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL-OK" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("index", [0, 1])
def test_likelihood_map_call_of_the_reference_notebook(index):
    """The third caller of the path: ``scripts/door_task/vis_pf_likelihoods.ipynb`` cell 3 evaluates ONE
    unimodal measurement model of the crossmodal particle filter (index 0 image, 1 proprioception / haptics) on a
    239 x 239 grid of states around a trajectory's state -- N = 1, M = 57,121 "particles", far beyond anything the
    filters use -- with one time step's observations.  Same call on the engine, against the CPU oracle: 1e-4."""
    _need_gpu()
    import multimodalfilter_amd as mmf

    dev = torch.device("cuda:0")
    span, resolution = 27 / 5.65, 0.02
    grid = np.mgrid[-span / 2.0: span / 2.0: resolution, -span / 2.0: span / 2.0: resolution]
    _, cols, rows = grid.shape
    assert cols * rows == 57121
    delta = np.concatenate((np.zeros((1, cols * rows, 1)), grid.T.reshape((1, cols * rows, 2))), axis=2)
    g = torch.Generator().manual_seed(3)
    center = torch.randn((1, 1, 3), generator=g)
    states = (center + torch.from_numpy(delta)).to(torch.float32)
    obs = {"image": torch.randn((1, 32, 32), generator=g).clamp(-1, 1),
           "gripper_pos": torch.randn((1, 3), generator=g), "gripper_sensors": torch.randn((1, 7), generator=g)}
    oracle = om.ParticleFilter(om.TASKS["door"], "crossmodal")
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=4, gain=1.0))
    oracle.eval()
    f = mmf.door_models.DoorCrossmodalParticleFilter()
    f.load_state_dict(oracle.state_dict())
    f.to(dev).eval()
    with torch.no_grad():
        want = oracle.measurement_model.measurement_models[index](states=states, observations=obs)
        got = f.measurement_model.measurement_models[index](states=states.to(dev), observations={k: v.to(dev) for k, v in obs.items()})
    assert got.shape == want.shape == (1, 57121)
    assert rel_err(got.cpu(), want, dims=1) < REL_TOL, rel_err(got.cpu(), want, dims=1)   # the (57121,) likelihood map as one vector
    # the map the notebook plots: the same argmax cell, or a tie within the tolerance
    top = int(want.argmax())
    assert float(want[0, top] - want[0, int(got.cpu().argmax())]) <= REL_TOL * abs(float(want[0, top]))
