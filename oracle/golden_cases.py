"""Golden-vector case table shared by ``oracle/capture_golden.py`` (which runs each case on
the REFERENCE's classes, in the build container only) and ``tests/test_oracle_golden.py``
(which runs the same case on ``oracle.models`` and compares).  TEST INFRASTRUCTURE.

A case = (name, reference class path, oracle factory, runner).  Runners are duck-typed:
they only use the torchfilter-style keyword API both implementations expose.
Inputs are seeded here; weights are seeded *by state_dict key*
(``oracle.models.seeded_state_dict``), so no multi-megabyte tensors are stored.
"""
from typing import Callable, Dict, List, NamedTuple

import numpy as np
import torch

from . import models as om
from .tf.base import ReplayNoise

T_STEPS = 4
N_MAX = 4
M_MAX = 8
WEIGHT_GAIN = 1.4


def make_inputs(task: om.TaskSpec, seed: int = 1234) -> Dict[str, np.ndarray]:
    """Observations / controls / states for ``T_STEPS`` x ``N_MAX`` (+ particles)."""
    rng = np.random.RandomState(seed + task.state_dim)
    d = task.state_dim
    T, N, M = T_STEPS, N_MAX, M_MAX
    f = np.float32
    img = np.clip(rng.standard_normal((T, N, 32, 32)) * 0.5, -1, 1).astype(f)
    img_dark = img.copy()
    img_dark[1, 1] = 0.0  # blacked-out frames (tasks/_door.py:188-197 semantics)
    img_dark[1, 2] = 0.0
    img_dark[3, 0] = 0.0
    return {
        "image": img,
        "image_dark": img_dark,
        "gripper_pos": rng.standard_normal((T, N, task.pos_dim)).astype(f),
        "gripper_sensors": rng.standard_normal((T, N, task.sensors_dim)).astype(f),
        "controls": rng.standard_normal((T, N, task.control_dim)).astype(f),
        "states0": rng.standard_normal((N, d)).astype(f),
        "particles": rng.standard_normal((N, M, d)).astype(f),
        "eps_init": rng.standard_normal((N, M, d)).astype(f),
        "eps": rng.standard_normal((T, N, M, d)).astype(f),
        "u": rng.uniform(0, 1, (T, N)).astype(f),
    }


DEVICE = "cpu"  # tests/test_gpu_*.py switch this to "cuda" to drive the HIP engine


def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(DEVICE)


def _obs(inp, t, n, dark=False):
    return {
        "image": _t(inp["image_dark" if dark else "image"][t, :n]),
        "gripper_pos": _t(inp["gripper_pos"][t, :n]),
        "gripper_sensors": _t(inp["gripper_sensors"][t, :n]),
    }


def _np(x):
    return x.detach().cpu().numpy().astype(np.float32)


# ------------------------------------------------------------------ runners
def run_dynamics(model, inp, n, m):
    x = _t(inp["particles"][:n, :m]).reshape(n * m, -1)
    u = _t(inp["controls"][0, :n]).repeat_interleave(m, dim=0)
    y, L = model(initial_states=x, controls=u)
    return {"states": _np(y), "scale_trils": _np(L)}


def run_pf_measurement(model, inp, n, m, dark=False):
    out = {}
    for t in (0, 1):
        ll = model(states=_t(inp["particles"][:n, :m]), observations=_obs(inp, t, n, dark))
        out[f"loglik_t{t}"] = _np(ll)
    return out


def run_pf_measurement_masked(model, inp, n, m):
    out = {}
    for mask in ([True, True], [True, False], [False, True]):
        model.enabled_models = mask
        ll = model(states=_t(inp["particles"][:n, :m]), observations=_obs(inp, 0, n))
        out["loglik_" + "".join("1" if b else "0" for b in mask)] = _np(ll)
    model.enabled_models = [True, True]
    return out


def run_obs_only(model, inp, n, m, dark=False):
    out = {}
    for t in (0, 1):
        r = model(observations=_obs(inp, t, n, dark))
        if isinstance(r, tuple):
            out[f"mean_t{t}"], out[f"tril_t{t}"] = _np(r[0]), _np(r[1])
        else:
            out[f"out_t{t}"] = _np(r)
    return out


def run_obs_fixed_noise(model, inp, n, m):
    """Virtual sensor with ``noise_R_tril`` set: a fixed ``(N, d)`` diagonal replaces the learned
    ``r`` head (``door_models/kf.py:36-37,111-115``)."""
    model.noise_R_tril = _t(np.abs(inp["states0"][:n]) + np.float32(0.1))
    return run_obs_only(model, inp, n, m)


def run_encoder(model, inp, n, m):
    return {"feat": _np(model(_t(inp["image"][0, :n])[:, None]))}


def run_vector_encoder(model, inp, n, m):
    return {"feat": _np(model(_t(inp["particles"][:n, :m])))}


def run_filter_steps(model, inp, n, m, dark=False, measurement_init=False, masks=None):
    """``initialize_beliefs`` + ``T_STEPS - 1`` steps, as ``eval_helpers.py:125-142`` does."""
    model.eval()
    d = inp["states0"].shape[1]
    out = {}
    with torch.no_grad():
        if hasattr(model, "noise"):  # particle filters: replay identical randomness
            model.num_particles = m
            model.noise = ReplayNoise(
                gaussians=[_t(inp["eps_init"][:n, :m])]
                + [_t(inp["eps"][t, :n, :m]) for t in range(1, T_STEPS)],
                uniforms=[_t(inp["u"][t, :n]) for t in range(1, T_STEPS)],
            )
        if masks is not None:
            model.enabled_models = masks
        if measurement_init:
            model.measurement_initialize_beliefs(_obs(inp, 0, n, dark))
        else:
            cov = (torch.eye(d, device=DEVICE) * 0.1)[None].expand(n, d, d)
            model.initialize_beliefs(mean=_t(inp["states0"][:n]), covariance=cov)
        ests = []
        for t in range(1, T_STEPS):
            ests.append(model(observations=_obs(inp, t, n, dark), controls=_t(inp["controls"][t, :n])))
        out["estimates"] = _np(torch.stack(ests))
        if hasattr(model, "particle_states"):
            out["particle_states"] = _np(model.particle_states)
            out["particle_log_weights"] = _np(model.particle_log_weights)
        if getattr(model, "weighted_covariances", None) is not None:
            out["weighted_covariances"] = _np(model.weighted_covariances)
        if getattr(model, "_belief_covariance", None) is not None:
            out["belief_covariance"] = _np(model._belief_covariance)
        if hasattr(model, "filter_models"):
            for i, f in enumerate(model.filter_models):
                if f._belief_covariance is not None:
                    out[f"sub{i}_mean"] = _np(f._belief_mean)
                    out[f"sub{i}_cov"] = _np(f._belief_covariance)
    if masks is not None:
        model.enabled_models = [True] * len(masks)
    return out


def run_jacobian(model, inp, n, m):
    J = model.jacobian(initial_states=_t(inp["states0"][:n]), controls=_t(inp["controls"][0, :n]))
    return {"jacobian": _np(J)}


class Case(NamedTuple):
    name: str
    ref: Callable  # (crossmodal module, TaskSpec) -> reference module; capture_golden only
    make: Callable[[om.TaskSpec], torch.nn.Module]
    run: Callable
    kw: dict = {}
    tasks: tuple = ("door", "push")
    shapes: tuple = ((N_MAX, M_MAX), (1, 1))


def _pkg(cm, t):
    return getattr(cm, f"{t.name}_models")


def _cls(cm, t, suffix):
    return getattr(_pkg(cm, t), f"{t.name.capitalize()}{suffix}")


def _ref_sensor_pair(cm, t):
    V = _cls(cm, t, "VirtualSensorModel")
    return [V(modalities={"image"}), V(modalities={"pos", "sensors"})]


def _cases() -> List[Case]:
    c: List[Case] = []
    A = c.append
    # R1 dynamics (+ the default autograd Jacobian through it)
    A(Case("dynamics_ekf", lambda cm, t: _cls(cm, t, "DynamicsModel")(),
           lambda t: om.DynamicsModel(t), run_dynamics))
    A(Case("dynamics_pf_brent", lambda cm, t: cm.door_models.DoorDynamicsModelBrent(),
           lambda t: om.DynamicsModel(t, brent_noise=True), run_dynamics, tasks=("door",)))
    A(Case("dynamics_jacobian", lambda cm, t: _cls(cm, t, "DynamicsModel")(),
           lambda t: om.DynamicsModel(t), run_jacobian))
    # R6 small encoder
    A(Case("state_encoder", lambda cm, t: _pkg(cm, t).layers.state_layers(64),
           lambda t: om.vector_encoder(t.state_dim, 64), run_vector_encoder))
    # R5 image encoders
    A(Case("image_encoder", lambda cm, t: _pkg(cm, t).layers.observation_image_layers(64),
           lambda t: om.image_encoder(64), run_encoder))
    A(Case("image_encoder_spanning",
           lambda cm, t: cm.push_models.layers.observation_image_layers(64, spanning_avg_pool=True),
           lambda t: om.image_encoder(64, True), run_encoder, tasks=("push",)))
    # R2 per-modality measurement models; R7 virtual sensors
    for tag, mods in (("image", {"image"}), ("possens", {"pos", "sensors"}),
                      ("all", {"image", "pos", "sensors"})):
        A(Case(f"pf_measurement_{tag}",
               (lambda mods: lambda cm, t: _cls(cm, t, "MeasurementModel")(modalities=set(mods)))(mods),
               (lambda mods: lambda t: om.MeasurementModel(t, mods))(mods), run_pf_measurement))
        A(Case(f"virtual_sensor_{tag}",
               (lambda mods: lambda cm, t: _cls(cm, t, "VirtualSensorModel")(modalities=set(mods)))(mods),
               (lambda mods: lambda t: om.VirtualSensorModel(t, mods))(mods), run_obs_only))
    A(Case("virtual_sensor_fixed_noise",
           lambda cm, t: _cls(cm, t, "VirtualSensorModel")(),
           lambda t: om.VirtualSensorModel(t), run_obs_fixed_noise))
    # R4 weight model, blackout off/on
    A(Case("pf_weight_model",
           lambda cm, t: getattr(_pkg(cm, t).crossmodal_pf, f"{t.name.capitalize()}CrossmodalWeightModel")(know_image_blackout=False),
           lambda t: om.CrossmodalWeightModel(t, False), run_obs_only))
    A(Case("pf_weight_model_blackout",
           lambda cm, t: getattr(_pkg(cm, t).crossmodal_pf, f"{t.name.capitalize()}CrossmodalWeightModel")(know_image_blackout=True),
           lambda t: om.CrossmodalWeightModel(t, True), run_obs_only, {"dark": True}))
    # R3 crossmodal PF measurement: weighted / unweighted / masks / blackout
    A(Case("pf_crossmodal_measurement",
           lambda cm, t: _cls(cm, t, "CrossmodalParticleFilter")().measurement_model,
           lambda t: om.ParticleFilter(t, "crossmodal").measurement_model, run_pf_measurement))
    A(Case("pf_crossmodal_measurement_masks",
           lambda cm, t: _cls(cm, t, "CrossmodalParticleFilter")().measurement_model,
           lambda t: om.ParticleFilter(t, "crossmodal").measurement_model, run_pf_measurement_masked))
    A(Case("pf_crossmodal_measurement_blackout",
           lambda cm, t: _cls(cm, t, "CrossmodalParticleFilterSeq5")().measurement_model,
           lambda t: om.ParticleFilter(t, "crossmodal", True).measurement_model,
           run_pf_measurement, {"dark": True}))
    A(Case("pf_unimodal_measurement",
           lambda cm, t: _cls(cm, t, "UnimodalParticleFilter")().measurement_model,
           lambda t: om.ParticleFilter(t, "unimodal").measurement_model, run_pf_measurement))
    # R8 EKF weight model (reshape quirk Q3 preserved)
    A(Case("kf_weight_model",
           lambda cm, t: _cls(cm, t, "CrossmodalKalmanFilterWeightModel")(state_dim=t.state_dim),
           lambda t: om.CrossmodalKalmanFilterWeightModel(t), run_obs_only))
    # R11 fused virtual sensors
    A(Case("crossmodal_virtual_sensor",
           lambda cm, t: cm.base_models.CrossmodalVirtualSensorModel(
               virtual_sensor_model=_ref_sensor_pair(cm, t),
               crossmodal_weight_model=_cls(cm, t, "CrossmodalKalmanFilterWeightModel")(state_dim=t.state_dim),
               state_dim=t.state_dim),
           lambda t: om.CrossmodalVirtualSensorModel(t), run_obs_only))
    A(Case("unimodal_virtual_sensor",
           lambda cm, t: cm.base_models.UnimodalVirtualSensorModel(
               virtual_sensor_model=_ref_sensor_pair(cm, t), state_dim=t.state_dim),
           lambda t: om.UnimodalVirtualSensorModel(t), run_obs_only,
           shapes=((N_MAX, M_MAX),)))  # the reference asserts at N == 1 (squeeze(1), unimodal_kf.py:99-102)
    # T1+R1-R4 wired: whole particle filters (reference models on the restated recursion)
    for kind, suffix in (("single", "ParticleFilter"), ("crossmodal", "CrossmodalParticleFilter"),
                         ("unimodal", "UnimodalParticleFilter")):
        A(Case(f"filter_pf_{kind}",
               (lambda suffix: lambda cm, t: _cls(cm, t, suffix)())(suffix),
               (lambda kind: lambda t: om.ParticleFilter(t, kind))(kind), run_filter_steps))
    A(Case("filter_pf_crossmodal_seq5", lambda cm, t: _cls(cm, t, "CrossmodalParticleFilterSeq5")(),
           lambda t: om.ParticleFilter(t, "crossmodal", True), run_filter_steps, {"dark": True}))
    # T2+R7: single EKF; R9 crossmodal EKF (plain, blackout, masks, measurement init); R10
    A(Case("filter_kf", lambda cm, t: _cls(cm, t, "KalmanFilter")(),
           lambda t: om.KalmanFilter(t), run_filter_steps))
    A(Case("filter_kf_crossmodal", lambda cm, t: _cls(cm, t, "CrossmodalKalmanFilter")(),
           lambda t: om.CrossmodalKalmanFilter(t), run_filter_steps))
    A(Case("filter_kf_crossmodal_blackout",
           lambda cm, t: _cls(cm, t, "CrossmodalKalmanFilter")(know_image_blackout=True),
           lambda t: om.CrossmodalKalmanFilter(t, know_image_blackout=True),
           run_filter_steps, {"dark": True}))
    A(Case("filter_kf_crossmodal_masked", lambda cm, t: _cls(cm, t, "CrossmodalKalmanFilter")(),
           lambda t: om.CrossmodalKalmanFilter(t), run_filter_steps, {"masks": [False, True]}))
    A(Case("filter_kf_crossmodal_measinit", lambda cm, t: _cls(cm, t, "CrossmodalKalmanFilter")(),
           lambda t: om.CrossmodalKalmanFilter(t), run_filter_steps, {"measurement_init": True}))
    A(Case("filter_kf_unimodal", lambda cm, t: _cls(cm, t, "UnimodalKalmanFilter")(),
           lambda t: om.UnimodalKalmanFilter(t), run_filter_steps))
    A(Case("filter_kf_unimodal_masked", lambda cm, t: _cls(cm, t, "UnimodalKalmanFilter")(),
           lambda t: om.UnimodalKalmanFilter(t), run_filter_steps, {"masks": [True, False]}))
    # R11 wired (push variants cannot be constructed in the reference: Q8, Q9)
    A(Case("filter_kf_meas_crossmodal",
           lambda cm, t: cm.door_models.DoorMeasurementCrossmodalKalmanFilter(),
           lambda t: om.build("DoorMeasurementCrossmodalKalmanFilter"), run_filter_steps,
           tasks=("door",)))
    A(Case("filter_kf_meas_unimodal",
           lambda cm, t: cm.door_models.DoorMeasurementUnimodalKalmanFilter(),
           lambda t: om.build("DoorMeasurementUnimodalKalmanFilter"), run_filter_steps,
           tasks=("door",), shapes=((N_MAX, M_MAX),)))
    return c


CASES = _cases()


def case_key(case: Case, task: str, n: int, m: int) -> str:
    return f"{case.name}/{task}/n{n}m{m}"


def run_case(case: Case, model, task: om.TaskSpec, inp, n, m):
    sd = om.seeded_state_dict(model, seed=0, gain=WEIGHT_GAIN)
    model.load_state_dict(sd)
    model.to(DEVICE)
    return case.run(model, inp, n, m, **case.kw)


# ------------------------------------------------------------------ H1 (eval RMSE arithmetic)
def make_eval_inputs(task: om.TaskSpec, seed: int = 99):
    rng = np.random.RandomState(seed)
    T, N, d = 40, 3, task.state_dim
    true = rng.standard_normal((T, N, d)).astype(np.float32)
    pred = (true[1:] + 0.1 * rng.standard_normal((T - 1, N, d))).astype(np.float32)
    return true, pred
