"""CPU restatement of the reference's crossmodal layer.  TEST INFRASTRUCTURE (oracle/__init__.py).

One parametrised module covers both tasks (door ``d=3``, push ``d=2``) instead of the
reference's two parallel directories.  Every class cites the reference lines it
follows; ``state_dict`` keys equal the reference's so the seeded
weights used for the golden vectors load into either.  Pinned by
``tests/test_oracle_golden.py`` against vectors produced by the reference's own
``crossmodal`` package (``oracle/capture_golden.py``).

Quirks of the reference that change numbers are preserved by default and gated behind
explicit flags: ``fix_weight_layout`` (Q3), ``feedback`` (Q1).
"""
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import tf
from .fp.nn import resblocks


@dataclass(frozen=True)
class TaskSpec:
    name: str
    state_dim: int
    control_dim: int = 7
    pos_dim: int = 3
    sensors_dim: int = 7
    # dynamics noise, as variances (door_models/dynamics.py:20-23,85-88; push :17-20)
    q_var: Sequence[float] = ()
    pf_noise_brent: bool = False  # door PF uses sqrt(var)/8 as the diagonal
    vs_image_spanning_pool: bool = False  # push virtual sensor (push_models/kf.py:50-52)
    pf_weight_resblocks: int = 3  # door crossmodal_pf.py:64-72 vs push :64-70
    rmse_scale: Sequence[float] = ()  # eval_helpers.py:166,195


DOOR = TaskSpec(
    "door", 3, q_var=(0.05, 0.01, 0.01), pf_noise_brent=True,
    rmse_scale=(0.39479038, 0.05650279, 0.0565098),
)
PUSH = TaskSpec(
    "push", 2, q_var=(0.02, 0.02), vs_image_spanning_pool=True, pf_weight_resblocks=1,
    rmse_scale=(0.0572766, 0.06118315),
)
TASKS = {"door": DOOR, "push": PUSH}
MODALITIES = ("image", "pos", "sensors")


# ----------------------------------------------------------------------------- encoders
def vector_encoder(in_dim: int, units: int) -> nn.Sequential:
    """Linear -> ReLU -> ResLinear (``door_models/layers.py:11-24,27-40,66-79,82-95``)."""
    return nn.Sequential(nn.Linear(in_dim, units), nn.ReLU(), resblocks.Linear(units))


class DualSpanningAvgPool(nn.Module):
    """Full-height and full-width average pools, flattened and concatenated
    (``push_models/layers.py:43-65``)."""

    def __init__(self, rows: int, cols: int, reduce_size: int = 1):
        super().__init__()
        self.pool_h = nn.Sequential(nn.AvgPool2d((rows, reduce_size)), nn.Flatten())
        self.pool_w = nn.Sequential(nn.AvgPool2d((reduce_size, cols)), nn.Flatten())

    def forward(self, x):
        return torch.cat((self.pool_h(x), self.pool_w(x)), dim=-1)


def image_encoder(units: int, spanning_pool: bool = False) -> nn.Sequential:
    """32x32x1 -> ``units`` features (``door_models/layers.py:43-63``;
    ``push_models/layers.py:68-104``).  Module indices match the reference."""
    head = [
        nn.Conv2d(1, 32, 5, padding=2),
        nn.ReLU(),
        resblocks.Conv2d(32, kernel_size=3),
        nn.Conv2d(32, 16, 3, padding=1),
        nn.ReLU(),
    ]
    if spanning_pool:
        tail = [nn.Conv2d(16, 2, 3, padding=1), DualSpanningAvgPool(32, 32, 2), nn.Linear(64, units)]
    else:
        tail = [nn.Conv2d(16, 8, 3, padding=1), nn.Flatten(), nn.Linear(8 * 32 * 32, units)]
    return nn.Sequential(*head, *tail, nn.ReLU(), resblocks.Linear(units))


class _ObservationEncoders:
    """Mixin: per-modality observation encoders named as the reference names them."""

    def _build_encoders(self, task: TaskSpec, modalities, units: int, spanning_pool=False):
        unknown = set(modalities) - set(MODALITIES)
        assert not unknown and len(modalities) > 0, f"bad modalities {modalities}"
        self.modalities = set(modalities)
        if "image" in self.modalities:
            self.observation_image_layers = image_encoder(units, spanning_pool)
        if "pos" in self.modalities:
            self.observation_pos_layers = vector_encoder(task.pos_dim, units)
        if "sensors" in self.modalities:
            self.observation_sensors_layers = vector_encoder(task.sensors_dim, units)

    def encode_observations(self, observations: Dict[str, torch.Tensor]) -> torch.Tensor:
        """``(N, units * len(modalities))`` in image, pos, sensors order."""
        assert type(observations) == dict
        feats = []
        if "image" in self.modalities:
            feats.append(self.observation_image_layers(observations["image"][:, None, :, :]))
        if "pos" in self.modalities:
            feats.append(self.observation_pos_layers(observations["gripper_pos"]))
        if "sensors" in self.modalities:
            feats.append(self.observation_sensors_layers(observations["gripper_sensors"]))
        return torch.cat(feats, dim=1)


def image_blackout_rows(image: torch.Tensor) -> torch.Tensor:
    """Rows whose image is all-zero (``door_models/crossmodal_pf.py:99-103``)."""
    N = image.shape[0]
    return torch.sum(torch.abs(image.reshape(N, -1)), dim=1) < 1e-8


# ----------------------------------------------------------------------------- R1 dynamics
class DynamicsModel(tf.base.DynamicsModel):
    """``x' = x + dir(x,u) * sigmoid(gate(x,u))`` with constant noise
    (``door_models/dynamics.py:37-67,102-134``; ``push_models/dynamics.py:34-64``)."""

    def __init__(self, task: TaskSpec, *, brent_noise: bool = False, units: int = 64):
        super().__init__(state_dim=task.state_dim)
        var = torch.tensor(list(task.q_var), dtype=torch.float32)
        if brent_noise:
            self.Q_scale_tril_diag = nn.Parameter(torch.sqrt(var) / 8.0, requires_grad=False)
        else:
            self.Q_scale_tril = nn.Parameter(
                torch.linalg.cholesky(torch.diag(var)), requires_grad=False
            )
        self.state_layers = vector_encoder(task.state_dim, units)
        self.control_layers = vector_encoder(task.control_dim, units)
        self.shared_layers = nn.Sequential(
            nn.Linear(2 * units, units),
            resblocks.Linear(units),
            resblocks.Linear(units),
            resblocks.Linear(units),
            nn.Linear(units, task.state_dim + 1),
        )
        self.units = units

    def scale_tril(self) -> torch.Tensor:
        if hasattr(self, "Q_scale_tril_diag"):
            return torch.diag(self.Q_scale_tril_diag)
        return self.Q_scale_tril

    def forward(self, *, initial_states, controls):
        R, d = initial_states.shape[:2]
        assert d == self.state_dim
        merged = torch.cat(
            (self.control_layers(controls), self.state_layers(initial_states)), dim=-1
        )
        out = self.shared_layers(merged)
        update = out[..., :d] * torch.sigmoid(out[..., -1:])
        return initial_states + update, self.scale_tril()[None].expand(R, d, d)


# ----------------------------------------------------------------------------- R2 PF measurement
class MeasurementModel(tf.base.ParticleFilterMeasurementModel, _ObservationEncoders):
    """Per-particle log-likelihood MLP on ``[obs features | state features]``
    (``door_models/pf.py:30-107``; ``push_models/pf.py:30-109``)."""

    def __init__(self, task: TaskSpec, modalities=MODALITIES, units: int = 64):
        super().__init__(state_dim=task.state_dim)
        self._build_encoders(task, modalities, units)
        self.state_layers = vector_encoder(task.state_dim, units)
        self.shared_layers = nn.Sequential(
            nn.Linear(units * (1 + len(self.modalities)), units),
            nn.ReLU(),
            resblocks.Linear(units),
            resblocks.Linear(units),
            nn.Linear(units, 1),
        )
        self.units = units

    def forward(self, *, states, observations):
        assert states.dim() == 3 and states.shape[2] == self.state_dim
        N, M, _ = states.shape
        obs = self.encode_observations(observations)
        obs = obs[:, None, :].expand(N, M, obs.shape[1])
        merged = torch.cat((obs, self.state_layers(states)), dim=2)
        return self.shared_layers(merged).squeeze(2)


# ----------------------------------------------------------------------------- R4 PF weight model
class CrossmodalWeightModel(nn.Module, _ObservationEncoders):
    """obs -> ``(N, 2)`` modality log-weights (``door_models/crossmodal_pf.py:52-106``;
    ``push_models/crossmodal_pf.py:52-104``)."""

    def __init__(self, task: TaskSpec, know_image_blackout: bool, units: int = 64):
        super().__init__()
        self.modality_count = 2
        self.know_image_blackout = know_image_blackout
        self._build_encoders(task, MODALITIES, units)
        self.fusion_layers = nn.Sequential(
            nn.Linear(3 * units, units),
            nn.ReLU(),
            *[resblocks.Linear(units) for _ in range(task.pf_weight_resblocks)],
            nn.Linear(units, self.modality_count),
        )

    def forward(self, *, observations):
        out = self.fusion_layers(self.encode_observations(observations))
        if self.know_image_blackout:
            # Q10: -inf image log-weight on blacked-out frames
            dark = image_blackout_rows(observations["image"])
            out = out.clone()
            out[dark, 0] = -math.inf
        return out


# ----------------------------------------------------------------------------- R3 crossmodal PF fusion
class CrossmodalParticleFilterMeasurementModel(tf.base.ParticleFilterMeasurementModel):
    """``ll = logsumexp_k(log beta_k + ll_k)``; without a weight model
    ``logsumexp_k(ll_k)`` (``base_models/crossmodal_pf.py:87-141``)."""

    def __init__(self, *, measurement_models, crossmodal_weight_model, state_dim: int):
        super().__init__(state_dim=state_dim)
        self.measurement_models = nn.ModuleList(measurement_models)
        self.crossmodal_weight_model = crossmodal_weight_model
        self.enabled_models: List[bool] = [True] * len(self.measurement_models)

    def forward(self, *, states, observations):
        on = list(self.enabled_models)
        assert len(on) == len(self.measurement_models) and all(type(b) == bool for b in on)
        ll = torch.stack(
            [m(states=states, observations=observations)
             for m, keep in zip(self.measurement_models, on) if keep],
            dim=2,
        )
        if self.crossmodal_weight_model is not None:
            beta = self.crossmodal_weight_model(observations=observations)[:, on]
            ll = ll + beta[:, None, :]
        return torch.logsumexp(ll, dim=2)


# ----------------------------------------------------------------------------- R7 virtual sensor
class VirtualSensorModel(tf.base.VirtualSensorModel, _ObservationEncoders):
    """obs -> ``(z, sqrt(diag(r)^2 + 1e-6 I))`` (``door_models/kf.py:31-126``;
    ``push_models/kf.py:31-128``)."""

    def __init__(self, task: TaskSpec, modalities=MODALITIES, units: int = 64,
                 add_R_noise: float = 1e-6, noise_R_tril: torch.Tensor = None):
        super().__init__(state_dim=task.state_dim)
        d = task.state_dim
        self.noise_R_tril = noise_R_tril  # fixed (N, d) diagonal replacing the r head (kf.py:36-37,111-115)
        self._build_encoders(task, modalities, units, spanning_pool=task.vs_image_spanning_pool)
        self.shared_layers = nn.Sequential(
            nn.Linear(units * len(self.modalities), 2 * units),
            nn.ReLU(),
            resblocks.Linear(2 * units),
            resblocks.Linear(2 * units),
        )

        def head():
            return nn.Sequential(nn.Linear(units, d), nn.ReLU(), resblocks.Linear(d), nn.Linear(d, d))

        self.r_layer = head()
        self.z_layer = head()
        self.units = units
        self.add_R_noise = add_R_noise

    def forward(self, *, observations):
        shared = self.shared_layers(self.encode_observations(observations))
        z = self.z_layer(shared[:, : self.units])
        r_hat = self.r_layer(shared[:, self.units:]) if self.noise_R_tril is None else self.noise_R_tril
        R = torch.diag_embed(r_hat) ** 2
        assert R.shape == (z.shape[0], self.state_dim, self.state_dim)
        if self.add_R_noise > 0:
            R = R + self.add_R_noise * torch.eye(self.state_dim, dtype=R.dtype, device=R.device)
        return z, torch.sqrt(R)


# ----------------------------------------------------------------------------- R8 KF weight model
class CrossmodalKalmanFilterWeightModel(nn.Module, _ObservationEncoders):
    """obs -> per-state-dimension modality weights ``(2, N, d)``
    (``door_models/crossmodal_kf.py:101-167``; push identical).

    Q3: the reference *reshapes* ``(N, 2d)`` to ``(2, N, d)`` (``:158``), mixing batch and
    feature axes; ``fix_weight_layout=True`` does the intended ``view(N,2,d).permute``.
    """

    def __init__(self, task: TaskSpec, units: int = 64, fix_weight_layout: bool = False):
        super().__init__()
        self.modality_count = 2
        self.state_dim = task.state_dim
        self.fix_weight_layout = fix_weight_layout
        self._build_encoders(task, MODALITIES, units)
        self.fusion_layers = nn.Sequential(
            nn.Linear(3 * units, units),
            nn.ReLU(),
            resblocks.Linear(units),
            nn.Linear(units, self.modality_count * self.state_dim),
            nn.Sigmoid(),
        )

    def forward(self, *, observations):
        out = self.fusion_layers(self.encode_observations(observations))
        N = out.shape[0]
        if self.fix_weight_layout:
            w = out.view(N, self.modality_count, self.state_dim).permute(1, 0, 2)
        else:
            w = out.reshape(self.modality_count, N, self.state_dim)
        return w / (torch.sum(w, dim=0) + 1e-9)


# ----------------------------------------------------------------------------- R12
def weighted_average(predictions: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """``sum_k w_k p_k / (sum_k w_k + 1e-9)`` over dim 0 (``base_models/utility.py:4-11``)."""
    assert predictions.shape == weights.shape
    return torch.sum(weights / (torch.sum(weights, dim=0) + 1e-9) * predictions, dim=0)


def _fuse_crossmodal(weights, means, covs):
    """``mu = wavg``; ``Sigma = sum_k (w_k w_k^T) (.) Sigma_k``
    (``base_models/crossmodal_kf.py:153-167``)."""
    mu = weighted_average(means, weights)
    outer = weights[..., :, None] * weights[..., None, :]
    return mu, torch.sum(outer * covs, dim=0)


def _sensor_fusion_crossmodal(weights, means, covs):
    """``Sigma = prod_k prod_i w_{k,i} * sum_k Sigma_k``
    (``base_models/crossmodal_kf.py:224-235,343-352``)."""
    mu = weighted_average(means, weights)
    mult = torch.prod(torch.prod(weights, dim=-1), dim=0)[:, None, None]
    return mu, mult * torch.sum(covs, dim=0)


# ----------------------------------------------------------------------------- single EKF
class KalmanFilter(tf.filters.VirtualSensorExtendedKalmanFilter):
    """``DoorKalmanFilter`` / ``PushKalmanFilter`` (``door_models/kf.py:14-28``)."""

    def __init__(self, task: TaskSpec, dynamics_model=None, virtual_sensor_model=None):
        if dynamics_model is None and virtual_sensor_model is None:
            dynamics_model = DynamicsModel(task)
            virtual_sensor_model = VirtualSensorModel(task)
        super().__init__(dynamics_model=dynamics_model, virtual_sensor_model=virtual_sensor_model)


def _modal_filters(task: TaskSpec):
    return [
        KalmanFilter(task, DynamicsModel(task), VirtualSensorModel(task, {"image"})),
        KalmanFilter(task, DynamicsModel(task), VirtualSensorModel(task, {"pos", "sensors"})),
    ]


def _modal_sensors(task: TaskSpec):
    return [VirtualSensorModel(task, {"image"}), VirtualSensorModel(task, {"pos", "sensors"})]


# ----------------------------------------------------------------------------- R9 crossmodal EKF
class CrossmodalKalmanFilter(tf.base.Filter):
    """K sub-EKFs fused by learned weights (``base_models/crossmodal_kf.py:39-240``) with
    the blackout override of ``door_models/crossmodal_kf.py:43-98``.

    ``feedback`` (Q1): the reference writes the fused belief to ``f.states_prev`` /
    ``f.states_covariance_prev`` (``:147-149``) but the sub-filters keep their belief in
    ``_belief_mean/_belief_covariance`` (read at ``:180``), so the write is inert:
    ``"none"`` (default) reproduces that; ``"belief"`` performs the intended write-back.
    """

    def __init__(self, task: TaskSpec, *, know_image_blackout: bool = False,
                 feedback: str = "none", fix_weight_layout: bool = False):
        super().__init__(state_dim=task.state_dim)
        self.filter_models = nn.ModuleList(_modal_filters(task))
        self.crossmodal_weight_model = CrossmodalKalmanFilterWeightModel(
            task, fix_weight_layout=fix_weight_layout)
        self.enabled_models: List[bool] = [True] * len(self.filter_models)
        self.know_image_blackout = know_image_blackout
        assert feedback in ("none", "belief")
        self.feedback = feedback
        self.weighted_covariances = None

    @property
    def state_covariance_estimate(self):
        return self.weighted_covariances

    def initialize_beliefs(self, *, mean, covariance):
        for f in self.filter_models:
            f.initialize_beliefs(mean=mean, covariance=covariance)

    def calculate_unimodal_states(self, observations, controls):
        live = [f for f, keep in zip(self.filter_models, self.enabled_models) if keep]
        means = torch.stack([f(observations=observations, controls=controls) for f in live])
        covs = torch.stack([f._belief_covariance for f in live])
        return means, covs

    def calculate_weighted_states(self, state_weights, unimodal_states, unimodal_covariances):
        return _fuse_crossmodal(state_weights, unimodal_states, unimodal_covariances)

    def _plain_forward(self, observations, controls):
        N = controls.shape[0]
        on = list(self.enabled_models)
        means, covs = self.calculate_unimodal_states(observations, controls)
        if sum(on) < len(on):
            w = torch.tensor(on, dtype=means.dtype, device=means.device)
            w = w[:, None, None].repeat(1, N, self.state_dim)
        else:
            w = self.crossmodal_weight_model(observations=observations)
        w = w[on]
        mu, Sigma = _fuse_crossmodal(w, means, covs)
        self.weighted_covariances = Sigma
        for f in self.filter_models:
            f.states_prev = mu  # inert, as in the reference (Q1)
            f.states_covariance_prev = Sigma
            if self.feedback == "belief":
                f._belief_mean, f._belief_covariance = mu, Sigma
        return mu

    def forward(self, *, observations, controls):
        if not self.know_image_blackout:
            return self._plain_forward(observations, controls)
        dark = image_blackout_rows(observations["image"])
        on = list(self.enabled_models)
        if int(dark.sum()) == 0 or sum(on) < len(on):  # batch-global test
            return self._plain_forward(observations, controls)
        # Q2: this branch skips the write-back and broadcasts (N,1) masks over (N,d)
        means, covs = self.calculate_unimodal_states(observations, controls)
        raw = self.crossmodal_weight_model(observations=observations)
        keep = (~dark).to(raw.dtype)[:, None]
        img_w = dark.to(raw.dtype)[:, None] * 1e-9 + keep * raw[0]
        oth_w = dark.to(raw.dtype)[:, None] * (1.0 - 1e-9) + keep * raw[1]
        mu, Sigma = _fuse_crossmodal(torch.stack([img_w, oth_w]), means, covs)
        self.weighted_covariances = Sigma
        return mu

    def measurement_initialize_beliefs(self, observations):
        """``base_models/crossmodal_kf.py:208-240``."""
        on = list(self.enabled_models)
        outs = [f.virtual_sensor_model(observations=observations)
                for f, keep in zip(self.filter_models, on) if keep]
        means = torch.stack([o[0] for o in outs])
        trils = torch.stack([o[1] for o in outs])
        w = self.crossmodal_weight_model(observations=observations)[on]
        mu, Sigma = _sensor_fusion_crossmodal(w, means, trils @ trils.transpose(-1, -2))
        self.initialize_beliefs(mean=mu, covariance=Sigma)


# ----------------------------------------------------------------------------- R10 unimodal EKF
class UnimodalKalmanFilter(tf.base.Filter):
    """Information-form fusion ``Sigma = (sum_k (Sigma_k+1e-9)^-1 + 1e-9)^-1``,
    ``mu = Sigma sum_k P_k mu_k`` (``base_models/unimodal_kf.py:118-270``).  Q6: the
    reference neither stores the fused covariance nor feeds the fused state back."""

    def __init__(self, task: TaskSpec):
        super().__init__(state_dim=task.state_dim)
        self.filter_models = nn.ModuleList(_modal_filters(task))
        self.enabled_models: List[bool] = [True] * len(self.filter_models)
        self.weighted_covariances = None

    @property
    def state_covariance_estimate(self):
        return self.weighted_covariances

    def initialize_beliefs(self, *, mean, covariance):
        for f in self.filter_models:
            f.initialize_beliefs(mean=mean, covariance=covariance)

    def forward(self, *, observations, controls):
        live = [f for f, keep in zip(self.filter_models, self.enabled_models) if keep]
        means = torch.stack([f(observations=observations, controls=controls) for f in live])
        if len(live) == 1:
            return means[0]
        prec = torch.stack([torch.inverse(f._belief_covariance + 1e-9) for f in live])
        Sigma = torch.inverse(torch.sum(prec, dim=0) + 1e-9)
        info = torch.sum(prec @ means[..., None], dim=0)
        return (Sigma @ info).squeeze(-1)


# ----------------------------------------------------------------------------- R11 fused sensors
class CrossmodalVirtualSensorModel(tf.base.VirtualSensorModel):
    """Fuse K virtual sensors before one EKF; returns a Cholesky factor
    (``base_models/crossmodal_kf.py:243-359``)."""

    def __init__(self, task: TaskSpec, fix_weight_layout: bool = False):
        super().__init__(state_dim=task.state_dim)
        self.virtual_sensor_model = nn.ModuleList(_modal_sensors(task))
        self.crossmodal_weight_model = CrossmodalKalmanFilterWeightModel(
            task, fix_weight_layout=fix_weight_layout)
        self.enabled_models: List[bool] = [True] * len(self.virtual_sensor_model)

    def forward(self, *, observations):
        on = list(self.enabled_models)
        outs = [m(observations=observations)
                for m, keep in zip(self.virtual_sensor_model, on) if keep]
        means = torch.stack([o[0] for o in outs])
        trils = torch.stack([o[1] for o in outs])
        N = means.shape[1]
        if sum(on) < len(on):
            w = torch.tensor(on, dtype=means.dtype, device=means.device)
            w = w[:, None, None].repeat(1, N, self.state_dim)
        else:
            w = self.crossmodal_weight_model(observations=observations)
        mu, Sigma = _sensor_fusion_crossmodal(w[on], means, trils @ trils.transpose(-1, -2))
        return mu, torch.linalg.cholesky(Sigma)


class UnimodalVirtualSensorModel(tf.base.VirtualSensorModel):
    """``base_models/unimodal_kf.py:13-115``.  Q5: "precision" is the element-wise
    ``1/(scale_tril + 1e-9)`` (off-diagonals become 1e9) and a covariance is returned
    where a scale-tril is expected -- preserved."""

    def __init__(self, task: TaskSpec):
        super().__init__(state_dim=task.state_dim)
        self.virtual_sensor_model = nn.ModuleList(_modal_sensors(task))
        self.enabled_models: List[bool] = [True] * len(self.virtual_sensor_model)

    def forward(self, *, observations):
        outs = [m(observations=observations)
                for m, keep in zip(self.virtual_sensor_model, self.enabled_models) if keep]
        means = torch.stack([o[0] for o in outs])
        trils = torch.stack([o[1] for o in outs])
        if len(outs) == 1:
            return means[0], (trils @ trils.transpose(-1, -2))[0]
        prec = 1.0 / (trils + 1e-9)
        w = torch.diagonal(prec, dim1=-2, dim2=-1)
        return weighted_average(means, w), torch.inverse(torch.sum(prec, dim=0) + 1e-9)


# ----------------------------------------------------------------------------- particle filters
class ParticleFilter(tf.filters.ParticleFilter):
    """The reference's PF classes: fixed model wiring plus the train/eval particle-count
    switch 30 <-> 300 (``door_models/pf.py:14-27``, ``crossmodal_pf.py:18-40``,
    ``unimodal_pf.py:9-29``; push mirrors).  ``train()`` returns ``self`` (the
    reference's override returns ``None``, Q7)."""

    TRAIN_PARTICLES = 30
    EVAL_PARTICLES = 300

    def __init__(self, task: TaskSpec, kind: str, know_image_blackout: bool = False,
                 resample_mode: str = "systematic"):
        dyn = DynamicsModel(task, brent_noise=task.pf_noise_brent)
        if kind == "single":
            meas = MeasurementModel(task)
        else:
            weight = (CrossmodalWeightModel(task, know_image_blackout)
                      if kind == "crossmodal" else None)
            assert kind in ("crossmodal", "unimodal")
            meas = CrossmodalParticleFilterMeasurementModel(
                measurement_models=[MeasurementModel(task, {"image"}),
                                    MeasurementModel(task, {"pos", "sensors"})],
                crossmodal_weight_model=weight,
                state_dim=task.state_dim,
            )
        super().__init__(dynamics_model=dyn, measurement_model=meas,
                         num_particles=self.TRAIN_PARTICLES, resample_mode=resample_mode)

    def train(self, mode: bool = True):
        self.num_particles = self.TRAIN_PARTICLES if mode else self.EVAL_PARTICLES
        return super().train(mode)


# ----------------------------------------------------------------------------- registry
def build(name: str, **kw) -> tf.base.Filter:
    """Construct a filter by the reference's class name (``door_models/__init__.py:5-19``,
    ``push_models/__init__.py:5-21``); LSTM baselines are out of scope."""
    for prefix, task in (("Door", DOOR), ("Push", PUSH)):
        if name.startswith(prefix):
            kind = name[len(prefix):]
            break
    else:
        raise KeyError(name)
    if kind == "ParticleFilter":
        return ParticleFilter(task, "single", **kw)
    if kind == "CrossmodalParticleFilter":
        return ParticleFilter(task, "crossmodal", **kw)
    if kind == "CrossmodalParticleFilterSeq5":
        return ParticleFilter(task, "crossmodal", know_image_blackout=True, **kw)
    if kind == "UnimodalParticleFilter":
        return ParticleFilter(task, "unimodal", **kw)
    if kind == "KalmanFilter":
        return KalmanFilter(task)
    if kind == "CrossmodalKalmanFilter":
        return CrossmodalKalmanFilter(task, **kw)
    if kind == "UnimodalKalmanFilter":
        return UnimodalKalmanFilter(task)
    if kind == "MeasurementCrossmodalKalmanFilter":
        return KalmanFilter(task, DynamicsModel(task), CrossmodalVirtualSensorModel(task, **kw))
    if kind == "MeasurementUnimodalKalmanFilter":
        return KalmanFilter(task, DynamicsModel(task), UnimodalVirtualSensorModel(task))
    raise KeyError(name)


MODEL_NAMES = [
    f"{p}{k}" for p in ("Door", "Push") for k in (
        "ParticleFilter", "CrossmodalParticleFilter", "CrossmodalParticleFilterSeq5",
        "UnimodalParticleFilter", "KalmanFilter", "CrossmodalKalmanFilter",
        "UnimodalKalmanFilter", "MeasurementCrossmodalKalmanFilter",
        "MeasurementUnimodalKalmanFilter")
]


def seeded_state_dict(module: nn.Module, seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic weights *by key name*, so the same tensors load into the reference's
    modules, this oracle's and the HIP engine's regardless of construction order.
    Fixed (``requires_grad=False``) noise parameters keep their constructed values."""
    import zlib

    out = {}
    for key, ref in module.state_dict().items():
        if key.rsplit(".", 1)[-1].startswith("Q_scale_tril"):
            out[key] = ref.clone()
            continue
        rng = np.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2 ** 31))
        if ref.dim() >= 2:
            fan_in = int(np.prod(ref.shape[1:]))
            w = rng.standard_normal(ref.shape) * gain / math.sqrt(fan_in)
        else:
            w = rng.standard_normal(ref.shape) * 0.1
        out[key] = torch.from_numpy(w.astype(np.float32))
    return out
