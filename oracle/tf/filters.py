"""Particle filter and (virtual-sensor) EKF recursions -- torchfilter.filters restated.

Follows the upstream step order recorded in SURVEY.md A.2 / 3.2 / 3.3 and the
reference's use of the classes: ``ParticleFilter(dynamics_model=, measurement_model=,
num_particles=)`` with a public mutable ``num_particles`` (``door_models/pf.py:18-27``);
``VirtualSensorExtendedKalmanFilter(dynamics_model=, virtual_sensor_model=)`` whose
belief is read back as ``_belief_covariance`` (``base_models/crossmodal_kf.py:180``).
Every random draw is an explicit tensor from ``self.noise`` (a ``NoiseSource``).
"""
import math

import numpy as np
import torch

from .. import resample as _rs
from ..fp.utils import SliceWrapper
from . import base


class ParticleFilter(base.Filter):
    def __init__(
        self,
        *,
        dynamics_model: base.DynamicsModel,
        measurement_model: base.ParticleFilterMeasurementModel,
        num_particles: int = 100,
        resample: bool = None,
        resample_mode: str = "systematic",
        estimation_method: str = "weighted_average",
        soft_resample_alpha: float = 1.0,
    ):
        super().__init__(state_dim=dynamics_model.state_dim)
        assert measurement_model.state_dim == self.state_dim
        assert 0.0 < soft_resample_alpha <= 1.0
        self.soft_resample_alpha = soft_resample_alpha
        self.dynamics_model = dynamics_model
        self.measurement_model = measurement_model
        self.num_particles = num_particles
        self.resample = resample  # None => resample iff not self.training
        assert resample_mode in ("systematic", "multinomial")
        self.resample_mode = resample_mode
        assert estimation_method in ("weighted_average", "argmax")
        self.estimation_method = estimation_method
        self.noise = base.NoiseSource(0)
        self.particle_states: torch.Tensor = None  # (N, M, d)
        self.particle_log_weights: torch.Tensor = None  # (N, M)
        self.last_resample_indices = None
        self._initialized = False

    def initialize_beliefs(self, *, mean, covariance):
        N, d = mean.shape
        assert d == self.state_dim and covariance.shape == (N, d, d)
        M = self.num_particles
        eps = self.noise.gaussian((N, M, d), like=mean)
        L = torch.linalg.cholesky(covariance)
        self.particle_states = mean[:, None, :] + torch.einsum("nij,nmj->nmi", L, eps)
        self.particle_log_weights = mean.new_full((N, M), -math.log(M))
        self._initialized = True

    def forward(self, *, observations, controls):
        assert self._initialized, "initialize_beliefs() first"
        N, M, d = self.particle_states.shape
        do_resample = (not self.training) if self.resample is None else self.resample
        if not do_resample and self.num_particles != M:
            self._adapt_particle_count()
            N, M, d = self.particle_states.shape

        # propagate: same control for each of a trajectory's M particles
        flat = self.particle_states.reshape(N * M, d)
        rep_controls = SliceWrapper(controls).map(
            lambda t: torch.repeat_interleave(t, repeats=M, dim=0)
        )
        pred, tril = self.dynamics_model(initial_states=flat, controls=rep_controls)
        eps = self.noise.gaussian((N, M, d), like=pred).reshape(N * M, d)
        self.particle_states = (pred + torch.einsum("rij,rj->ri", tril, eps)).reshape(N, M, d)

        # reweight + normalise
        loglik = self.measurement_model(states=self.particle_states, observations=observations)
        assert loglik.shape == (N, M)
        logw = self.particle_log_weights + loglik
        logw = logw - torch.logsumexp(logw, dim=1, keepdim=True)
        self.particle_log_weights = logw

        if self.estimation_method == "weighted_average":
            estimate = torch.sum(torch.exp(logw)[:, :, None] * self.particle_states, dim=1)
        else:
            best = torch.argmax(logw, dim=1)
            estimate = self.particle_states[torch.arange(N), best]

        if do_resample:
            self._resample()
        return estimate

    def _adapt_particle_count(self):
        """Upstream's adaptation when a non-resampling step finds ``num_particles != M``
        (SURVEY.md A.2): whole copies of the set first, then a sample without replacement shared
        by the batch; weights gathered and re-normalised.  Upstream draws the permutation with
        ``torch.randperm`` (global RNG); here it is the arg-sort of ``M`` explicit uniforms."""
        N, M, d = self.particle_states.shape
        Mo = int(self.num_particles)
        copies = (Mo // M) * M
        parts = []
        if copies > 0:
            parts.append(torch.arange(M).repeat(copies // M))
        if Mo - copies > 0:
            u = self.noise.uniform((M,), like=self.particle_log_weights)
            parts.append(torch.argsort(u, stable=True)[: Mo - copies])
        idx = torch.cat(parts)[None, :].expand(N, Mo)
        self.particle_states = torch.gather(self.particle_states, 1, idx[:, :, None].expand(N, Mo, d))
        lw = torch.gather(self.particle_log_weights, 1, idx)
        self.particle_log_weights = lw - torch.logsumexp(lw, dim=1, keepdim=True)

    def _resample(self):
        N, M, d = self.particle_states.shape
        Mo = self.num_particles
        like = self.particle_log_weights
        if self.resample_mode == "systematic":
            u = self.noise.uniform((N,), like=like)
        else:
            u = self.noise.uniform((N, Mo), like=like)
        alpha = float(self.soft_resample_alpha)
        idx = _rs.resample_indices(
            like.detach().cpu().numpy(), u.detach().cpu().numpy(), self.resample_mode, Mo, alpha
        )
        idx_t = torch.from_numpy(idx.astype(np.int64)).to(like.device)
        self.last_resample_indices = idx_t
        self.particle_states = torch.gather(
            self.particle_states, 1, idx_t[:, :, None].expand(N, Mo, d)
        )
        if alpha < 1.0:
            # upstream ``_resample``: survivors keep ``logw - log(mixture)``, re-normalised -- the
            # importance weights that make the soft draw unbiased (and differentiable upstream)
            lw = like - torch.logsumexp(like, dim=1, keepdim=True)
            a = float(np.float32(alpha))
            mix = torch.logaddexp(lw + math.log(a), torch.full_like(lw, math.log((1.0 - a) / M)))
            new = torch.gather(lw - mix, 1, idx_t)
            self.particle_log_weights = new - torch.logsumexp(new, dim=1, keepdim=True)
        else:
            self.particle_log_weights = like.new_full((N, Mo), -math.log(Mo))


class _IdentityMeasurementModel(base.KalmanFilterMeasurementModel):
    """``C = I``: the virtual sensor already speaks state space."""

    def __init__(self, *, state_dim: int):
        super().__init__(state_dim=state_dim, observation_dim=state_dim)
        self.scale_tril = None

    def forward(self, *, states):
        return states, self.scale_tril

    def jacobian(self, *, states):
        N, d = states.shape
        return torch.eye(d, dtype=states.dtype, device=states.device)[None].expand(N, d, d)


class ExtendedKalmanFilter(base.Filter):
    """predict ``S- = A S A^T + L L^T`` / correct ``K = S- C^T (C S- C^T + R)^-1``,
    ``S = (I - K C) S-`` -- no Joseph form, no symmetrisation."""

    def __init__(self, *, dynamics_model, measurement_model):
        super().__init__(state_dim=dynamics_model.state_dim)
        self.dynamics_model = dynamics_model
        self.measurement_model = measurement_model
        self._belief_mean = None
        self._belief_covariance = None
        self._initialized = False

    @property
    def belief_mean(self):
        return self._belief_mean

    @belief_mean.setter
    def belief_mean(self, v):
        self._belief_mean = v

    @property
    def belief_covariance(self):
        return self._belief_covariance

    @belief_covariance.setter
    def belief_covariance(self, v):
        self._belief_covariance = v

    def initialize_beliefs(self, *, mean, covariance):
        N, d = mean.shape
        assert d == self.state_dim and covariance.shape == (N, d, d)
        self._belief_mean = mean
        self._belief_covariance = covariance
        self._initialized = True

    def forward(self, *, observations, controls):
        assert self._initialized, "initialize_beliefs() first"
        self._predict_step(controls=controls)
        self._update_step(observations=observations)
        return self._belief_mean

    def _predict_step(self, *, controls):
        mu, Sigma = self._belief_mean, self._belief_covariance
        mu_pred, L = self.dynamics_model(initial_states=mu, controls=controls)
        A = self.dynamics_model.jacobian(initial_states=mu, controls=controls)
        self._belief_mean = mu_pred
        self._belief_covariance = A @ Sigma @ A.transpose(-1, -2) + L @ L.transpose(-1, -2)

    def _update_step(self, *, observations):
        mu, Sigma = self._belief_mean, self._belief_covariance
        y_hat, Rtril = self.measurement_model(states=mu)
        C = self.measurement_model.jacobian(states=mu)
        R = Rtril @ Rtril.transpose(-1, -2)
        S = C @ Sigma @ C.transpose(-1, -2) + R
        K = Sigma @ C.transpose(-1, -2) @ torch.inverse(S)
        self._belief_mean = mu + (K @ (observations - y_hat)[:, :, None]).squeeze(-1)
        eye = torch.eye(K.shape[-1], dtype=K.dtype, device=K.device)
        self._belief_covariance = (eye - K @ C) @ Sigma


class VirtualSensorExtendedKalmanFilter(ExtendedKalmanFilter):
    def __init__(self, *, dynamics_model, virtual_sensor_model):
        ident = _IdentityMeasurementModel(state_dim=dynamics_model.state_dim)
        super().__init__(dynamics_model=dynamics_model, measurement_model=ident)
        self.virtual_sensor_model = virtual_sensor_model

    def forward(self, *, observations, controls):
        z, Rtril = self.virtual_sensor_model(observations=observations)
        self.measurement_model.scale_tril = Rtril
        return super().forward(observations=z, controls=controls)


# ------------------------------------------------------------------------------ unscented filter
class JulierSigmaPointStrategy:
    def __init__(self, lambd=None):
        self.lambd = lambd

    def compute_lambda(self, dim):
        return 3.0 - dim if self.lambd is None else float(self.lambd)

    def compute_sigma_weights(self, dim):
        lambd = self.compute_lambda(dim)
        wm = torch.full((2 * dim + 1,), 1.0 / (2.0 * (dim + lambd)))
        wm[0] = lambd / (dim + lambd)
        return wm.clone(), wm  # (weights_c, weights_m)


class MerweSigmaPointStrategy:
    def __init__(self, alpha=1e-2, beta=2.0, kappa=None):
        self.alpha, self.beta, self.kappa = alpha, beta, kappa

    def compute_lambda(self, dim):
        kappa = 3.0 - dim if self.kappa is None else float(self.kappa)
        return self.alpha ** 2 * (dim + kappa) - dim

    def compute_sigma_weights(self, dim):
        lambd = self.compute_lambda(dim)
        wm = torch.full((2 * dim + 1,), 1.0 / (2.0 * (dim + lambd)))
        wm[0] = lambd / (dim + lambd)
        wc = wm.clone()
        wc[0] = wm[0] + 1.0 - self.alpha ** 2 + self.beta
        return wc, wm


def sigma_points(mean, covariance, lambd):
    """``(N, 2d+1, d)``: the mean, then ``mean +/- sqrt(d + lambda) chol(covariance)[:, i]``."""
    N, d = mean.shape
    L = torch.linalg.cholesky(covariance) * math.sqrt(d + lambd)
    cols = L.transpose(-1, -2)  # row i = column i of L
    return torch.cat([mean[:, None, :], mean[:, None, :] + cols, mean[:, None, :] - cols], dim=1)


class VirtualSensorUnscentedKalmanFilter(ExtendedKalmanFilter):
    """Upstream's UKF with a virtual sensor, written in its GENERAL form (unscented transform
    through the dynamics, then through the measurement model -- here the identity, ``C = I``):
    predict ``mu- = sum wm X'``, ``Sigma- = sum wc (X' - mu-)(..)^T + L L^T`` (noise at the belief
    mean); update with fresh sigma points of ``(mu-, Sigma-)``: ``P_yy = sum wc (Y - y)(..)^T + R``,
    ``P_xy = sum wc (X - mu-)(Y - y)^T``, ``K = P_xy P_yy^-1``."""

    def __init__(self, *, dynamics_model, virtual_sensor_model, sigma_point_strategy=None):
        ident = _IdentityMeasurementModel(state_dim=dynamics_model.state_dim)
        super().__init__(dynamics_model=dynamics_model, measurement_model=ident)
        self.virtual_sensor_model = virtual_sensor_model
        self.sigma_point_strategy = sigma_point_strategy or JulierSigmaPointStrategy()

    def forward(self, *, observations, controls):
        assert self._initialized, "initialize_beliefs() first"
        z, Rtril = self.virtual_sensor_model(observations=observations)
        mu, Sigma = self._belief_mean, self._belief_covariance
        N, d = mu.shape
        P = 2 * d + 1
        lambd = self.sigma_point_strategy.compute_lambda(d)
        wc, wm = self.sigma_point_strategy.compute_sigma_weights(d)
        X = sigma_points(mu, Sigma, lambd)
        rep = SliceWrapper(controls).map(lambda t: torch.repeat_interleave(t, repeats=P, dim=0))
        Xp = self.dynamics_model(initial_states=X.reshape(N * P, d), controls=rep)[0].reshape(N, P, d)
        _, L = self.dynamics_model(initial_states=mu, controls=controls)
        mu_p = torch.einsum("p,npi->ni", wm, Xp)
        e = Xp - mu_p[:, None, :]
        S_p = torch.einsum("p,npi,npj->nij", wc, e, e) + L @ L.transpose(-1, -2)
        # measurement update through the (identity) measurement model
        Xs = sigma_points(mu_p, S_p, lambd)
        Y = Xs
        y = torch.einsum("p,npi->ni", wm, Y)
        ey, ex = Y - y[:, None, :], Xs - mu_p[:, None, :]
        Pyy = torch.einsum("p,npi,npj->nij", wc, ey, ey) + Rtril @ Rtril.transpose(-1, -2)
        Pxy = torch.einsum("p,npi,npj->nij", wc, ex, ey)
        K = Pxy @ torch.inverse(Pyy)
        self._belief_mean = mu_p + (K @ (z - y)[:, :, None]).squeeze(-1)
        self._belief_covariance = S_p - K @ Pyy @ K.transpose(-1, -2)
        return self._belief_mean
