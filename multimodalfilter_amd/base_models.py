"""Multimodal fusion classes -- the API of ``/root/reference/crossmodal/base_models`` with the
fusion arithmetic in HIP.

* ``CrossmodalParticleFilterMeasurementModel`` (``base_models/crossmodal_pf.py:33-141``):
  ``logsumexp_k(log beta_k + ll_k)`` is folded into the epilogue of the measurement kernel
  (``mmf_pf_measure`` with ``combine``), so no ``(N, M, K)`` tensor is ever materialised.
* ``CrossmodalKalmanFilter`` (``base_models/crossmodal_kf.py:39-240``) and
  ``UnimodalKalmanFilter`` (``base_models/unimodal_kf.py:118-270``): all sub-filters'
  predict / correct steps *and* the fusion of their beliefs are ONE ``mmf_ekf_step`` launch.
* ``CrossmodalVirtualSensorModel`` / ``UnimodalVirtualSensorModel``
  (``crossmodal_kf.py:243-359``, ``unimodal_kf.py:13-115``): per-trajectory (N rows) and
  off the per-particle path; evaluated with device-side torch ops.

Reference quirks that change numbers are preserved and flag-gated.
"""
import abc
import ctypes
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import _abi, base, engine, filters
from .engine import call_with_image_feat, encode_observation_images, use_autograd
from .utils import tree_index, tree_leading_shape, tree_map


def _check_mask(mask, n):
    assert isinstance(mask, list)
    assert len(mask) == n
    for x in mask:
        assert type(x) == bool


class _EnabledModels:
    @property
    def enabled_models(self) -> List[bool]:
        return self._enabled_models

    @enabled_models.setter
    def enabled_models(self, enabled_models: List[bool]) -> None:
        _check_mask(enabled_models, self._num_models())
        self._enabled_models = enabled_models


def weighted_average(predictions: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """``base_models/utility.py:4-11``."""
    assert predictions.shape == weights.shape
    weights = weights / (torch.sum(weights, dim=0) + 1e-9)
    return torch.sum(weights * predictions, dim=0)


# ===================================================================== particle filter side
class CrossmodalWeightModel(nn.Module, abc.ABC):
    """``forward(*, observations) -> (N, modality_count)`` log-weights."""

    def __init__(self, modality_count: int):
        super().__init__()
        self.modality_count = modality_count

    @abc.abstractmethod
    def forward(self, *, observations) -> torch.Tensor:
        ...


class CrossmodalParticleFilterMeasurementModel(base.ParticleFilterMeasurementModel, _EnabledModels):
    def __init__(self, *, measurement_models: List[base.ParticleFilterMeasurementModel],
                 crossmodal_weight_model: Optional[CrossmodalWeightModel], state_dim: int):
        super().__init__(state_dim=state_dim)
        self.measurement_models = nn.ModuleList(measurement_models)
        self.crossmodal_weight_model = crossmodal_weight_model
        self._enabled_models: List[bool] = [True for _ in self.measurement_models]

    def _num_models(self):
        return len(self.measurement_models)

    def _fusable(self) -> bool:
        return all(hasattr(m, "forward_encoded") for m in self.measurement_models)

    def encode_observations(self, observations):
        """Everything that depends on the observation only: each unimodal model's hoisted
        join-layer bias and the modality log-weights."""
        ctx = {}
        live = [m if self._enabled_models[i] else None for i, m in enumerate(self.measurement_models)]
        # every image encoder of this step (unimodal models + weight model) in one K4 batch
        feats = encode_observation_images(live + [self.crossmodal_weight_model], observations)
        for i, m in enumerate(live):
            if m is not None:
                enc = call_with_image_feat(m.encode_observations, feats[i], observations=observations) \
                    if feats[i] is not None else m.encode_observations(observations)
                for k, v in enc.items():
                    ctx[f"m{i}.{k}"] = v
        if self.crossmodal_weight_model is not None:
            ctx["modality_log_weights"] = call_with_image_feat(
                self.crossmodal_weight_model, feats[-1], observations=observations
            ).to(torch.float32).contiguous()
        return ctx

    def fused_measurements(self, ctx):
        """Enabled unimodal networks with their hoisted biases and modality log-weight columns,
        for the native step loop; ``None`` when a unimodal model is not a fused network."""
        if not self._fusable() or not all(hasattr(m, "fused_measurements") for m in self.measurement_models):
            return None
        beta = ctx.get("modality_log_weights")
        K = self._num_models()
        out = []
        for i, m in enumerate(self.measurement_models):
            if self._enabled_models[i]:
                sub = {k[len(f"m{i}."):]: v for k, v in ctx.items() if k.startswith(f"m{i}.")}
                net, bias, _ = m.fused_measurements(sub)[0][0]
                out.append((net, bias, None if beta is None else beta.view(-1)[i:]))
        return out, K

    def train_plan(self, ctx):
        """Enabled unimodal networks for the native training recursion (``engine.PfTrainLoopFunction``):
        ``([(network, column of beta | None)], [hoisted bias (R, 64)], beta (R, K) | None, K)``; ``None`` when a
        unimodal model is not a fused network.  ``ctx`` comes from ``encode_observations_autograd``."""
        if ctx is None or not all(hasattr(m, "_net") and hasattr(m, "train_plan") for m in self.measurement_models):
            return None
        beta = ctx.get("modality_log_weights")
        nets, biases = [], []
        for i, m in enumerate(self.measurement_models):
            if self._enabled_models[i]:
                nets.append((m._net, i if beta is not None else None))
                biases.append(ctx[f"m{i}.bias"])
        if not nets:
            return None
        return nets, biases, beta, self._num_models()

    def forward_encoded(self, states: torch.Tensor, ctx) -> torch.Tensor:
        N, M, _ = states.shape
        loglik = torch.empty((N, M), dtype=torch.float32, device=states.device)
        beta = ctx.get("modality_log_weights")
        K = self._num_models()
        first = True
        for i, m in enumerate(self.measurement_models):
            if not self._enabled_models[i]:
                continue
            sub = {k[len(f"m{i}."):]: v for k, v in ctx.items() if k.startswith(f"m{i}.")}
            m.forward_encoded(states, sub, loglik=loglik, combine=not first,
                              modality_logw=None if beta is None else beta.view(-1)[i:],
                              logw_stride=K)
            first = False
        assert not first, "no measurement model enabled"
        return loglik

    def encode_observations_autograd(self, observations):
        """Training counterpart of ``encode_observations`` (differentiable torch ops on ``R`` rows):
        each enabled unimodal model's hoisted bias and the modality log-weights; ``None`` when a
        unimodal model does not implement the protocol."""
        if not all(hasattr(m, "encode_observations_autograd") for m in self.measurement_models):
            return None
        ctx = {}
        for i, m in enumerate(self.measurement_models):
            if self._enabled_models[i]:
                for k, v in m.encode_observations_autograd(observations).items():
                    ctx[f"m{i}.{k}"] = v
        if self.crossmodal_weight_model is not None:
            ctx["modality_log_weights"] = self.crossmodal_weight_model(observations=observations)
        return ctx

    def forward_encoded_autograd(self, states, ctx):
        """``logsumexp_k(log beta_k + ll_k)`` (``crossmodal_pf.py:106-139``) from hoisted terms."""
        N, M, _ = states.shape
        on = self._enabled_models
        ll = torch.stack(
            [m.forward_encoded_autograd(states, {k[len(f"m{i}."):]: v for k, v in ctx.items()
                                                 if k.startswith(f"m{i}.")})
             for i, m in enumerate(self.measurement_models) if on[i]], dim=2)
        beta = ctx.get("modality_log_weights")
        if beta is not None:
            ll = ll + beta[:, on][:, None, :]
        return torch.logsumexp(ll, dim=2)

    def forward(self, *, states, observations):
        N, M, _state_dim = states.shape
        if self._fusable() and not use_autograd(self):
            with torch.no_grad():
                return self.forward_encoded(states.contiguous(), self.encode_observations(observations))
        # user-supplied unimodal models: stock formulation (crossmodal_pf.py:106-139)
        ll = torch.stack(
            [m(states=states, observations=observations)
             for i, m in enumerate(self.measurement_models) if self._enabled_models[i]], dim=2)
        assert ll.shape == (N, M, np.sum(self._enabled_models))
        if self.crossmodal_weight_model is not None:
            beta = self.crossmodal_weight_model(observations=observations)[:, self._enabled_models]
            assert beta.shape == (N, np.sum(self._enabled_models))
            ll = ll + beta[:, None, :]
        return torch.logsumexp(ll, dim=2)


# ===================================================================== Kalman filter side
class CrossmodalKalmanFilterWeightModel(nn.Module, abc.ABC):
    """``forward(*, observations) -> (modality_count, N, state_dim)`` weights."""

    def __init__(self, modality_count: int, state_dim: int):
        super().__init__()
        self.modality_count = modality_count
        self.state_dim = state_dim

    @abc.abstractmethod
    def forward(self, *, observations) -> torch.Tensor:
        ...


class _FusedKalmanFilters(base.Filter, _EnabledModels):
    """Shared machinery: K virtual-sensor EKFs stepped and fused by one K3 launch."""

    def __init__(self, *, filter_models: List[filters.VirtualSensorExtendedKalmanFilter],
                 state_dim: int):
        super().__init__(state_dim=state_dim)
        self.filter_models = nn.ModuleList(filter_models)
        self._enabled_models: List[bool] = [True for _ in self.filter_models]
        self.weighted_covariances = None

    def _num_models(self):
        return len(self.filter_models)

    @property
    def state_covariance_estimate(self):
        return self.weighted_covariances

    def initialize_beliefs(self, *, mean: torch.Tensor, covariance: torch.Tensor):
        N = mean.shape[0]
        assert mean.shape == (N, self.state_dim)
        assert covariance.shape == (N, self.state_dim, self.state_dim)
        for model in self.filter_models:
            model.initialize_beliefs(mean=mean, covariance=covariance)

    # -- belief-independent work of one time step: virtual sensors (+ fusion weights), with
    #    every image encoder involved batched into one K4 launch sequence
    def _encode_step(self, observations):
        on = self._enabled_models
        sensors = [f.virtual_sensor_model if on[i] else None for i, f in enumerate(self.filter_models)]
        wm = getattr(self, "crossmodal_weight_model", None)
        if wm is not None and np.sum(on) < len(on):
            wm = None  # masked fusion uses 0/1 weights (crossmodal_kf.py:124-133)
        feats = encode_observation_images(sensors + [wm], observations)
        out = {"sensor": [None if m is None else call_with_image_feat(m, feats[i], observations=observations)
                          for i, m in enumerate(sensors)],
               "weights": None if wm is None else call_with_image_feat(wm, feats[-1], observations=observations)}
        return out

    def _encode_controls(self, controls):
        return [f.dynamics_model.encode_controls(controls)
                if hasattr(f.dynamics_model, "predict_with_jacobian") else None
                for f in self.filter_models]

    def _fused_step(self, controls, enc, ctrl, *, fusion: int, fuse_w, feedback: int):
        """Every enabled sub-filter's predict + correct and the fusion in ONE launch.
        Returns ``(mu_fused, Sigma_fused, mu_k, Sigma_k)`` (fused ones ``None`` for fusion 0)."""
        live_idx = [i for i, on in enumerate(self._enabled_models) if on]
        live = [self.filter_models[i] for i in live_idx]
        A, mu_pred, L, z, r = [], [], [], [], []
        for i, f in zip(live_idx, live):
            assert f._initialized, "Kalman filter not initialized!"
            mp, Ak, Lk = f._predict_pieces(controls, None if ctrl is None else ctrl[i])
            A.append(Ak); mu_pred.append(mp); L.append(Lk)
            z.append(enc["sensor"][i][0].to(torch.float32)); r.append(enc["sensor"][i][1].to(torch.float32))
        K = len(live)
        N, d = mu_pred[0].shape
        dev = mu_pred[0].device
        mu = torch.empty((K, N, d), dtype=torch.float32, device=dev)
        Sigma = torch.stack([f._belief_covariance for f in live]).contiguous()
        mu_f = Sigma_f = None
        if fusion:
            mu_f = torch.empty((N, d), dtype=torch.float32, device=dev)
            Sigma_f = torch.empty((N, d, d), dtype=torch.float32, device=dev)
        _abi.ekf_step(torch.stack(A).contiguous(), torch.stack(mu_pred).contiguous(),
                      torch.stack(L).contiguous(), torch.stack(z).contiguous(),
                      torch.stack(r).contiguous(),
                      None if fuse_w is None else fuse_w.to(torch.float32).contiguous(),
                      mu, Sigma, mu_f, Sigma_f, fusion=fusion, feedback=feedback)
        for k, f in enumerate(live):
            f._belief_mean, f._belief_covariance = mu[k], Sigma[k]
        return mu_f, Sigma_f, mu, Sigma

    def _autograd_unimodal(self, observations, controls):
        live = [f for i, f in enumerate(self.filter_models) if self._enabled_models[i]]
        means = torch.stack([f(observations=observations, controls=controls) for f in live])
        covs = torch.stack([f._belief_covariance for f in live])
        return means, covs

    @engine.checked_step
    def forward(self, *, observations, controls):
        N, _ = controls.shape
        if use_autograd(self):
            return self._forward_autograd(observations, controls)
        with torch.no_grad():
            return self._forward_encoded(observations, controls, self._encode_step(observations),
                                         self._encode_controls(controls))

    def _encode_loop(self, observations, T, N):
        """``_encode_step`` for all ``T`` steps at once: virtual sensors (and, when the weight
        model exposes its row-wise part as ``raw_weights``, the fusion weights) are evaluated on
        the ``T*N`` flattened rows in one launch sequence; only the per-step, batch-coupled
        finish of the weights (the reference's reshape quirk, Q3) stays inside the loop."""
        on = self._enabled_models
        flat = tree_map(observations, lambda x: x.reshape((T * N,) + tuple(x.shape[2:])))
        sensors = [f.virtual_sensor_model if on[i] else None for i, f in enumerate(self.filter_models)]
        wm = getattr(self, "crossmodal_weight_model", None)
        if wm is not None and np.sum(on) < len(on):
            wm = None
        batched_w = wm is not None and hasattr(wm, "raw_weights")
        # only ROW-WISE sensors may see the T*N flattened rows: a sensor that couples the rows of
        # a batch (batch statistics, the reshape-quirk weight model inside a fused sensor) must be
        # evaluated one step at a time, exactly as step-by-step ``forward`` would
        row_wise = [m is not None and getattr(m, "row_wise", False) for m in sensors]
        feats = encode_observation_images([m if rw else None for m, rw in zip(sensors, row_wise)]
                                          + [wm if batched_w else None], flat)
        outs = []
        for i, m in enumerate(sensors):
            if m is None:
                outs.append(None)
            elif row_wise[i]:
                outs.append(call_with_image_feat(m, feats[i], observations=flat))
            else:
                steps = [m(observations=tree_index(observations, t)) for t in range(T)]
                outs.append((torch.cat([z for z, _ in steps]), torch.cat([r for _, r in steps])))
        raw = call_with_image_feat(wm.raw_weights, feats[-1], observations=flat) if batched_w else None
        w_all = wm.finish_weights_steps(raw, T) if batched_w and hasattr(wm, "finish_weights_steps") else None
        encs = []
        for t in range(T):
            sl = slice(t * N, (t + 1) * N)
            if wm is None:
                w = None
            elif w_all is not None:
                w = w_all[t]
            elif batched_w:
                w = wm.finish_weights(raw[sl])
            else:
                w = wm(observations=tree_index(observations, t))
            encs.append({"sensor": [None if o is None else (o[0][sl], o[1][sl]) for o in outs], "weights": w})
        self._loop_sensor_outputs = outs  # (T*N, ...) blocks, for the native step loop
        return encs

    def _native_plan(self, encs, T, N, observations=None):
        """``(fusion, feedback, fuse_w (T, K, N, d) | None[, feedback gate (T,) int32 | None])`` for
        ``mmf_ekf_forward_loop``, or ``None`` when this filter's step is not a plain K5 + K3 sequence."""
        return None

    def _native_loop(self, encs, ctrl_all, T, N, observations=None):
        """All ``T`` steps through ``mmf_ekf_forward_loop`` (one C call); ``None`` -> Python loop."""
        plan = self._native_plan(encs, T, N, observations)
        live_idx = [i for i, on in enumerate(self._enabled_models) if on]
        live = [self.filter_models[i] for i in live_idx]
        if plan is None or T == 0 or len(live) > _abi.LOOP_MAX_MEAS:
            return None
        dyns = [f.dynamics_model for f in live]
        if not all(hasattr(m, "_net") and hasattr(m, "predict_with_jacobian") for m in dyns):
            return None
        if any(ctrl_all[i] is None for i in live_idx) or len({m._net.n_res for m in dyns}) != 1:
            return None
        fusion, feedback, fuse_w = plan[:3]
        gate = plan[3] if len(plan) > 3 else None  # (T,) int32 device words: step t writes back only where != 0
        outs = self._loop_sensor_outputs
        d, K = self.state_dim, len(live)
        dev = live[0]._belief_mean.device
        for f in live:
            assert f._initialized, "Kalman filter not initialized!"
        f32 = lambda x: x.to(torch.float32)
        z = torch.stack([f32(outs[i][0]).view(T, N, d) for i in live_idx], dim=1).contiguous()
        r = torch.stack([f32(outs[i][1]).view(T, N, d, d) for i in live_idx], dim=1).contiguous()
        mu = torch.stack([f._belief_mean for f in live]).contiguous()
        Sigma = torch.stack([f._belief_covariance for f in live]).contiguous()
        q = torch.stack([m.scale_tril() for m in dyns]).to(torch.float32).contiguous()
        mu_pred, A = torch.empty_like(mu), torch.empty_like(Sigma)
        Sigma_f = torch.empty((N, d, d), dtype=torch.float32, device=dev)
        est = torch.empty((T, N, d), dtype=torch.float32, device=dev)
        prec = dyns[0]._net.precision_code()
        blobs = [m._net.blob(prec) for m in dyns]
        biases = [ctrl_all[i]["bias"] for i in live_idx]
        fw = None if fuse_w is None else f32(fuse_w).contiguous()
        P = lambda t: None if t is None else ctypes.c_void_p(_abi.ptr(t))
        a = _abi.MmfEkfLoopArgs()
        a.T, a.N, a.d, a.K, a.fusion, a.feedback = T, N, d, K, fusion, feedback
        a.n_res_dyn, a.precision = dyns[0]._net.n_res, prec
        a.range_flag = (ctypes.c_void_p(_abi.ptr(engine.range_flag(dev), dtype=torch.int32))
                        if prec != _abi.PREC_F32 else None)
        for k in range(K):
            a.dyn_packed[k], a.dyn_bias[k] = P(blobs[k]), P(biases[k])
        a.q_tril, a.z, a.r_tril, a.fuse_w = P(q), P(z), P(r), P(fw)
        a.mu, a.Sigma, a.mu_pred, a.A, a.Sigma_f, a.estimates = P(mu), P(Sigma), P(mu_pred), P(A), P(Sigma_f), P(est)
        if gate is not None:
            a.feedback_gate = ctypes.c_void_p(_abi.ptr(gate, dtype=torch.int32))
        engine.run_ekf_loop(a, mu, Sigma)
        for k, f in enumerate(live):
            f._belief_mean, f._belief_covariance = mu[k], Sigma[k]
        return est, (Sigma_f if fusion else None)

    @engine.checked_loop
    def forward_loop(self, *, observations, controls):
        """Sensors, fusion weights and control encoders do not depend on the belief: they are
        evaluated ahead of the recursion for all ``T*N`` rows at once."""
        T, N = tree_leading_shape(controls)[:2]
        if use_autograd(self):
            return base.Filter.forward_loop(self, observations=observations, controls=controls)
        with torch.no_grad():
            encs = self._encode_loop(observations, T, N)
            flat = tree_map(controls, lambda x: x.reshape((T * N,) + tuple(x.shape[2:])))
            ctrl_all = self._encode_controls(flat)
            native = self._native_loop(encs, ctrl_all, T, N, observations)
            if native is not None:
                return self._after_native_loop(*native)
            out = []
            for t in range(T):
                sl = slice(t * N, (t + 1) * N)
                ctrl = [None if c is None else {k: v[sl] for k, v in c.items()} for c in ctrl_all]
                out.append(self._forward_encoded(tree_index(observations, t), tree_index(controls, t),
                                                 encs[t], ctrl))
        return torch.stack(out, dim=0)


class CrossmodalKalmanFilter(_FusedKalmanFilters):
    """Learned-weight fusion of unimodal EKFs (``base_models/crossmodal_kf.py:39-240``):
    ``mu = sum_k w_k mu_k / (sum_k w_k + 1e-9)``, ``Sigma = sum_k (w_k w_k^T) (.) Sigma_k``.

    ``feedback``: the reference assigns the fused belief to ``f.states_prev`` /
    ``f.states_covariance_prev`` (``:147-149``) while its sub-filters keep their belief in
    ``_belief_mean`` / ``_belief_covariance`` (``:180``), so that write never reaches them
    (SURVEY.md appendix C, Q1).  ``"none"`` (default) reproduces this; ``"belief"`` does the
    write-back the code intends.
    """

    def __init__(self, *, filter_models, crossmodal_weight_model: CrossmodalKalmanFilterWeightModel,
                 state_dim: int, feedback: str = "none"):
        super().__init__(filter_models=filter_models, state_dim=state_dim)
        self.crossmodal_weight_model = crossmodal_weight_model
        assert feedback in ("none", "belief")
        self.feedback = feedback

    def _state_weights(self, raw, N, device):
        on = self._enabled_models
        if np.sum(on) < len(on):
            w = torch.tensor(on, dtype=torch.float32, device=device)
            w = w[:, None, None].repeat(1, N, self.state_dim)
        else:
            w = raw
        w = _select_enabled(w, on)
        assert w.shape == (np.sum(on), N, self.state_dim)
        return w

    def _native_plan(self, encs, T, N, observations=None):
        dev = self.filter_models[0]._belief_mean.device
        w = torch.stack([self._state_weights(e["weights"], N, dev) for e in encs])
        return 1, (1 if self.feedback == "belief" else 0), w

    def _after_native_loop(self, estimates, Sigma_f):
        self.weighted_covariances = Sigma_f
        for f in self.filter_models:  # inert attributes, exactly as the reference sets them
            f.states_prev = estimates[-1]
            f.states_covariance_prev = Sigma_f
        return estimates

    def _forward_encoded(self, observations, controls, enc, ctrl):
        N = tree_leading_shape(controls)[0]
        dev = self.filter_models[0]._belief_mean.device
        w = self._state_weights(enc["weights"], N, dev)
        fb = 1 if self.feedback == "belief" else 0
        mu_f, Sigma_f, _, _ = self._fused_step(controls, enc, ctrl, fusion=1, fuse_w=w, feedback=fb)
        self.weighted_covariances = Sigma_f
        for f in self.filter_models:  # inert attributes, exactly as the reference sets them
            f.states_prev = mu_f
            f.states_covariance_prev = Sigma_f
        return mu_f

    def _forward_autograd(self, observations, controls):
        """Differentiable torch formulation (training backend "autograd"), as the reference."""
        N = controls.shape[0]
        means, covs = self._autograd_unimodal(observations, controls)
        raw = None
        if np.sum(self._enabled_models) == len(self._enabled_models):
            raw = self.crossmodal_weight_model(observations=observations)
        w = self._state_weights(raw, N, means.device)
        mu, Sigma = self.calculate_weighted_states(w, means, covs)
        self.weighted_covariances = Sigma
        for f in self.filter_models:
            f.states_prev, f.states_covariance_prev = mu, Sigma
            if self.feedback == "belief":
                f._belief_mean, f._belief_covariance = mu, Sigma
        return mu

    # kept for API parity with the reference (``crossmodal_kf.py:153-186``)
    def calculate_weighted_states(self, state_weights, unimodal_states, unimodal_covariances):
        model_dim, N, state_dim = state_weights.shape
        assert state_dim == self.state_dim
        mu = weighted_average(unimodal_states, state_weights)
        cw = state_weights.unsqueeze(-1).repeat((1, 1, 1, self.state_dim))
        cw = cw * cw.transpose(-1, -2)
        return mu, torch.sum(cw * unimodal_covariances, 0)

    def calculate_unimodal_states(self, observations, controls):
        with torch.no_grad():
            _, _, mu, Sigma = self._fused_step(controls, self._encode_step(observations),
                                               self._encode_controls(controls),
                                               fusion=0, fuse_w=None, feedback=0)
        return mu, Sigma

    def measurement_initialize_beliefs(self, observations):
        """``crossmodal_kf.py:208-240`` (per trajectory, off the hot path)."""
        on = self._enabled_models
        with torch.no_grad():
            outs = [f.virtual_sensor_model(observations=observations)
                    for i, f in enumerate(self.filter_models) if on[i]]
            means = torch.stack([x[0] for x in outs])
            trils = torch.stack([x[1] for x in outs])
            covs = trils @ trils.transpose(-1, -2)
            w = _select_enabled(self.crossmodal_weight_model(observations=observations), on)
            mu = weighted_average(means, w)
            mult = torch.prod(torch.prod(w, dim=-1), dim=0).unsqueeze(-1).unsqueeze(-1)
            self.initialize_beliefs(mean=mu, covariance=mult * torch.sum(covs, dim=0))


class UnimodalKalmanFilter(_FusedKalmanFilters):
    """Information-form fusion (``base_models/unimodal_kf.py:118-270``):
    ``Sigma = (sum_k (Sigma_k + 1e-9)^-1 + 1e-9)^-1``, ``mu = Sigma sum_k P_k mu_k``.
    As in the reference the fused belief is returned but never fed back (Q6)."""

    def __init__(self, *, filter_models, state_dim: int):
        super().__init__(filter_models=filter_models, state_dim=state_dim)

    def _forward_autograd(self, observations, controls):
        means, covs = self._autograd_unimodal(observations, controls)
        if means.shape[0] == 1:
            return means[0]
        prec = torch.inverse(covs + 1e-9)
        Sigma = torch.inverse(torch.sum(prec, dim=0) + 1e-9)
        return (Sigma @ torch.sum(prec @ means[..., None], dim=0)).squeeze(-1)

    def _native_plan(self, encs, T, N, observations=None):
        return (0 if np.sum(self._enabled_models) == 1 else 2), 0, None

    def _after_native_loop(self, estimates, Sigma_f):
        return estimates

    def _forward_encoded(self, observations, controls, enc, ctrl):
        if np.sum(self._enabled_models) == 1:
            _, _, mu, _ = self._fused_step(controls, enc, ctrl, fusion=0, fuse_w=None, feedback=0)
            return mu[0]
        mu_f, _, _, _ = self._fused_step(controls, enc, ctrl, fusion=2, fuse_w=None, feedback=0)
        return mu_f


def _select_enabled(w: torch.Tensor, on) -> torch.Tensor:
    """``w[on]`` for a host-side boolean list ``on`` without the device round trip of boolean-mask
    indexing (mask upload + ``nonzero`` + a host sync per call -- once per time step in the loop
    prologue): all enabled is a no-op, otherwise the enabled rows are stacked as views."""
    on = [bool(x) for x in on]
    if all(on):
        return w
    return torch.stack([w[i] for i, x in enumerate(on) if x])


# ===================================================================== fused virtual sensors
def _fuse_sensors(means, trils, w, *, mode: int):
    """R11 algebra in one launch (``mmf_fuse_virtual_sensors``)."""
    K, N, d = means.shape
    f32 = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    z_out = torch.empty((N, d), dtype=torch.float32, device=means.device)
    t_out = torch.empty((N, d, d), dtype=torch.float32, device=means.device)
    _abi.fuse_virtual_sensors(f32(means), f32(trils), f32(w), z_out, t_out, mode)
    return z_out, t_out


class CrossmodalVirtualSensorModel(base.VirtualSensorModel, _EnabledModels):
    """``base_models/crossmodal_kf.py:243-359``: fuse K virtual sensors *before* one EKF;
    ``Sigma = (prod_k prod_i w_ki) sum_k Sigma_k``; returns a Cholesky factor."""

    def __init__(self, *, virtual_sensor_model: List[base.VirtualSensorModel],
                 crossmodal_weight_model: CrossmodalKalmanFilterWeightModel, state_dim: int):
        super().__init__(state_dim=state_dim)
        self.virtual_sensor_model = nn.ModuleList(virtual_sensor_model)
        self.crossmodal_weight_model = crossmodal_weight_model
        self._enabled_models: List[bool] = [True for _ in self.virtual_sensor_model]

    def _num_models(self):
        return len(self.virtual_sensor_model)

    def forward(self, *, observations):
        on = self._enabled_models
        outs = [m(observations=observations) for i, m in enumerate(self.virtual_sensor_model) if on[i]]
        means = torch.stack([x[0] for x in outs])
        trils = torch.stack([x[1] for x in outs])
        N = means.shape[1]
        if np.sum(on) < len(on):
            w = torch.tensor(on, dtype=torch.float32, device=means.device)
            w = w[:, None, None].repeat(1, N, self.state_dim)
        else:
            w = self.crossmodal_weight_model(observations=observations)
        w = _select_enabled(w, on)
        assert w.shape == (np.sum(on), N, self.state_dim)
        if not use_autograd(self):
            return _fuse_sensors(means, trils, w, mode=1)
        covs = trils @ trils.transpose(-1, -2)
        mu = weighted_average(means, w)
        mult = torch.prod(torch.prod(w, dim=-1), dim=0).unsqueeze(-1).unsqueeze(-1)
        assert mult.shape == (N, 1, 1)
        return mu, torch.linalg.cholesky(mult * torch.sum(covs, dim=0))


class UnimodalVirtualSensorModel(base.VirtualSensorModel, _EnabledModels):
    """``base_models/unimodal_kf.py:13-115`` including its quirk Q5 (element-wise
    ``1 / (scale_tril + 1e-9)`` used as "precision"; a covariance returned where a
    scale-tril is expected)."""

    def __init__(self, *, virtual_sensor_model: List[base.VirtualSensorModel], state_dim: int):
        super().__init__(state_dim=state_dim)
        self.virtual_sensor_model = nn.ModuleList(virtual_sensor_model)
        self._enabled_models: List[bool] = [True for _ in self.virtual_sensor_model]

    def _num_models(self):
        return len(self.virtual_sensor_model)

    def forward(self, *, observations):
        on = self._enabled_models
        outs = [m(observations=observations) for i, m in enumerate(self.virtual_sensor_model) if on[i]]
        means = torch.stack([x[0] for x in outs])
        trils = torch.stack([x[1] for x in outs])
        if not use_autograd(self):
            return _fuse_sensors(means, trils, None, mode=2)
        if np.sum(on) == 1:
            return means[0], (trils @ trils.transpose(-1, -2))[0]
        prec = 1.0 / (trils + 1e-9)
        w = torch.diagonal(prec, dim1=-2, dim2=-1)
        assert w.shape == means.shape
        return weighted_average(means, w), torch.inverse(torch.sum(prec, dim=0) + 1e-9)
