"""The ``torchfilter.base`` API surface, MI355X-side.

Same class names, keyword-only signatures, shapes and ``assert`` behaviour as the
interfaces the reference's models subclass (``/root/reference/crossmodal/door_models/
dynamics.py:11,37-42``, ``door_models/pf.py:30,63-65``, ``door_models/kf.py:31,81-83``,
``base_models/crossmodal_kf.py:39``), so door-/push-style model classes drop
in unchanged.  On top of that contract, models may implement the *encoded* protocol, which
is what lets a filter hoist per-trajectory work out of the per-particle kernels and batch
observation encoders over a whole ``forward_loop``:

    DynamicsModel.encode_controls(controls)            -> ctx   (tensors with leading N)
    DynamicsModel.propagate_encoded(states, ctx, noise) -> states'
    ParticleFilterMeasurementModel.encode_observations(observations) -> ctx
    ParticleFilterMeasurementModel.forward_encoded(states, ctx)       -> (N, M) log-lik
    VirtualSensorModel.forward(observations)  (already per trajectory)

Models that only implement ``forward`` still work: the filters fall back to calling it
(on the GPU) and use the HIP kernels for the recursion itself.
"""
import abc
from typing import Tuple

import torch
import torch.nn as nn

from .utils import tree_index, tree_leading_shape, tree_map


class DynamicsModel(nn.Module, abc.ABC):
    def __init__(self, *, state_dim: int):
        super().__init__()
        self.state_dim = state_dim

    @abc.abstractmethod
    def forward(self, *, initial_states, controls) -> Tuple[torch.Tensor, torch.Tensor]:
        """``(R, d), controls (R, ...) -> (states (R, d), scale_trils (R, d, d))``."""

    def forward_loop(self, *, initial_states, controls):
        """Open-loop rollout (``eval_helpers.py:135-137``): ``(T,N,...)`` -> ``(T,N,d)``, ``(T,N,d,d)``."""
        T = tree_leading_shape(controls)[0]
        x, xs, trils = initial_states, [], []
        for t in range(T):
            x, L = self(initial_states=x, controls=tree_index(controls, t))
            xs.append(x)
            trils.append(L)
        return torch.stack(xs, dim=0), torch.stack(trils, dim=0)

    def jacobian(self, *, initial_states, controls) -> torch.Tensor:
        """``J[n,i,j] = d f_i / d x_j``.  Generic autograd default (batch replicated ``d`` times,
        as upstream) for user models written in torch ops; the built-in
        dynamics models override it with the forward-mode HIP kernel (K5)."""
        with torch.enable_grad():
            N, d = initial_states.shape
            x = initial_states.detach().clone()[:, None, :].expand(N, d, d).contiguous()
            rep = tree_map(controls, lambda t: torch.repeat_interleave(t, repeats=d, dim=0))
            x.requires_grad_(True)
            y = self(initial_states=x.reshape(-1, d), controls=rep)[0].reshape(N, d, d)
            mask = torch.eye(d, dtype=x.dtype, device=x.device)[None].expand(N, d, d)
            (jac,) = torch.autograd.grad(y, x, mask, create_graph=True)
        return jac


class ParticleFilterMeasurementModel(nn.Module, abc.ABC):
    def __init__(self, *, state_dim: int):
        super().__init__()
        self.state_dim = state_dim

    @abc.abstractmethod
    def forward(self, *, states, observations) -> torch.Tensor:
        """``states (N, M, d)``, observations with leading ``N`` -> log-likelihoods ``(N, M)``."""


class KalmanFilterMeasurementModel(nn.Module, abc.ABC):
    def __init__(self, *, state_dim: int, observation_dim: int):
        super().__init__()
        self.state_dim = state_dim
        self.observation_dim = observation_dim

    @abc.abstractmethod
    def forward(self, *, states):
        """``(N, d) -> (expected observations (N, o), scale_tril (N, o, o))``."""


class VirtualSensorModel(nn.Module, abc.ABC):
    def __init__(self, *, state_dim: int):
        super().__init__()
        self.state_dim = state_dim

    @abc.abstractmethod
    def forward(self, *, observations):
        """observations -> ``(virtual observation (N, d), scale_tril (N, d, d))``."""


class Filter(nn.Module, abc.ABC):
    """Stateful recursive estimator; the belief lives on the module between calls and is
    reset by ``initialize_beliefs`` (``eval_helpers.py:128-131``)."""

    def __init__(self, *, state_dim: int):
        super().__init__()
        self.state_dim = state_dim

    @abc.abstractmethod
    def initialize_beliefs(self, *, mean: torch.Tensor, covariance: torch.Tensor) -> None:
        ...

    def forward(self, *, observations, controls) -> torch.Tensor:
        out = self.forward_loop(observations=tree_map(observations, lambda t: t[None]),
                                controls=tree_map(controls, lambda t: t[None]))
        return out[0]

    def forward_loop(self, *, observations, controls) -> torch.Tensor:
        """``(T, N, ...)`` in, ``(T, N, d)`` out (``eval_helpers.py:139-146``)."""
        from . import engine  # (engine imports this module's siblings)

        return engine.checked_loop(Filter._forward_loop_steps)(self, observations=observations, controls=controls)

    def _forward_loop_steps(self, *, observations, controls) -> torch.Tensor:
        T = tree_leading_shape(controls)[0]
        assert tree_leading_shape(observations)[0] == T
        out = [self(observations=tree_index(observations, t), controls=tree_index(controls, t))
               for t in range(T)]
        return torch.stack(out, dim=0)
