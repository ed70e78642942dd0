"""Abstract model / filter interfaces (torchfilter.base restated).

Keyword-only signatures everywhere, exactly as the reference calls them
(``crossmodal/door_models/dynamics.py:37-42``, ``door_models/pf.py:63-65``,
``door_models/kf.py:81-83``, ``eval_helpers.py:128-142``).
"""
import abc
from typing import Tuple

import torch
import torch.nn as nn

from ..fp.utils import SliceWrapper


class NoiseSource:
    """Explicit randomness.  Oracle and HIP engine consume the same pre-drawn tensors, so
    every draw goes through one of these instead of torch's hidden global RNG."""

    def __init__(self, seed: int = 0):
        self._gen = torch.Generator(device="cpu").manual_seed(seed)

    def gaussian(self, shape, *, like: torch.Tensor) -> torch.Tensor:
        return torch.randn(shape, generator=self._gen, dtype=torch.float32).to(like)

    def uniform(self, shape, *, like: torch.Tensor) -> torch.Tensor:
        return torch.rand(shape, generator=self._gen, dtype=torch.float32).to(like)


class ReplayNoise(NoiseSource):
    """Feeds pre-drawn tensors in call order (``gaussians`` and ``uniforms`` are lists)."""

    def __init__(self, gaussians=(), uniforms=()):
        self._g = list(gaussians)
        self._u = list(uniforms)

    def gaussian(self, shape, *, like):
        t = self._g.pop(0)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(like)

    def uniform(self, shape, *, like):
        t = self._u.pop(0)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(like)


class DynamicsModel(nn.Module, abc.ABC):
    """``forward(*, initial_states (R,d), controls) -> (states (R,d), scale_trils (R,d,d))``."""

    def __init__(self, *, state_dim: int):
        super().__init__()
        self.state_dim = state_dim

    @abc.abstractmethod
    def forward(self, *, initial_states, controls) -> Tuple[torch.Tensor, torch.Tensor]:
        ...

    def forward_loop(self, *, initial_states, controls):
        """Open-loop rollout over ``controls (T,N,...)`` -> ``(T,N,d)``, ``(T,N,d,d)``
        (called at ``eval_helpers.py:135-137``)."""
        T = SliceWrapper(controls).shape[0]
        x = initial_states
        xs, trils = [], []
        for t in range(T):
            x, L = self(initial_states=x, controls=SliceWrapper(controls)[t])
            xs.append(x)
            trils.append(L)
        return torch.stack(xs, dim=0), torch.stack(trils, dim=0)

    def jacobian(self, *, initial_states, controls) -> torch.Tensor:
        """Default autograd Jacobian ``J[n,i,j] = d f_i / d x_j`` (batch
        replicated ``d`` times, identity mask, one ``autograd.grad``, ``create_graph``).
        The reference never overrides it (grep: 0 hits)."""
        with torch.enable_grad():
            N, d = initial_states.shape
            x = initial_states.detach().clone()[:, None, :].expand(N, d, d).contiguous()
            rep_controls = SliceWrapper(controls).map(
                lambda t: torch.repeat_interleave(t, repeats=d, dim=0)
            )
            x.requires_grad_(True)
            y = self(initial_states=x.reshape(-1, d), controls=rep_controls)[0].reshape(N, d, d)
            mask = torch.eye(d, dtype=x.dtype, device=x.device)[None].expand(N, d, d)
            (jac,) = torch.autograd.grad(y, x, mask, create_graph=True)
        return jac


class ParticleFilterMeasurementModel(nn.Module, abc.ABC):
    """``forward(*, states (N,M,d), observations) -> log-likelihoods (N,M)``."""

    def __init__(self, *, state_dim: int):
        super().__init__()
        self.state_dim = state_dim

    @abc.abstractmethod
    def forward(self, *, states, observations) -> torch.Tensor:
        ...


class KalmanFilterMeasurementModel(nn.Module, abc.ABC):
    """``forward(*, states (N,d)) -> (expected observations (N,o), scale_tril (N,o,o))``."""

    def __init__(self, *, state_dim: int, observation_dim: int):
        super().__init__()
        self.state_dim = state_dim
        self.observation_dim = observation_dim

    @abc.abstractmethod
    def forward(self, *, states):
        ...

    def jacobian(self, *, states) -> torch.Tensor:
        with torch.enable_grad():
            N, d = states.shape
            o = self.observation_dim
            x = states.detach().clone()[:, None, :].expand(N, o, d).contiguous()
            x.requires_grad_(True)
            y = self(states=x.reshape(-1, d))[0].reshape(N, o, o)
            mask = torch.eye(o, dtype=x.dtype, device=x.device)[None].expand(N, o, o)
            (jac,) = torch.autograd.grad(y, x, mask, create_graph=True)
        return jac


class VirtualSensorModel(nn.Module, abc.ABC):
    """``forward(*, observations) -> (virtual observation (N,d), scale_tril (N,d,d))``."""

    def __init__(self, *, state_dim: int):
        super().__init__()
        self.state_dim = state_dim

    @abc.abstractmethod
    def forward(self, *, observations):
        ...


class Filter(nn.Module, abc.ABC):
    """Stateful recursive estimator: belief lives on the module between calls."""

    def __init__(self, *, state_dim: int):
        super().__init__()
        self.state_dim = state_dim

    @abc.abstractmethod
    def initialize_beliefs(self, *, mean: torch.Tensor, covariance: torch.Tensor) -> None:
        ...

    def forward(self, *, observations, controls) -> torch.Tensor:
        """One time step.  Optional when ``forward_loop`` is overridden (the reference's
        LSTM baselines do exactly that, ``door_models/lstm.py:49-100``)."""
        out = self.forward_loop(
            observations=SliceWrapper(observations).map(lambda t: t[None]),
            controls=SliceWrapper(controls).map(lambda t: t[None]),
        )
        return out[0]

    def forward_loop(self, *, observations, controls) -> torch.Tensor:
        """``(T,N,...)`` in, ``(T,N,d)`` out (``eval_helpers.py:139-146``); sequential in t."""
        T = SliceWrapper(controls).shape[0]
        assert SliceWrapper(observations).shape[0] == T
        out = [
            self(observations=SliceWrapper(observations)[t], controls=SliceWrapper(controls)[t])
            for t in range(T)
        ]
        return torch.stack(out, dim=0)
