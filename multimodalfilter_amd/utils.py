"""Small host-side helpers: dict-or-tensor slicing (the role ``fannypack.utils.SliceWrapper``
plays in ``/root/reference/crossmodal/eval_helpers.py:88-142``) and explicit randomness."""
from typing import Any, Callable

import torch


def tree_map(x: Any, fn: Callable):
    if isinstance(x, dict):
        return {k: fn(v) for k, v in x.items()}
    return fn(x)


def tree_index(x: Any, index):
    return tree_map(x, lambda t: t[index])


def tree_leading_shape(x: Any):
    if isinstance(x, dict):
        shapes = [tuple(v.shape) for v in x.values()]
        out = []
        for dims in zip(*shapes):
            if len(set(dims)) != 1:
                break
            out.append(dims[0])
        return tuple(out)
    return tuple(x.shape)


def rank_seed(base_seed: int = 0) -> int:
    """``base_seed`` offset by this process's data-parallel rank (``torch.distributed`` when
    initialised, else torchrun's ``RANK``): ranks own different trajectories and must not
    propagate them with identical noise streams."""
    import os

    import torch.distributed as dist

    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else int(os.environ.get("RANK", "0"))
    return int(base_seed) + 1_000_003 * rank


class NoiseSource:
    """Every random draw of a filter goes through one of these, so that a CPU oracle and
    the HIP engine can consume identical, pre-drawn tensors.  The generator is created on
    first use (per device) from ``seed`` -- offset by the data-parallel rank unless
    ``per_rank=False`` -- and then ADVANCES: consecutive draws are independent."""

    def __init__(self, seed: int = 0, device=None, per_rank: bool = True):
        self.seed = seed
        self.per_rank = per_rank
        self._gens = {}

    def _gen(self, device):
        key = str(device)
        if key not in self._gens:
            seed = rank_seed(self.seed) if self.per_rank else self.seed
            self._gens[key] = torch.Generator(device=device).manual_seed(seed)
        return self._gens[key]

    def gaussian(self, shape, *, like: torch.Tensor) -> torch.Tensor:
        return torch.randn(shape, generator=self._gen(like.device), dtype=torch.float32,
                           device=like.device)

    def uniform(self, shape, *, like: torch.Tensor) -> torch.Tensor:
        return torch.rand(shape, generator=self._gen(like.device), dtype=torch.float32,
                          device=like.device)

    # T consecutive per-step draws as one contiguous (T, *shape) block, in the same order a
    # step-by-step loop would consume them (gaussian then uniform, per step)
    def draw_steps(self, T: int, gauss_shape, unif_shape, *, like: torch.Tensor):
        gs, us = [], []
        for _ in range(T):
            gs.append(self.gaussian(gauss_shape, like=like))
            us.append(self.uniform(unif_shape, like=like) if unif_shape is not None else None)
        g = torch.stack(gs).contiguous()
        u = torch.stack(us).contiguous() if unif_shape is not None else None
        return g, u


class ReplayNoise(NoiseSource):
    """Feeds pre-drawn tensors in call order."""

    def __init__(self, gaussians=(), uniforms=()):
        self._g = list(gaussians)
        self._u = list(uniforms)

    def gaussian(self, shape, *, like):
        t = self._g.pop(0)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(device=like.device, dtype=torch.float32).contiguous()

    def uniform(self, shape, *, like):
        t = self._u.pop(0)
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(device=like.device, dtype=torch.float32).contiguous()


class StackedNoise(NoiseSource):
    """Pre-drawn randomness kept as contiguous blocks: ``eps0 (N, M, d)`` for
    ``initialize_beliefs``, ``eps (T, N, M, d)`` and ``u (T, N[, M])`` for the steps.  The C
    step loop consumes the blocks directly (zero copy); step-by-step use gets views."""

    def __init__(self, eps0, eps, u):
        self._eps0, self._eps, self._u = eps0, eps, u
        self._tg = self._tu = 0

    def gaussian(self, shape, *, like):
        if self._eps0 is not None:
            t, self._eps0 = self._eps0, None
        else:
            t, self._tg = self._eps[self._tg], self._tg + 1
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(device=like.device, dtype=torch.float32).contiguous()

    def uniform(self, shape, *, like):
        t, self._tu = self._u[self._tu], self._tu + 1
        assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
        return t.to(device=like.device, dtype=torch.float32).contiguous()

    def draw_steps(self, T, gauss_shape, unif_shape, *, like):
        assert self._eps0 is None, "initialize_beliefs() consumes eps0 first"
        g = self._eps[self._tg:self._tg + T]
        assert tuple(g.shape) == (T,) + tuple(gauss_shape)
        self._tg += T
        u = None
        if unif_shape is not None:
            u = self._u[self._tu:self._tu + T]
            assert tuple(u.shape) == (T,) + tuple(unif_shape)
            self._tu += T
        return g.to(device=like.device, dtype=torch.float32).contiguous(), \
            None if u is None else u.to(device=like.device, dtype=torch.float32).contiguous()


class CounterBlock:
    """Marker returned by ``CounterNoise.draw_steps`` in place of a ``(T, N, M, d)`` tensor: the ``T``
    per-step noise blocks are counter steps ``step0 .. step0 + T - 1`` of ``seed`` -- generated inside the
    dynamics kernel, never materialised."""

    def __init__(self, seed: int, step0: int, traj0: int, T: int, shape):
        self.seed, self.step0, self.traj0, self.T, self.shape = seed, step0, traj0, T, tuple(shape)


class CounterNoise(NoiseSource):
    """Counter-based randomness (``include/mmf_philox.h``): every ``(N, M, d)`` Gaussian block is counter
    step ``t`` of Philox4x32-10 under ``seed`` -- a pure function of (seed, t, global trajectory index,
    particle) -- and every ``(N,)`` block of resampling uniforms likewise.  A native ``forward_loop``
    generates the Gaussians inside the dynamics kernel (no ``(T, N, M, d)`` tensor exists); step-by-step
    use materialises one block per call with ``mmf_philox_normals``; ``oracle.strict.philox_normals``
    reproduces every draw bit for bit on the CPU.  ``traj_offset``: index of this shard's first
    trajectory, so a sharded run draws what the unsharded one would."""

    def __init__(self, seed: int = 0, traj_offset: int = 0):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.traj_offset = int(traj_offset)
        self.step_gaussian = 0
        self.step_uniform = 0

    def gaussian(self, shape, *, like: torch.Tensor) -> torch.Tensor:
        from . import _abi

        assert len(shape) == 3 and shape[2] <= 4, "counter noise draws (N, M, d <= 4) blocks"
        out = torch.empty(tuple(shape), dtype=torch.float32, device=like.device)
        _abi.philox_normals(self.seed, self.step_gaussian, self.traj_offset, out)
        self.step_gaussian += 1
        return out

    def uniform(self, shape, *, like: torch.Tensor) -> torch.Tensor:
        from . import _abi

        assert len(shape) == 1, "counter noise draws one uniform per trajectory (systematic resampling)"
        out = torch.empty((1, shape[0]), dtype=torch.float32, device=like.device)
        _abi.philox_uniforms(self.seed, self.step_uniform, self.traj_offset, out)
        self.step_uniform += 1
        return out[0]

    def draw_steps(self, T: int, gauss_shape, unif_shape, *, like: torch.Tensor):
        from . import _abi

        block = CounterBlock(self.seed, self.step_gaussian, self.traj_offset, T, gauss_shape)
        self.step_gaussian += T
        u = None
        if unif_shape is not None:
            assert len(unif_shape) == 1
            u = torch.empty((T, unif_shape[0]), dtype=torch.float32, device=like.device)
            _abi.philox_uniforms(self.seed, self.step_uniform, self.traj_offset, u)
            self.step_uniform += T
        return block, u
