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


class NoiseSource:
    """Every random draw of a filter goes through one of these, so that a CPU oracle and
    the HIP engine can consume identical, pre-drawn tensors."""

    def __init__(self, seed: int = 0, device=None):
        self.seed = seed
        self._gens = {}

    def _gen(self, device):
        key = str(device)
        if key not in self._gens:
            self._gens[key] = torch.Generator(device=device).manual_seed(self.seed)
        return self._gens[key]

    def gaussian(self, shape, *, like: torch.Tensor) -> torch.Tensor:
        return torch.randn(shape, generator=self._gen(like.device), dtype=torch.float32,
                           device=like.device)

    def uniform(self, shape, *, like: torch.Tensor) -> torch.Tensor:
        return torch.rand(shape, generator=self._gen(like.device), dtype=torch.float32,
                          device=like.device)


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
