"""End-to-end filter training step.

Mirrors ``torchfilter.train.train_filter`` as the reference calls it
(``/root/reference/crossmodal/train_helpers.py:124-162``; upstream behaviour SURVEY.md A.2):
a batch of subsequences ``(L, N, ...)``, belief initialised at a sample
``~ N(states[0], initial_covariance)`` (or from the first observation), ``forward_loop`` over
``[1:]``, mean-squared error against ``states[1:]``, one optimiser step.  Randomness is explicit
(``noise``), as everywhere in this package.

The filter must be in ``train()`` mode with a training backend selected
(``engine.set_training_backend("hip")``: per-particle networks forward + backward in HIP,
K6; ``"autograd"``: torch ops throughout).  With ``torch.distributed`` initialised,
``all_reduce=True`` averages the gradients over ranks before the step (X2).
"""
from typing import Dict, Optional

import torch

from . import distributed, engine
from .utils import NoiseSource

# initial-belief perturbations when neither the caller nor the filter brings a noise source: ONE
# module-level generator that advances from step to step (upstream ``train_filter`` samples a
# fresh ``MultivariateNormal`` per batch), seeded per data-parallel rank on first use
_DEFAULT_NOISE = NoiseSource(20201025)


def default_noise(filter_model) -> NoiseSource:
    """The persistent source a training step draws from by default: the filter's own ``noise``
    (particle filters carry one) or the module-level one -- never a fresh seed-0 generator."""
    own = getattr(filter_model, "noise", None)
    return own if type(own) is NoiseSource else _DEFAULT_NOISE  # replayed / stacked blocks are the steps' own


def filter_loss(filter_model, batch: Dict[str, torch.Tensor], *, initial_covariance: torch.Tensor,
                noise: Optional[NoiseSource] = None, measurement_initialize: bool = False) -> torch.Tensor:
    """MSE of ``forward_loop`` on ``batch`` = ``{"states" (L, N, d), "controls" (L, N, 7), "image",
    "gripper_pos", "gripper_sensors" (L, N, ...)}`` (time-major)."""
    assert filter_model.training, "call filter_model.train() first"
    assert engine.TRAINING_BACKEND is not None, "select engine.set_training_backend('hip' | 'autograd')"
    states = batch["states"]
    L, N, d = states.shape
    obs = {k: batch[k] for k in ("image", "gripper_pos", "gripper_sensors")}
    if measurement_initialize and hasattr(filter_model, "measurement_initialize_beliefs"):
        filter_model.measurement_initialize_beliefs({k: v[0] for k, v in obs.items()})
    else:
        noise = noise if noise is not None else default_noise(filter_model)
        tril = torch.linalg.cholesky(initial_covariance.to(torch.float32))
        mean = states[0] + noise.gaussian((N, d), like=states) @ tril.t()
        filter_model.initialize_beliefs(mean=mean, covariance=initial_covariance[None].expand(N, d, d))
    pred = filter_model.forward_loop(observations={k: v[1:] for k, v in obs.items()}, controls=batch["controls"][1:])
    return torch.mean((pred - states[1:]) ** 2)


def train_filter_step(filter_model, batch, optimizer: torch.optim.Optimizer, *, initial_covariance: torch.Tensor,
                      noise: Optional[NoiseSource] = None, measurement_initialize: bool = False,
                      all_reduce: bool = False) -> float:
    """One optimisation step on one batch of subsequences; returns the loss."""
    optimizer.zero_grad(set_to_none=True)
    loss = filter_loss(filter_model, batch, initial_covariance=initial_covariance, noise=noise,
                       measurement_initialize=measurement_initialize)
    loss.backward()
    if all_reduce:
        distributed.all_reduce_gradients(filter_model)
    optimizer.step()
    return float(loss.detach())
