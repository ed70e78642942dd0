"""Seeded synthetic door-/push-task trajectories (the reference's datasets are Google-Drive
downloads, ``/root/reference/crossmodal/tasks/_door.py:11-20``, unreachable offline).

Shapes, value ranges and masking semantics follow the reference's loaders: z-scored states
(``tasks/_door.py:261-268``), 7-d controls whose last channel is a binary contact flag
(``:211-222,269-296``), 32x32 single-channel images in ``[-1, 1]``
(``scripts/door_task/data_collection/simulate_door.py:114``) zeroed by a per-frame blackout
mask with probability ``image_blackout_ratio`` (``tasks/_door.py:188-197``).
Everything is drawn on the CPU from one ``torch.Generator`` so that a CPU oracle and the HIP
engine can be fed identical tensors.
"""
from typing import Dict

import torch


def make_trajectories(*, state_dim: int, T: int, N: int, seed: int,
                      image_blackout_ratio: float = 0.0) -> Dict[str, torch.Tensor]:
    """Returns CPU tensors: ``states (T+1, N, d)``, ``controls (T+1, N, 7)`` and observations
    ``image (T+1, N, 32, 32)``, ``gripper_pos (T+1, N, 3)``, ``gripper_sensors (T+1, N, 7)``;
    index 0 is the initial time step (``eval_helpers.py:125-142`` filters on ``[1:]``)."""
    assert 0.0 <= image_blackout_ratio < 1.0
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = state_dim
    steps = 0.05 * torch.randn((T, N, d), generator=g)
    x0 = torch.randn((1, N, d), generator=g)
    states = torch.cat((x0, x0 + torch.cumsum(steps, dim=0)), dim=0)

    controls = torch.randn((T + 1, N, 7), generator=g)
    contact = (torch.rand((T + 1, N), generator=g) < 0.5).float()
    controls[..., 6] = (contact - 0.5) / 0.5  # binary flag, z-scored

    pos = torch.randn((T + 1, N, 3), generator=g)
    sensors = torch.randn((T + 1, N, 7), generator=g)

    # smooth blob whose centre is linear in the first two state dimensions
    grid = torch.linspace(-1.0, 1.0, 32)
    yy, xx = torch.meshgrid(grid, grid, indexing="ij")
    cx = torch.tanh(0.4 * states[..., 0])[..., None, None]
    cy = torch.tanh(0.4 * states[..., 1 % d])[..., None, None]
    blob = torch.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / 0.08)
    image = (2.0 * blob - 1.0) + 0.05 * torch.randn((T + 1, N, 32, 32), generator=g)
    image = image.clamp_(-1.0, 1.0)
    if image_blackout_ratio > 0:
        keep = (torch.rand((T + 1, N), generator=g) >= image_blackout_ratio).float()
        image = image * keep[..., None, None]
    return {"states": states, "controls": controls, "image": image.contiguous(),
            "gripper_pos": pos, "gripper_sensors": sensors}


def observations_of(traj: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {k: traj[k] for k in ("image", "gripper_pos", "gripper_sensors")}


def draw_filter_noise(*, T: int, N: int, M: int, state_dim: int, seed: int, mode: str = "systematic"):
    """Pre-drawn randomness for a particle filter run: initial particles, per-step process
    noise and per-step resampling uniforms (lists, in consumption order)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    eps0 = torch.randn((N, M, state_dim), generator=g)
    eps = [torch.randn((N, M, state_dim), generator=g) for _ in range(T)]
    us = [torch.rand((N,) if mode == "systematic" else (N, M), generator=g) for _ in range(T)]
    return eps0, eps, us


def calibrate_measurement_heads(pf, observations, states, target_std: float = 1.2) -> float:
    """Randomly initialised measurement heads give almost flat log-likelihoods, which makes
    resampling trivial (identity) and flatters a benchmark.  Scale every unimodal head so the
    per-trajectory std of the log-likelihood over particles is ``target_std``
    (ESS/M = exp(-std^2) ~ 0.24 for log-normal weights).  Works on any filter exposing the
    reference's module layout (``measurement_model[.measurement_models[i]].shared_layers[4]``)."""
    meas = pf.measurement_model
    subs = list(getattr(meas, "measurement_models", [meas]))
    scale_used = 1.0
    with torch.no_grad():
        for m in subs:
            ll = m(states=states, observations=observations)
            std = float(ll.std(dim=1).mean())
            s = target_std / max(std, 1e-6)
            head = m.shared_layers[4]
            head.weight.mul_(s)
            head.bias.mul_(s)
            scale_used = s
    return scale_used


def stabilise_dynamics(filter_model, gain: float = 2e-3) -> None:
    """Randomly initialised dynamics networks are expanding maps: ``x' = x + dir(x) * gate`` with
    ``|dir(x)| ~ c |x|`` multiplies the state by ``(1 + c)`` per step, so an untrained filter
    leaves any finite range after a few hundred steps (|x| ~ 9e6 at step 400, fp32 and f16x3
    alike).  Scaling the direction rows of the dynamics head by ``gain`` keeps the same
    arithmetic per step (no work is removed) and bounds the growth to ``exp(gain * c * steps)``,
    so a benchmark run of any practical length stays finite.  Works on any filter exposing the
    reference's module layout (``dynamics_model.shared_layers[-1]``; fused EKFs: every
    ``filter_models[k].dynamics_model``)."""
    subs = list(getattr(filter_model, "filter_models", [filter_model]))
    with torch.no_grad():
        for f in subs:
            head = f.dynamics_model.shared_layers[-1]
            d = head.out_features - 1  # rows 0..d-1: direction, row d: gate logit
            head.weight[:d].mul_(gain)
            head.bias[:d].mul_(gain)

