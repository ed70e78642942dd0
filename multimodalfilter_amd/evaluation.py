"""Evaluation harness: the role of ``/root/reference/crossmodal/eval_helpers.py:70-217`` on
device-resident ``(T, N, ...)`` batches, plus the multi-GPU reduction of its error statistic.
"""
from typing import Dict

import numpy as np
import torch

from .task_models import DOOR, PUSH

START_TRUNCATION = 30


def run_filter(filter_model, traj: Dict[str, torch.Tensor], *, initial_cov_scale: float = 0.1,
               measurement_initialize: bool = False) -> torch.Tensor:
    """Initialise the belief at ``states[0]`` with covariance ``0.1 I`` (or from the first
    observation, ``eval_helpers.py:116-131``) and filter ``[1:]`` (``:139-142``)."""
    states = traj["states"]
    T1, N, d = states.shape
    obs = {k: traj[k] for k in ("image", "gripper_pos", "gripper_sensors")}
    with torch.no_grad():
        if measurement_initialize and hasattr(filter_model, "measurement_initialize_beliefs"):
            filter_model.measurement_initialize_beliefs({k: v[0] for k, v in obs.items()})
        else:
            cov = (torch.eye(d, device=states.device) * initial_cov_scale)[None].expand(N, d, d)
            filter_model.initialize_beliefs(mean=states[0], covariance=cov)
        return filter_model.forward_loop(observations={k: v[1:] for k, v in obs.items()},
                                         controls=traj["controls"][1:])


def per_trajectory_mse(predicted: torch.Tensor, true: torch.Tensor,
                       start: int = START_TRUNCATION) -> torch.Tensor:
    """``(T, N, d)`` x2 -> ``(N, d)``: mean over time of the squared error after a burn-in
    (``eval_helpers.py:149-157``)."""
    err = predicted[start:] - true[start:]
    return torch.mean(err ** 2, dim=0)


def raw_rmse(per_batch_mse: torch.Tensor) -> np.ndarray:
    """``sqrt(mean_N)`` per state dimension (``eval_helpers.py:160``)."""
    return np.sqrt(np.mean(per_batch_mse.detach().cpu().numpy(), axis=0))


def task_metrics(task_name: str, rmse: np.ndarray) -> Dict[str, float]:
    """Unit conversion of ``eval_helpers.py:166-177`` (door) and ``:192-203`` (push)."""
    spec = DOOR if task_name == "door" else PUSH
    scaled = rmse * np.array(spec.rmse_scale)
    out = {"raw_rmse": [float(x) for x in rmse]}
    if task_name == "door":
        out["theta_rmse_deg"] = float(scaled[0] * 180.0 / np.pi)
        out["x_rmse_cm"] = float(scaled[1] * 100.0)
        out["y_rmse_cm"] = float(scaled[2] * 100.0)
    else:
        out["x_rmse_cm"] = float(scaled[0] * 100.0)
        out["y_rmse_cm"] = float(scaled[1] * 100.0)
    return out
