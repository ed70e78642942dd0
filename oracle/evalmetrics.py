"""RMSE arithmetic of the reference's evaluation (H1).  TEST INFRASTRUCTURE.

Follows ``/root/reference/crossmodal/eval_helpers.py:149-160`` (drop the first 30 steps,
mean over time, mean over the batch, square root, per state dimension) and the unit
conversions at ``:166-177`` (door) and ``:192-203`` (push).  Pinned by
``tests/golden/eval.npz``, produced by the reference's own ``run_eval``.
"""
import numpy as np

from .models import TASKS

START_TRUNCATION = 30


def raw_rmse(predicted: np.ndarray, true: np.ndarray, start: int = START_TRUNCATION) -> np.ndarray:
    """``predicted``, ``true``: ``(T, N, d)`` aligned (truth already shifted by one step)."""
    err = np.asarray(predicted)[start:] - np.asarray(true)[start:]
    per_batch_mse = np.mean(err ** 2, axis=0)
    return np.sqrt(np.mean(per_batch_mse, axis=0))


def task_rmse(task_name: str, predicted, true) -> dict:
    task = TASKS[task_name]
    raw = raw_rmse(predicted, true)
    scaled = raw * np.array(task.rmse_scale)
    out = {"raw_rmse": [float(x) for x in raw]}
    if task_name == "door":
        out["theta_rmse_deg"] = float(scaled[0] * 180.0 / np.pi)
        out["x_rmse_cm"] = float(scaled[1] * 100.0)
        out["y_rmse_cm"] = float(scaled[2] * 100.0)
    else:
        out["x_rmse_cm"] = float(scaled[0] * 100.0)
        out["y_rmse_cm"] = float(scaled[1] * 100.0)
    return out
