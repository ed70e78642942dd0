"""``torchfilter.data`` restated: the three datasets the reference's curricula build
(``crossmodal/train_helpers.py:39,63,83,110,143``).  TEST INFRASTRUCTURE (``oracle/__init__.py``).

The package is absent from ``/root/reference`` and un-pinned (``setup.py:12-15``); what follows is
its published behaviour, anchored on the reference's call sites -- PARITY UNPINNED:

* ``SingleStepDataset(trajectories=)``: every consecutive pair of a trajectory as
  ``(initial_state x_t, next_state x_{t+1}, observation o_{t+1}, control u_{t+1})``.
* ``SubsequenceDataset(trajectories=, subsequence_length=)``: consecutive, non-overlapping pieces
  of ``subsequence_length`` steps (the tail that does not fill a piece is dropped), each a
  ``(states (L, d), observations {(L, ...)}, controls (L, 7))`` triple.
* ``ParticleFilterMeasurementDataset(trajectories=, covariance=, samples_per_pair=)``: for every
  ``(state, observation)`` pair, ``samples_per_pair`` perturbed states -- the first half drawn from
  ``N(state, covariance)``, the second half from the wider ``N(state, 5 covariance)`` -- each with
  the target ``log N(noisy_state; state, covariance)``.  Randomness is explicit (``seed``) where
  upstream uses numpy's global state.
"""
import math

import numpy as np
import torch


class SingleStepDataset(torch.utils.data.Dataset):
    def __init__(self, *, trajectories):
        self.samples = []
        for states, observations, controls in trajectories:
            T = len(states)
            for t in range(T - 1):
                self.samples.append((states[t], states[t + 1], {k: v[t + 1] for k, v in observations.items()},
                                     controls[t + 1]))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        return self.samples[index]


class SubsequenceDataset(torch.utils.data.Dataset):
    def __init__(self, *, trajectories, subsequence_length: int):
        L = subsequence_length
        self.samples = []
        for states, observations, controls in trajectories:
            for s in range(0, len(states) - L + 1, L):
                self.samples.append((states[s:s + L], {k: v[s:s + L] for k, v in observations.items()}, controls[s:s + L]))

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        return self.samples[index]


def gaussian_log_pdf(x: np.ndarray, mean: np.ndarray, covariance: np.ndarray) -> np.ndarray:
    """``log N(x; mean, covariance)`` for rows of ``x`` / ``mean`` (what upstream evaluates with
    ``scipy.stats.multivariate_normal.logpdf``)."""
    d = covariance.shape[0]
    e = np.asarray(x, dtype=np.float64) - np.asarray(mean, dtype=np.float64)
    sol = np.linalg.solve(covariance.astype(np.float64), e.T).T
    _, logdet = np.linalg.slogdet(covariance.astype(np.float64))
    return -0.5 * (np.sum(e * sol, axis=-1) + d * math.log(2.0 * math.pi) + logdet)


class ParticleFilterMeasurementDataset(torch.utils.data.Dataset):
    FAR_SCALE = 5.0  # covariance multiplier of the "far" half of the samples

    def __init__(self, *, trajectories, covariance: np.ndarray, samples_per_pair: int, seed: int = 0):
        self.covariance = np.asarray(covariance, dtype=np.float64)
        self.samples_per_pair = samples_per_pair
        self.pairs = []
        for states, observations, _controls in trajectories:
            for t in range(len(states)):
                self.pairs.append((states[t], {k: v[t] for k, v in observations.items()}))
        self.rng = np.random.RandomState(seed)

    def __len__(self):
        return len(self.pairs) * self.samples_per_pair

    def __getitem__(self, index):
        state, observation = self.pairs[index // self.samples_per_pair]
        near = index % self.samples_per_pair < self.samples_per_pair * 0.5
        cov = self.covariance if near else self.covariance * self.FAR_SCALE
        noisy = self.rng.multivariate_normal(mean=state, cov=cov).astype(np.float32)
        target = np.float32(gaussian_log_pdf(noisy[None], np.asarray(state)[None], self.covariance)[0])
        return noisy, observation, target
