"""Dataset layer: raw recorded trajectories -> the normalised
``TrajectoryNumpy`` triples the reference's loaders produce, and GPU-resident ``(T, N, ...)``
batches for evaluation and training.

What it restates (host-side numpy; nothing here is on the timed path):

* ``/root/reference/crossmodal/tasks/_door.py:72-313`` and ``tasks/_push.py:97-416``
  (``_load_trajectories``): field selection, F/T + contact stacking, image masking by
  ``use_vision`` / ``sequential_image_rate`` / ``image_blackout_ratio``, controls =
  ``[previous end-effector position, position delta, contact]``, z-scoring with the datasets'
  constants, ``start_timestep``.  The three datasets are described declaratively (``DatasetSpec``)
  instead of by branches.
* ``/root/reference/crossmodal/eval_helpers.py:84-106``: list of trajectories -> one
  ``(T, N, ...)`` batch, truncated to the shortest trajectory (``stack_trajectories``).
* ``torchfilter.data.SubsequenceDataset`` as ``train_helpers.py:141-151`` uses it: every
  trajectory cut into consecutive length-``L`` pieces, shuffled, ``drop_last`` batches
  (``SubsequenceBatcher``; keeps everything on the device, yields time-major batches for
  ``train.train_filter_step``).

* ``torchfilter.data.SingleStepDataset`` and ``ParticleFilterMeasurementDataset`` as
  ``train_helpers.py:39,83-87,110`` use them (``SingleStepBatcher``,
  ``ParticleFilterMeasurementBatcher``; published behaviour of the absent package, restated in
  ``oracle/tf/data.py`` too).

Reading the recordings themselves (``load_hdf5``) goes through ``h5py`` when it is installed and
otherwise through ``libhdf5`` directly (``hdf5lite.py``, ctypes): the container has no ``h5py`` and no
datasets (Drive URLs, ``tasks/_door.py:11-20``), but it has the HDF5 C library, so the tests write real
HDF5 recordings of the synthetic raw trajectories and read them back.  Everything after the file read is
pinned against the reference's own loaders (``tests/golden/loaders.npz``).
"""
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .types import TrajectoryNumpy

F32 = np.float32


def _row(*values) -> np.ndarray:
    return np.array([values], dtype=F32)


@dataclass(frozen=True)
class DatasetSpec:
    """Where a dataset keeps each quantity and how it is z-scored (mean, std per channel)."""
    name: str
    state_dim: int
    state_columns: Tuple[Tuple[str, int], ...]   # (raw key, column) per state dimension
    eef_key: str
    sensor_fields: Tuple[Tuple[str, Optional[int]], ...]  # (raw key, target column or None = append in order)
    contact_key: str
    image_is_rgb: bool
    image_stride: int
    norm: Dict[str, Tuple[np.ndarray, np.ndarray]]


# tasks/_door.py:137-147 (states), :154-166 (sensors), :177 (image ::2), :225-297 (constants)
DOOR = DatasetSpec(
    name="door", state_dim=3,
    state_columns=(("object-state", 1), ("object-state", 3), ("object-state", 4)),
    eef_key="eef_pos",
    sensor_fields=(("ee-force-obs", None), ("ee-torque-obs", None), ("contact-obs", None)),
    contact_key="contact-obs", image_is_rgb=False, image_stride=2,
    norm={
        "gripper_pos": (_row(0.37334135, -0.10821614, 1.5769919), _row(0.13496609, 0.14862472, 0.04533212)),
        "gripper_sensors": (_row(11.064128, -1.7103539, 28.303621, 0.06923943, 1.661722, -0.14174654, 0.63155425),
                            _row(36.36674, 18.355747, 58.651367, 1.8596123, 4.574878, 0.64844555, 0.48232532)),
        "states": (_row(0.64900873, -0.00079839, -0.00069189), _row(0.39479038, 0.05650279, 0.0565098)),
        "controls": (_row(3.7333974e-01, -1.0831217e-01, 1.5769361, 3.1821314e-06, 9.5862495e-05, 4.8311016e-05, 6.3155425e-01),
                     _row(0.134951, 0.14904341, 0.04531819, 0.00323106, 0.00411722, 0.00165688, 0.48232532)),
    })

# tasks/_push.py:166-173 (states), :183-199 (sensors), :210-213 (image), :328-395 (constants)
PUSH_MUJOCO = DatasetSpec(
    name="push", state_dim=2,
    state_columns=(("Cylinder0_pos", 0), ("Cylinder0_pos", 1)),
    eef_key="eef_pos",
    sensor_fields=(("force", None), ("contact", None)),
    contact_key="contact", image_is_rgb=False, image_stride=1,
    norm={
        "gripper_pos": (_row(0.46806443, -0.0017836, 0.88028437), _row(0.02410769, 0.02341035, 0.04018243)),
        "gripper_sensors": (_row(0.49182904, 0.045039989, -3.2791464, -0.0033874984, 0.011552566, -0.00084817986, 0.21303751),
                            _row(1.6152629, 1.666905, 1.9186896, 0.14219016, 0.14232528, 0.01675198, 0.40950698)),
        "states": (_row(0.4970164, -0.00916641), _row(0.0572766, 0.06118315)),
        "controls": (_row(0.46594709, -0.0025247163, 0.88094306, 0.0001293995, -5.4364675e-05, -0.00061112235, 0.22041667),
                     _row(0.02239027, 0.02356066, 0.0405312, 0.00054858, 0.0005754, 0.00046352, 0.41451886)),
    })

# tasks/_push.py:160-171 (states from "pos"[:, (0, 2)]), :189-192 (3 force channels + contact in
# column 6), :211 (RGB mean), :264-325 (constants)
PUSH_KLOSS = DatasetSpec(
    name="push-kloss", state_dim=2,
    state_columns=(("pos", 0), ("pos", 2)),
    eef_key="tip",
    sensor_fields=(("force", 0), ("contact", 6)),
    contact_key="contact", image_is_rgb=True, image_stride=1,
    norm={
        "gripper_pos": (_row(-0.00360131, 0.0, 0.00022349), _row(0.07005621, 1.0, 0.06883541)),
        "gripper_sensors": (_row(0.0304424347, 0.016132861, -0.000247517393, 0.0, 0.0, 0.0, 0.625842857),
                            _row(2.09539968, 2.0681382, 0.00373115, 1.0, 1.0, 1.0, 0.48390451)),
        "states": (_row(-0.00279736, -0.00027878), _row(0.06409658, 0.06649422)),
        "controls": (_row(-0.00355868486, 0.0, 0.000234369027, -4.26185595e-05, 0.0, -1.08724583e-05, 0.625842857),
                     _row(0.0693582, 1.0, 0.06810329, 0.01176415, 1.0, 0.0115694, 0.48390451)),
    })


def image_mask(timesteps: int, *, use_vision: bool = True, image_blackout_ratio: float = 0.0,
               sequential_image_rate: int = 1, rng: Optional[np.random.Generator] = None) -> np.ndarray:
    """``(T, 1, 1)`` 0/1 mask of the frames that are kept (``tasks/_door.py:181-197``): none
    without vision; every ``sequential_image_rate``-th frame; or each frame independently with
    probability ``1 - image_blackout_ratio`` (explicit ``rng`` instead of numpy's global state)."""
    assert 1 > image_blackout_ratio >= 0
    assert image_blackout_ratio == 0 or sequential_image_rate == 1
    mask = np.zeros((timesteps, 1, 1), dtype=F32)
    if not use_vision:
        return mask
    if image_blackout_ratio == 0.0:
        mask[::sequential_image_rate, 0, 0] = 1.0
        return mask
    rng = rng if rng is not None else np.random.default_rng()
    return (rng.uniform(size=(timesteps,)) > image_blackout_ratio).astype(F32).reshape((timesteps, 1, 1))


def controls_from_end_effector(eef_positions: np.ndarray, contact: np.ndarray) -> np.ndarray:
    """``[previous position (first repeated), position delta, contact]`` -> ``(T, 7)``
    (``tasks/_door.py:205-222``)."""
    prev = np.roll(eef_positions, shift=1, axis=0)
    prev[0] = eef_positions[0]
    return np.concatenate([prev, eef_positions - prev, contact[:, None]], axis=1).astype(F32)


def trajectory_from_raw(raw: Dict[str, np.ndarray], spec: DatasetSpec, *, use_vision: bool = True,
                        use_proprioception: bool = True, use_haptics: bool = True,
                        image_blackout_ratio: float = 0.0, sequential_image_rate: int = 1,
                        start_timestep: int = 0, rng: Optional[np.random.Generator] = None,
                        reference_aliasing: bool = True) -> TrajectoryNumpy:
    """One recorded trajectory (dict of ``(T, ...)`` arrays as stored in the HDF5 files) -> the
    normalised ``TrajectoryNumpy`` the reference's ``_load_trajectories`` appends
    (pinned by ``tests/golden/loaders.npz``: the reference's own loaders run on synthetic
    recordings, ``oracle/capture_golden.py``).

    ``reference_aliasing`` (quirk, default preserved): in the reference ``observations
    ["gripper_pos"]`` IS the raw end-effector array (``tasks/_door.py:150``,
    ``tasks/_push.py:178-181``), so ``use_proprioception=False`` zeroes the positions the controls
    are built from as well: controls become ``[0, 0, 0, 0, 0, 0, contact]`` before z-scoring.
    ``False`` keeps the real positions in the controls.  ``rng``: anything with
    ``uniform(size=)`` (``np.random.RandomState`` reproduces the reference's global-RNG draws)."""
    T = len(raw[spec.state_columns[0][0]])
    states = np.stack([raw[k][:, c] for k, c in spec.state_columns], axis=1).astype(F32)

    pos = np.array(raw[spec.eef_key], dtype=F32)
    assert pos.shape == (T, 3)
    sensors = np.zeros((T, 7), dtype=F32)
    col = 0
    for key, at in spec.sensor_fields:
        v = np.asarray(raw[key], dtype=F32)
        v = v[:, None] if v.ndim == 1 else v
        start = col if at is None else at
        sensors[:, start:start + v.shape[1]] = v
        col = start + v.shape[1]
    if all(at is None for _, at in spec.sensor_fields):
        assert col == 7, "force / torque / contact channels must fill the 7 sensor columns"

    image = np.asarray(raw["image"], dtype=F32)
    if spec.image_is_rgb:
        image = image.mean(axis=-1)
    image = image[:, ::spec.image_stride, ::spec.image_stride].copy()
    assert image.shape == (T, 32, 32)
    image *= image_mask(T, use_vision=use_vision, image_blackout_ratio=image_blackout_ratio,
                        sequential_image_rate=sequential_image_rate, rng=rng)

    # modalities that are switched off are zeroed BEFORE normalisation, as in the reference
    eef = np.array(raw[spec.eef_key], dtype=F32)
    if not use_proprioception:
        pos[:] = 0
        if reference_aliasing:
            eef[:] = 0
    if not use_haptics:
        sensors[:] = 0
    controls = controls_from_end_effector(eef, np.asarray(raw[spec.contact_key], dtype=F32))
    observations = {"gripper_pos": pos, "gripper_sensors": sensors, "image": image}
    for key, target in (("gripper_pos", pos), ("gripper_sensors", sensors), ("states", states), ("controls", controls)):
        mean, std = spec.norm[key]
        target -= mean
        target /= std
    return TrajectoryNumpy(states[start_timestep:], {k: v[start_timestep:] for k, v in observations.items()},
                           controls[start_timestep:])


def _trajectory_order(name: str) -> int:
    return int("".join(ch for ch in name if ch.isdigit()) or 0)


def load_hdf5(path: str, spec: DatasetSpec, *, max_trajectories: Optional[int] = None,
              **dataset_args) -> List[TrajectoryNumpy]:
    """Read a ``fannypack.data.TrajectoriesFile``-style HDF5 recording (one group per trajectory,
    one dataset per key; ``tasks/_door.py:121-126``) and normalise every trajectory as the reference's
    loader does.  With ``h5py`` when it is installed; otherwise through the HDF5 C library itself
    (``hdf5lite``: ``H5Fopen`` / ``H5Literate`` / ``H5Dread`` via ctypes -- what ``h5py`` wraps), so the
    read is a real one either way (``tests/test_data_cpu.py`` writes recordings with the same library,
    contiguous and chunked + deflate)."""
    try:
        import h5py
    except ImportError:
        h5py = None
    out = []
    if h5py is not None:
        with h5py.File(path, "r") as f:
            for name in sorted(f.keys(), key=_trajectory_order):
                if max_trajectories is not None and len(out) >= max_trajectories:
                    break
                out.append(trajectory_from_raw({k: np.array(v) for k, v in f[name].items()}, spec, **dataset_args))
        return out
    from . import hdf5lite

    groups = hdf5lite.read_groups(path)
    for name in sorted(groups, key=_trajectory_order):
        if max_trajectories is not None and len(out) >= max_trajectories:
            break
        out.append(trajectory_from_raw(groups[name], spec, **dataset_args))
    return out


def stack_trajectories(trajectories: Sequence[TrajectoryNumpy], device) -> Dict[str, torch.Tensor]:
    """List of trajectories -> ``{"states" (T, N, d), "controls" (T, N, 7), "image", "gripper_pos",
    "gripper_sensors" (T, N, ...)}`` on ``device``, truncated to the shortest trajectory
    (``eval_helpers.py:84-106``); the layout ``evaluation.run_filter`` and ``bench.py`` consume."""
    assert len(trajectories) > 0
    T = min(t.states.shape[0] for t in trajectories)
    to = lambda arrays: torch.from_numpy(np.stack([a[:T] for a in arrays], axis=1)).to(device=device, dtype=torch.float32)
    batch = {"states": to([t.states for t in trajectories]), "controls": to([t.controls for t in trajectories])}
    for key in ("image", "gripper_pos", "gripper_sensors"):
        batch[key] = to([t.observations[key] for t in trajectories])
    return batch


class SubsequenceBatcher:
    """Device-resident subsequence batches for ``train.train_filter_step``.

    Every trajectory is cut into consecutive pieces of ``subsequence_length`` steps (the tail that
    does not fill a piece is dropped), all pieces live on the device as one ``(L, P, ...)`` block,
    and an epoch is a seeded permutation of the pieces in ``drop_last`` batches of
    ``batch_size`` -- what ``DataLoader(SubsequenceDataset(...), shuffle=True, drop_last=True)``
    does in ``train_helpers.py:141-151``, minus the host round trip per batch."""

    def __init__(self, trajectories: Sequence[TrajectoryNumpy], *, subsequence_length: int, batch_size: int,
                 device, seed: int = 0):
        L = subsequence_length
        pieces = []
        for t in trajectories:
            for s in range(0, t.states.shape[0] - L + 1, L):
                pieces.append(TrajectoryNumpy(t.states[s:s + L], {k: v[s:s + L] for k, v in t.observations.items()},
                                              t.controls[s:s + L]))
        assert pieces, "no trajectory is as long as subsequence_length"
        self.data = stack_trajectories(pieces, device)
        self.batch_size = batch_size
        self.num_pieces = len(pieces)
        self._gen = torch.Generator(device="cpu").manual_seed(seed)

    def __len__(self) -> int:
        return self.num_pieces // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        order = torch.randperm(self.num_pieces, generator=self._gen)
        dev = self.data["states"].device
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size].to(dev)
            yield {k: v.index_select(1, idx) for k, v in self.data.items()}


class _DeviceBatcher:
    """Seeded, shuffled batches of a dict of device tensors that share dim 0."""

    def __init__(self, tensors: Dict[str, torch.Tensor], batch_size: int, seed: int, drop_last: bool):
        self.data = tensors
        self.count = next(iter(tensors.values())).shape[0]
        self.batch_size = batch_size
        self.drop_last = drop_last
        self._gen = torch.Generator(device="cpu").manual_seed(seed)

    def __len__(self) -> int:
        return self.count // self.batch_size if self.drop_last else -(-self.count // self.batch_size)

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        order = torch.randperm(self.count, generator=self._gen)
        dev = next(iter(self.data.values())).device
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size].to(dev)
            yield {k: v.index_select(0, idx) for k, v in self.data.items()}


class SingleStepBatcher(_DeviceBatcher):
    """``torchfilter.data.SingleStepDataset`` behind a shuffling loader, device-resident: every
    consecutive pair of every trajectory as ``initial_states x_t``, ``next_states x_{t+1}``,
    ``controls u_{t+1}`` and the observations ``o_{t+1}`` (``train_helpers.py:39-43,110-114``)."""

    def __init__(self, trajectories: Sequence[TrajectoryNumpy], *, batch_size: int, device, seed: int = 0):
        cat = lambda arrays: torch.from_numpy(np.concatenate(arrays, axis=0)).to(device=device, dtype=torch.float32)
        t = {"initial_states": cat([tr.states[:-1] for tr in trajectories]),
             "next_states": cat([tr.states[1:] for tr in trajectories]),
             "controls": cat([tr.controls[1:] for tr in trajectories])}
        for key in ("image", "gripper_pos", "gripper_sensors"):
            t[key] = cat([tr.observations[key][1:] for tr in trajectories])
        super().__init__(t, batch_size, seed, drop_last=False)


def gaussian_log_pdf(x: torch.Tensor, mean: torch.Tensor, covariance: torch.Tensor) -> torch.Tensor:
    """``log N(x; mean, covariance)`` row-wise, in fp64."""
    d = covariance.shape[0]
    e = (x - mean).double()
    sol = torch.linalg.solve(covariance.double(), e.t()).t()
    return -0.5 * ((e * sol).sum(-1) + d * np.log(2.0 * np.pi) + torch.logdet(covariance.double()))


class ParticleFilterMeasurementBatcher(_DeviceBatcher):
    """``torchfilter.data.ParticleFilterMeasurementDataset`` (``train_helpers.py:83-91``): for every
    ``(state, observation)`` pair ``samples_per_pair`` perturbed states -- the first half from
    ``N(state, covariance)``, the second half from ``N(state, 5 covariance)`` -- each with the
    regression target ``log N(noisy_state; state, covariance)``.  All draws are made once, on the
    CPU, from ``seed`` (upstream redraws from numpy's global RNG on every access)."""

    FAR_SCALE = 5.0

    def __init__(self, trajectories: Sequence[TrajectoryNumpy], *, covariance: np.ndarray, samples_per_pair: int,
                 batch_size: int, device, seed: int = 0):
        cat = lambda arrays: torch.from_numpy(np.concatenate(arrays, axis=0)).to(torch.float32)
        states = cat([tr.states for tr in trajectories])
        P, d = states.shape
        S = samples_per_pair
        cov = torch.as_tensor(np.asarray(covariance), dtype=torch.float64)
        g = torch.Generator(device="cpu").manual_seed(seed)
        eps = torch.randn((P, S, d), generator=g, dtype=torch.float64)
        scale = torch.where(torch.arange(S) < S * 0.5, 1.0, float(np.sqrt(self.FAR_SCALE))).double()
        noisy = states.double()[:, None, :] + (eps * scale[None, :, None]) @ torch.linalg.cholesky(cov).t()
        target = gaussian_log_pdf(noisy.reshape(P * S, d), states.double().repeat_interleave(S, dim=0), cov)
        rep = lambda x: x.repeat_interleave(S, dim=0).to(device)
        t = {"noisy_states": noisy.reshape(P * S, d).float().to(device), "log_likelihoods": target.float().to(device)}
        for key in ("image", "gripper_pos", "gripper_sensors"):
            t[key] = rep(cat([tr.observations[key] for tr in trajectories]))
        super().__init__(t, batch_size, seed + 1, drop_last=False)
