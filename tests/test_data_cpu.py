"""Dataset layer (``multimodalfilter_amd/data.py``) on synthetic recordings: the transformations
of ``tasks/_door.py:72-313`` / ``tasks/_push.py:97-416`` and the batching of
``eval_helpers.py:84-106`` / ``train_helpers.py:141-151``."""
import os

import numpy as np
import pytest
import torch

from multimodalfilter_amd import data


def _raw_door(T, rng):
    return {"object-state": rng.normal(size=(T, 5)).astype(np.float32),
            "eef_pos": rng.normal(size=(T, 3)).astype(np.float32),
            "ee-force-obs": rng.normal(size=(T, 3)).astype(np.float32),
            "ee-torque-obs": rng.normal(size=(T, 3)).astype(np.float32),
            "contact-obs": (rng.uniform(size=(T,)) > 0.5).astype(np.float32),
            "image": rng.uniform(size=(T, 64, 64)).astype(np.float32)}


def test_door_trajectory_fields_controls_and_normalisation():
    rng = np.random.default_rng(0)
    T = 9
    raw = _raw_door(T, rng)
    keep = {k: v.copy() for k, v in raw.items()}
    tr = data.trajectory_from_raw(raw, data.DOOR)
    n = data.DOOR.norm
    # states = (theta, hinge x, hinge y) = object-state columns 1, 3, 4, z-scored
    want_states = (keep["object-state"][:, [1, 3, 4]] - n["states"][0]) / n["states"][1]
    np.testing.assert_allclose(tr.states, want_states, rtol=1e-6)
    # sensors = force | torque | contact
    sens = np.concatenate([keep["ee-force-obs"], keep["ee-torque-obs"], keep["contact-obs"][:, None]], axis=1)
    np.testing.assert_allclose(tr.observations["gripper_sensors"], (sens - n["gripper_sensors"][0]) / n["gripper_sensors"][1], rtol=1e-6)
    np.testing.assert_allclose(tr.observations["gripper_pos"], (keep["eef_pos"] - n["gripper_pos"][0]) / n["gripper_pos"][1], rtol=1e-6)
    # controls = previous position (first repeated), delta, contact
    prev = np.concatenate([keep["eef_pos"][:1], keep["eef_pos"][:-1]])
    ctrl = np.concatenate([prev, keep["eef_pos"] - prev, keep["contact-obs"][:, None]], axis=1)
    np.testing.assert_allclose(tr.controls, (ctrl - n["controls"][0]) / n["controls"][1], rtol=1e-5)
    assert float(np.abs(tr.controls[0, 3:6] * n["controls"][1][0, 3:6] + n["controls"][0][0, 3:6]).max()) < 1e-6
    # image: every second pixel, all frames kept by default
    np.testing.assert_array_equal(tr.observations["image"], keep["image"][:, ::2, ::2])
    assert tr.states.dtype == tr.controls.dtype == np.float32
    assert raw["eef_pos"] is not tr.observations["gripper_pos"]  # inputs are not modified
    np.testing.assert_array_equal(raw["eef_pos"], keep["eef_pos"])


def test_image_masks_modalities_and_start_timestep():
    rng = np.random.default_rng(1)
    T = 12
    raw = _raw_door(T, rng)
    seq = data.trajectory_from_raw(raw, data.DOOR, sequential_image_rate=5)
    lit = [t for t in range(T) if np.abs(seq.observations["image"][t]).sum() > 0]
    assert lit == [0, 5, 10]
    none = data.trajectory_from_raw(raw, data.DOOR, use_vision=False)
    assert float(np.abs(none.observations["image"]).sum()) == 0.0
    a = data.trajectory_from_raw(raw, data.DOOR, image_blackout_ratio=0.6, rng=np.random.default_rng(7))
    b = data.trajectory_from_raw(raw, data.DOOR, image_blackout_ratio=0.6, rng=np.random.default_rng(7))
    np.testing.assert_array_equal(a.observations["image"], b.observations["image"])  # explicit randomness
    dark = sum(float(np.abs(a.observations["image"][t]).sum()) == 0.0 for t in range(T))
    assert 0 < dark < T
    off = data.trajectory_from_raw(raw, data.DOOR, use_proprioception=False, use_haptics=False, start_timestep=4)
    n = data.DOOR.norm
    assert off.states.shape == (T - 4, 3) and off.observations["image"].shape == (T - 4, 32, 32)
    np.testing.assert_allclose(off.observations["gripper_pos"], np.broadcast_to(-n["gripper_pos"][0] / n["gripper_pos"][1], (T - 4, 3)), rtol=1e-6)
    np.testing.assert_allclose(off.observations["gripper_sensors"][0], (-n["gripper_sensors"][0] / n["gripper_sensors"][1])[0], rtol=1e-6)
    assert float(np.abs(off.controls).sum()) > 0  # controls keep the real end-effector motion


def test_push_datasets_field_layouts():
    rng = np.random.default_rng(2)
    T = 6
    mj = {"Cylinder0_pos": rng.normal(size=(T, 3)).astype(np.float32), "eef_pos": rng.normal(size=(T, 3)).astype(np.float32),
          "force": rng.normal(size=(T, 6)).astype(np.float32), "contact": np.ones(T, dtype=np.float32),
          "image": rng.uniform(size=(T, 32, 32)).astype(np.float32)}
    tr = data.trajectory_from_raw(mj, data.PUSH_MUJOCO)
    n = data.PUSH_MUJOCO.norm
    np.testing.assert_allclose(tr.states, (mj["Cylinder0_pos"][:, :2] - n["states"][0]) / n["states"][1], rtol=1e-6)
    assert tr.observations["gripper_sensors"].shape == (T, 7) and tr.controls.shape == (T, 7)
    kl = {"pos": rng.normal(size=(T, 3)).astype(np.float32), "tip": rng.normal(size=(T, 3)).astype(np.float32),
          "force": rng.normal(size=(T, 3)).astype(np.float32), "contact": np.zeros(T, dtype=np.float32),
          "image": rng.uniform(size=(T, 32, 32, 3)).astype(np.float32)}
    tk = data.trajectory_from_raw(kl, data.PUSH_KLOSS)
    nk = data.PUSH_KLOSS.norm
    np.testing.assert_allclose(tk.states, (kl["pos"][:, [0, 2]] - nk["states"][0]) / nk["states"][1], rtol=1e-6)
    sens = np.zeros((T, 7), dtype=np.float32)
    sens[:, :3] = kl["force"]
    np.testing.assert_allclose(tk.observations["gripper_sensors"], (sens - nk["gripper_sensors"][0]) / nk["gripper_sensors"][1], rtol=1e-6)
    np.testing.assert_allclose(tk.observations["image"], kl["image"].mean(-1), rtol=1e-6)


def test_stack_and_subsequence_batches():
    rng = np.random.default_rng(3)
    trajs = [data.trajectory_from_raw(_raw_door(T, rng), data.DOOR) for T in (11, 9, 14)]
    batch = data.stack_trajectories(trajs, "cpu")
    assert batch["states"].shape == (9, 3, 3) and batch["image"].shape == (9, 3, 32, 32)
    assert batch["controls"].shape == (9, 3, 7) and batch["states"].dtype == torch.float32
    np.testing.assert_array_equal(batch["gripper_pos"][:, 1].numpy(), trajs[1].observations["gripper_pos"][:9])
    loader = data.SubsequenceBatcher(trajs, subsequence_length=4, batch_size=2, device="cpu", seed=5)
    assert loader.num_pieces == 2 + 2 + 3 and len(loader) == 3
    seen = []
    for b in loader:
        assert b["states"].shape == (4, 2, 3) and b["image"].shape == (4, 2, 32, 32)
        seen += [tuple(np.round(b["states"][0, k].numpy(), 5)) for k in range(2)]
    assert len(set(seen)) == 6  # six distinct pieces in an epoch, one dropped
    first = [tuple(np.round(b["states"][0, 0].numpy(), 5)) for b in data.SubsequenceBatcher(
        trajs, subsequence_length=4, batch_size=2, device="cpu", seed=5)]
    assert first == seen[::2]  # seeded order


# ------------------------------------------------------------------ loader parity with the reference
_LOADER_CASES = {
    "default": {}, "no_vision": {"use_vision": False}, "no_proprioception": {"use_proprioception": False},
    "no_haptics": {"use_haptics": False}, "sequential3": {"sequential_image_rate": 3},
    "start2": {"start_timestep": 2}, "blackout": {"image_blackout_ratio": 0.4},
}


@pytest.mark.parametrize("ds,spec", [("door", data.DOOR), ("push", data.PUSH_MUJOCO), ("push-kloss", data.PUSH_KLOSS)])
@pytest.mark.parametrize("case", sorted(_LOADER_CASES))
def test_trajectory_from_raw_matches_the_reference_loaders(golden_dir, ds, spec, case):
    """``tests/golden/loaders.npz``: the REFERENCE's ``_load_trajectories`` (``tasks/_door.py:72-313``,
    ``tasks/_push.py:97-416``) run on synthetic recordings by ``oracle/capture_golden.py``.  Field
    selection, masking, the controls construction (including the aliasing quirk under
    ``use_proprioception=False``), the z-scoring constants and ``start_timestep`` must agree
    exactly; blackout masks replay numpy's global stream through a ``RandomState``."""
    z = np.load(os.path.join(golden_dir, "loaders.npz"))
    rng = np.random.RandomState(0)
    for i in range(2):
        raw = {k.split("/", 3)[3]: z[k] for k in z.files if k.startswith(f"raw/{ds}/{i}/")}
        t = data.trajectory_from_raw(raw, spec, rng=rng, **_LOADER_CASES[case])
        want = {k.rsplit("/", 1)[1]: z[k] for k in z.files if k.startswith(f"{ds}/{case}/{i}/")}
        assert set(want) == {"states", "controls", "image", "gripper_pos", "gripper_sensors"}
        np.testing.assert_allclose(t.states, want["states"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(t.controls, want["controls"], rtol=1e-6, atol=1e-6)
        for k in ("image", "gripper_pos", "gripper_sensors"):
            np.testing.assert_allclose(t.observations[k], want[k], rtol=1e-6, atol=1e-6, err_msg=k)
    if case == "no_proprioception":  # the quirk, and its documented opt-out
        raw = {k.split("/", 3)[3]: z[k] for k in z.files if k.startswith(f"raw/{ds}/0/")}
        kept = data.trajectory_from_raw(raw, spec, use_proprioception=False, reference_aliasing=False)
        full = data.trajectory_from_raw(raw, spec)
        np.testing.assert_allclose(kept.controls, full.controls)


def test_load_hdf5_reads_groups_in_order(monkeypatch, tmp_path):
    """``load_hdf5`` against a stand-in ``h5py`` (the real package is absent): one group per
    trajectory, datasets by key, numeric group order, ``max_trajectories``, dataset arguments."""
    import sys
    import types

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "loaders.npz"))
    raws = [{k.split("/", 3)[3]: z[k] for k in z.files if k.startswith(f"raw/door/{i}/")} for i in range(2)]

    class FakeFile(dict):
        def __init__(self, path, mode):
            assert mode == "r"
            super().__init__({"trajectory10": raws[1], "trajectory2": raws[0]})

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=FakeFile))
    out = data.load_hdf5(str(tmp_path / "x.hdf5"), data.DOOR, start_timestep=1)
    assert len(out) == 2
    np.testing.assert_allclose(out[0].states, z["door/default/0/states"][1:], rtol=1e-6, atol=1e-7)  # "2" before "10"
    np.testing.assert_allclose(out[1].states, z["door/default/1/states"][1:], rtol=1e-6, atol=1e-7)
    assert len(data.load_hdf5(str(tmp_path / "x.hdf5"), data.DOOR, max_trajectories=1)) == 1


# ------------------------------------------------------------------ pre-training datasets
def _toy_trajectories(n=3, T=6, d=3, seed=1):
    from multimodalfilter_amd.types import TrajectoryNumpy

    rng = np.random.RandomState(seed)
    f = np.float32
    return [TrajectoryNumpy(rng.standard_normal((T + i, d)).astype(f),
                            {"image": rng.standard_normal((T + i, 32, 32)).astype(f),
                             "gripper_pos": rng.standard_normal((T + i, 3)).astype(f),
                             "gripper_sensors": rng.standard_normal((T + i, 7)).astype(f)},
                            rng.standard_normal((T + i, 7)).astype(f)) for i in range(n)]


def test_single_step_batcher_matches_the_oracle_dataset():
    from oracle.tf.data import SingleStepDataset

    trajs = _toy_trajectories()
    ds = SingleStepDataset(trajectories=trajs)
    b = data.SingleStepBatcher(trajs, batch_size=4, device="cpu", seed=3)
    assert b.count == len(ds) == sum(len(t.states) - 1 for t in trajs)
    for i in (0, 5, len(ds) - 1):  # same pairs, same order before shuffling
        x0, x1, obs, u = ds[i]
        np.testing.assert_array_equal(b.data["initial_states"][i].numpy(), x0)
        np.testing.assert_array_equal(b.data["next_states"][i].numpy(), x1)
        np.testing.assert_array_equal(b.data["controls"][i].numpy(), u)
        np.testing.assert_array_equal(b.data["gripper_pos"][i].numpy(), obs["gripper_pos"])
    seen = torch.cat([batch["initial_states"] for batch in b])
    assert seen.shape[0] == b.count and len(b) == -(-b.count // 4)  # every pair once per epoch


def test_particle_filter_measurement_batcher_targets_and_spread():
    from oracle.tf.data import ParticleFilterMeasurementDataset, gaussian_log_pdf

    trajs = _toy_trajectories(n=2, T=40)
    d, S = 3, 10
    cov = np.diag([0.1, 0.2, 0.05])
    b = data.ParticleFilterMeasurementBatcher(trajs, covariance=cov, samples_per_pair=S, batch_size=16, device="cpu", seed=4)
    states = np.concatenate([t.states for t in trajs]).repeat(S, axis=0)
    noisy = b.data["noisy_states"].numpy()
    # targets: log N(noisy; state, covariance), the closed form the oracle dataset evaluates
    np.testing.assert_allclose(b.data["log_likelihoods"].numpy(), gaussian_log_pdf(noisy, states, cov), rtol=2e-5, atol=2e-5)
    # first half of every pair's samples ~ N(state, cov), second half ~ N(state, 5 cov)
    e = (noisy - states).reshape(-1, S, d)
    near, far = e[:, : S // 2].reshape(-1, d), e[:, S // 2:].reshape(-1, d)
    np.testing.assert_allclose(near.var(0), np.diag(cov), rtol=0.25)
    np.testing.assert_allclose(far.var(0), 5 * np.diag(cov), rtol=0.25)
    # observations repeat per sample
    np.testing.assert_array_equal(b.data["gripper_pos"][:S].numpy(), np.repeat(trajs[0].observations["gripper_pos"][:1], S, 0))
    # the oracle dataset has the same structure (its own RNG): half near, half far, same target formula
    ds = ParticleFilterMeasurementDataset(trajectories=trajs, covariance=cov, samples_per_pair=S, seed=0)
    assert len(ds) == b.count
    n, o, ll = ds[7]
    np.testing.assert_allclose(ll, gaussian_log_pdf(n[None], trajs[0].states[0][None], cov)[0], rtol=1e-5)


@pytest.mark.parametrize("compress", [False, True])
def test_load_hdf5_reads_a_real_hdf5_recording(tmp_path, compress):
    """A REAL HDF5 file in ``fannypack.data.TrajectoriesFile`` layout (groups ``trajectory<i>``, one dataset per
    key; contiguous, and chunked + deflate as ``compress=True`` stores them), written with the HDF5 C library and
    read back by ``data.load_hdf5`` (``h5py`` if installed, else ``hdf5lite`` on the same library): the
    normalised trajectories equal the REFERENCE loader's outputs for the same raw recordings
    (``tests/golden/loaders.npz``)."""
    from multimodalfilter_amd import hdf5lite

    try:
        hdf5lite.lib()
    except (ImportError, OSError) as e:
        pytest.skip(f"no libhdf5 in this environment: {e}")
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "loaders.npz"))
    for task, spec in (("door", data.DOOR), ("push", data.PUSH_MUJOCO)):
        n = len({k.split("/")[2] for k in z.files if k.startswith(f"raw/{task}/")})
        raws = [{k.split("/", 3)[3]: z[k] for k in z.files if k.startswith(f"raw/{task}/{i}/")} for i in range(n)]
        path = str(tmp_path / f"{task}.hdf5")
        # group names as TrajectoriesFile writes them; lexicographic order would put 10 before 2
        names = [f"trajectory{i if i == 0 else i + 9}" for i in range(n)]
        hdf5lite.write_groups(path, dict(zip(names, raws)), compress=compress)
        with open(path, "rb") as fh:
            assert fh.read(8) == b"\x89HDF\r\n\x1a\n"
        got = data.load_hdf5(path, spec)
        assert len(got) == n
        for i, t in enumerate(got):
            np.testing.assert_allclose(t.states, z[f"{task}/default/{i}/states"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(t.controls, z[f"{task}/default/{i}/controls"], rtol=1e-6, atol=1e-6)
            for k in ("image", "gripper_pos", "gripper_sensors"):
                np.testing.assert_allclose(t.observations[k], z[f"{task}/default/{i}/{k}"], rtol=1e-6, atol=1e-6)
        assert len(data.load_hdf5(path, spec, max_trajectories=1)) == 1
