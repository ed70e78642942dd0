"""Dataset layer (``multimodalfilter_amd/data.py``) on synthetic recordings: the transformations
of ``tasks/_door.py:72-313`` / ``tasks/_push.py:97-416`` and the batching of
``eval_helpers.py:84-106`` / ``train_helpers.py:141-151``."""
import numpy as np
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
