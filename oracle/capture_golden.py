"""Generate ``tests/golden/*.npz`` from the REFERENCE's own crossmodal package.

Runs in the build container only (``/root/reference`` does not exist on the GPU box, and
no reference source or bytecode travels -- only the vectors this script writes).  The
reference's ``crossmodal`` package imports ``torchfilter`` and ``fannypack``; those names
are resolved to this repo's restatements (``oracle.tf`` / ``oracle.fp``), so what the
vectors pin is the reference's *crossmodal layer* (models + fusion math)
and the reference's RMSE arithmetic (H1, ``crossmodal/eval_helpers.py:149-160``).

    python -m oracle.capture_golden            # rewrites tests/golden/{door,push,eval}.npz
"""
import os
import sys
import warnings

import numpy as np
import torch

REFERENCE = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference_crossmodal():
    import oracle.fp as fp
    import oracle.tf as tf

    alias = {
        "torchfilter": tf, "torchfilter.base": tf.base, "torchfilter.filters": tf.filters,
        "torchfilter.types": tf.types, "torchfilter.data": tf.data, "torchfilter.train": tf.train,
        "fannypack": fp, "fannypack.nn": fp.nn, "fannypack.nn.resblocks": fp.nn.resblocks,
        "fannypack.utils": fp.utils, "fannypack.data": fp.data,
    }
    sys.modules.update(alias)
    sys.path.insert(0, REFERENCE)
    sys.dont_write_bytecode = True  # never write into /root/reference
    import crossmodal  # noqa: E402

    return crossmodal


def capture_models(cm):
    from oracle import golden_cases as gc
    from oracle import models as om

    for tname, task in om.TASKS.items():
        inp = gc.make_inputs(task)
        blob = {f"input/{k}": v for k, v in inp.items()}
        for case in gc.CASES:
            if tname not in case.tasks:
                continue
            for n, m in case.shapes:
                torch.manual_seed(0)
                ref_model = case.ref(cm, task)
                # same keys <=> same weights: fail early if the oracle drifts
                ora_keys = set(case.make(task).state_dict().keys())
                ref_keys = set(ref_model.state_dict().keys())
                assert ora_keys == ref_keys, (case.name, sorted(ora_keys ^ ref_keys)[:8])
                out = gc.run_case(case, ref_model, task, inp, n, m)
                for k, v in out.items():
                    assert np.all(np.isfinite(v) | np.isneginf(v)), (case.name, k)
                    blob[f"{gc.case_key(case, tname, n, m)}/{k}"] = v
        path = os.path.join(OUT, f"{tname}.npz")
        np.savez_compressed(path, **blob)
        print(f"wrote {path}: {len(blob)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def capture_eval(cm):
    """H1: drive the reference's ``run_eval`` with a replay filter and synthetic
    trajectories; store its RMSE outputs."""
    from oracle import golden_cases as gc
    from oracle import models as om
    import oracle.tf as tf

    class Replay(tf.base.Filter):
        def __init__(self, d, pred):
            super().__init__(state_dim=d)
            self.anchor = torch.nn.Parameter(torch.zeros(1))
            self.pred = pred

        def initialize_beliefs(self, *, mean, covariance):
            self.init_mean = mean

        def forward_loop(self, *, observations, controls):
            return torch.from_numpy(self.pred)

    blob = {}
    for tname, task in om.TASKS.items():
        true, pred = gc.make_eval_inputs(task)
        T, N, d = true.shape
        trajs = [
            tf.types.TrajectoryNumpy(true[:, n], {"gripper_pos": np.zeros((T, 3), np.float32)},
                                     np.zeros((T, 7), np.float32))
            for n in range(N)
        ]
        Task = cm.tasks.DoorTask if tname == "door" else cm.tasks.PushTask
        eh = cm.eval_helpers
        eh.filter_model = Replay(d, pred)
        eh.task = type("T", (), {"get_eval_trajectories": staticmethod(lambda **kw: trajs)})
        eh.dataset_args = {}
        # the final unit-conversion branch compares ``task`` with the real task classes
        real_task = Task
        fake = eh.task
        orig_globals = eh.run_eval.__globals__
        orig_globals["task"] = real_task
        saved = real_task.get_eval_trajectories
        real_task.get_eval_trajectories = classmethod(lambda cls, **kw: trajs)
        try:
            res = eh.run_eval()
        finally:
            real_task.get_eval_trajectories = saved
        blob[f"{tname}/true"] = true
        blob[f"{tname}/pred"] = pred
        for k, v in res.items():
            blob[f"{tname}/{k}"] = np.asarray(v, dtype=np.float64)
    path = os.path.join(OUT, "eval.npz")
    np.savez_compressed(path, **blob)
    print(f"wrote {path}: {len(blob)} arrays")


def synthetic_recordings(seed: int = 7, T: int = 7):
    """Raw trajectories shaped like the three HDF5 recordings (keys and shapes read off
    ``tasks/_door.py:137-177`` and ``tasks/_push.py:160-213``), two per dataset."""
    rng = np.random.RandomState(seed)
    f = np.float32
    contact = lambda: (rng.uniform(size=(T,)) > 0.4).astype(f)
    coarse = lambda lo, hi, shape: (np.round(rng.uniform(lo, hi, shape) * 4) / 4).astype(f)  # few levels: small fixture
    out = {"door": [], "push": [], "push-kloss": []}
    for _ in range(2):
        out["door"].append({
            "object-state": rng.standard_normal((T, 5)).astype(f), "eef_pos": rng.standard_normal((T, 3)).astype(f),
            "ee-force-obs": (10 * rng.standard_normal((T, 3))).astype(f), "ee-torque-obs": rng.standard_normal((T, 3)).astype(f),
            "contact-obs": contact(), "image": coarse(-1, 1, (T, 64, 64))})
        out["push"].append({
            "object-state": rng.standard_normal((T, 4)).astype(f), "Cylinder0_pos": rng.standard_normal((T, 3)).astype(f),
            "eef_pos": rng.standard_normal((T, 3)).astype(f), "force": rng.standard_normal((T, 6)).astype(f),
            "contact": contact(), "image": coarse(-1, 1, (T, 32, 32))})
        out["push-kloss"].append({
            "pos": rng.standard_normal((T, 3)).astype(f), "tip": rng.standard_normal((T, 3)).astype(f),
            "force": rng.standard_normal((T, 3)).astype(f), "contact": contact(),
            "image": coarse(0, 1, (T, 32, 32, 3))})
    return out


LOADER_CASES = [
    ("default", {}),
    ("no_vision", {"use_vision": False}),
    ("no_proprioception", {"use_proprioception": False}),
    ("no_haptics", {"use_haptics": False}),
    ("sequential3", {"sequential_image_rate": 3}),
    ("start2", {"start_timestep": 2}),
    ("blackout", {"image_blackout_ratio": 0.4}),
]


def capture_loaders(cm):
    """Dataset layer: the reference's ``_load_trajectories`` of both tasks
    run on synthetic recordings.  ``fannypack.data.TrajectoriesFile`` / ``cached_drive_file``
    (Drive download + HDF5 read) are replaced by an in-memory list for the duration of the call;
    everything after the file read is the reference's code."""
    import copy

    import oracle.fp.data as fpdata

    raws = synthetic_recordings()
    blob = {}
    for ds, recs in raws.items():
        for i, r in enumerate(recs):
            for k, v in r.items():
                blob[f"raw/{ds}/{i}/{k}"] = v

    class Files:
        def __init__(self, recs):
            self.recs = recs

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __iter__(self):
            return iter(copy.deepcopy(self.recs))  # the reference mutates what it reads

    saved = fpdata.TrajectoriesFile, fpdata.cached_drive_file
    try:
        fpdata.cached_drive_file = lambda name, url: name
        for ds, module, fname, extra in (("door", cm.tasks._door, "panda_door_pull_10.hdf5", {}),
                                         ("push", cm.tasks._push, "gentle_push_10.hdf5", {}),
                                         ("push-kloss", cm.tasks._push, "kloss_val.hdf5", {"kloss_dataset": True})):
            fpdata.TrajectoriesFile = lambda name, ds=ds: Files(raws[ds])
            for tag, args in LOADER_CASES:
                np.random.seed(0)  # blackout masks come from numpy's global RNG
                trajs = module._load_trajectories(fname, **args, **extra)
                assert len(trajs) == 2
                for i, t in enumerate(trajs):
                    states, obs, controls = t
                    blob[f"{ds}/{tag}/{i}/states"] = np.asarray(states)
                    blob[f"{ds}/{tag}/{i}/controls"] = np.asarray(controls)
                    for k in ("image", "gripper_pos", "gripper_sensors"):
                        blob[f"{ds}/{tag}/{i}/{k}"] = np.asarray(obs[k])
    finally:
        fpdata.TrajectoriesFile, fpdata.cached_drive_file = saved
    path = os.path.join(OUT, "loaders.npz")
    np.savez_compressed(path, **blob)
    print(f"wrote {path}: {len(blob)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


def main():
    warnings.filterwarnings("ignore")
    torch.set_num_threads(4)
    os.makedirs(OUT, exist_ok=True)
    cm = import_reference_crossmodal()
    capture_models(cm)
    capture_eval(cm)
    capture_loaders(cm)


if __name__ == "__main__":
    main()
