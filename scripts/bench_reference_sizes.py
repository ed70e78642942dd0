"""The sizes the REFERENCE itself runs (not BASELINE's scaled-up configs), each with its CPU-oracle twin:

* evaluation -- ``eval_helpers.run_eval`` (``/root/reference/crossmodal/eval_helpers.py:125-142``) on the
  door particle filter: 300 particles in eval mode (``door_models/pf.py:24-27``), trajectories of up
  to 800 steps, a few dozen trajectories per batch:  32 x 300 x 800.
* end-to-end training -- ``train_helpers.train_e2e(subsequence_length=16, batch_size=32)``
  (``scripts/door_task/train_door.py:63-71``), 30 particles in train mode:  32 x 30 x 16,
  forward + backward + optimiser step.

    python scripts/bench_reference_sizes.py [--eval-steps 800] [--train-iters 10] [--no-cpu]

One JSON line per regime.  Under ``rocprofv3 --kernel-trace`` the trace gives the GPU-busy fraction of
each timed region (``scripts/gpu_busy.py``).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def eval_regime(args, dev):
    import bench
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import synthetic

    N, M, T, d = 32, 300, args.eval_steps, 3
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    synthetic.stabilise_dynamics(f)
    traj = bench.to_device(synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=5), dev)
    cal = traj["states"][0][:, None, :] + 0.3 * torch.randn((N, 256, d), device=dev)
    synthetic.calibrate_measurement_heads(f, {k: traj[k][0] for k in ("image", "gripper_pos", "gripper_sensors")}, cal)
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=6)
    noise = (eps0.to(dev), torch.stack(eps).to(dev), torch.stack(us).to(dev))
    f.reserve(steps=T, batch=N, particles=M)
    times = []
    for it in range(args.eval_repeats + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_pf(f, traj, noise, M)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    enc_ms = bench.belief_independent_ms(f, traj)
    out = {"regime": "eval", "filter": "DoorCrossmodalParticleFilter", "batch": N, "particles": M, "steps": T,
           "ms_per_step": 1e3 * best / T, "particle_steps_per_s": N * M * T / best,
           "ms_per_forward_loop": 1e3 * best, "encoders_ms_per_step": enc_ms / T, "recursion_ms_per_step": (1e3 * best - enc_ms) / T,
           "persistent": os.environ.get("MMF_PF_PERSISTENT", "1")}
    if not args.no_cpu:
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        Tc = 24
        sd = {k: v.detach().cpu() for k, v in f.state_dict().items()}
        tr = synthetic.make_trajectories(state_dim=d, T=Tc, N=N, seed=5)
        e0, e, u = synthetic.draw_filter_noise(T=Tc, N=N, M=M, state_dim=d, seed=6)
        _, dt, _ = bench.oracle_pf_run("DoorCrossmodalParticleFilter", sd, tr, e0, e, u, M, warm=2, keep_beliefs=False)
        out["cpu_oracle_ms_per_step"] = 1e3 * dt / (Tc - 2)
        out["cpu_oracle_threads"] = torch.get_num_threads()
    return out


def eval_ekf_regime(args, dev, cls):
    """The extended Kalman filters at the reference's evaluation size (BASELINE config 1: 32 trajectories; sequences of
    hundreds of steps, eval_helpers.py:139-142): whole ``forward_loop`` per step, the recursion as ONE persistent launch
    (csrc/ekf_persistent.inc) against two launches per step."""
    import bench
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic

    N, T, d = 32, args.eval_steps, 3
    torch.manual_seed(0)
    f = mmf.model_types("door")[cls]().to(dev).eval()
    traj = bench.to_device(synthetic.make_trajectories(state_dim=d, T=T + 1, N=N, seed=5), dev)
    obs = {k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
    cov = (torch.eye(d, device=dev) * 0.1)[None].expand(N, d, d)
    out = {"regime": "eval_ekf", "filter": cls, "batch": N, "steps": T}
    old = engine.EKF_PERSISTENT
    try:
        for persistent in (True, False):
            engine.EKF_PERSISTENT = persistent
            times = []
            for it in range(args.eval_repeats + 1):
                f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                est = f.forward_loop(observations=obs, controls=traj["controls"][1:])
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
            key = "persistent" if persistent else "two_launches_per_step"
            out[key] = {"ms_per_step": 1e3 * min(times[1:]) / T, "trajectory_steps_per_s": N * T / min(times[1:])}
            out.setdefault("estimates", est.clone())
            out["same_bits"] = bool(torch.equal(out["estimates"], est))
    finally:
        engine.EKF_PERSISTENT = old
    del out["estimates"]
    return out


def train_regime(args, dev, backend):
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic, train

    N, M, L, d = 32, 30, 16, 3
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).train()
    assert f.num_particles == M
    batch = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=11).items()}
    cov = torch.eye(d, device=dev) * 0.1
    engine.set_training_backend(backend)
    opt = torch.optim.Adam(f.parameters(), lr=1e-4)
    f.noise = mmf.NoiseSource(seed=5)
    times, losses = [], []
    for it in range(args.train_iters + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses.append(train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=f.noise))
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times = sorted(times[2:])
    out = {"regime": "train_e2e", "backend": backend, "filter": "DoorCrossmodalParticleFilter", "batch": N,
           "particles": M, "subsequence_length": L, "ms_per_optimiser_step": 1e3 * times[len(times) // 2],
           "ms_best": 1e3 * times[0], "loss_first_last": [losses[0], losses[-1]]}
    if backend == "hip":  # the same step as ONE hipGraph replay per batch (train.GraphedFilterStep: same kernels, same bits)
        torch.manual_seed(0)
        fg = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).train()
        fg.noise = mmf.NoiseSource(seed=5)
        step = train.GraphedFilterStep(fg, torch.optim.Adam(fg.parameters(), lr=1e-4, capturable=True, fused=True), initial_covariance=cov,
                                       noise=fg.noise, eager_steps=2)
        gt, gl = [], []
        for it in range(args.train_iters + 4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gl.append(step(batch))
            torch.cuda.synchronize()
            gt.append(time.perf_counter() - t0)
        gt = sorted(gt[4:])
        out["hipgraph_replay_ms_per_optimiser_step"] = 1e3 * gt[len(gt) // 2]
        out["hipgraph_replay_ms_best"] = 1e3 * gt[0]
        out["hipgraph_note"] = ("train.GraphedFilterStep, Adam(capturable=True); bit-identical to eager steps with the same optimiser: "
                                "tests/test_gpu_training.py::test_graphed_training_step_replays_the_eager_step")
    engine.set_training_backend(None)
    return out


def train_cpu_twin(args):
    """The oracle (torch, CPU, autograd) on the same training step: the reference's own formulation."""
    from multimodalfilter_amd import synthetic
    from oracle import models as om

    N, M, L, d = 32, 30, 16, 3
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    o = om.build("DoorCrossmodalParticleFilter")
    o.load_state_dict(om.seeded_state_dict(o, seed=8, gain=1.0))
    o.train()
    assert o.num_particles == M
    batch = synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=11)
    obs = {k: batch[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")}
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    opt = torch.optim.Adam(o.parameters(), lr=1e-4)
    times = []
    for it in range(4):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        o.initialize_beliefs(mean=batch["states"][0], covariance=cov)
        pred = o.forward_loop(observations=obs, controls=batch["controls"][1:])
        loss = torch.mean((pred - batch["states"][1:]) ** 2)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    return {"regime": "train_e2e", "backend": "cpu oracle (torch autograd)", "threads": torch.get_num_threads(),
            "batch": N, "particles": M, "subsequence_length": L, "ms_per_optimiser_step": 1e3 * min(times[1:])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eval-steps", type=int, default=800)
    ap.add_argument("--eval-repeats", type=int, default=3)
    ap.add_argument("--train-iters", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", default="eval,ekf,train")
    ap.add_argument("--backends", default="hip,autograd")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if "eval" in args.only:
        print(json.dumps(eval_regime(args, dev)), flush=True)
    if "ekf" in args.only:
        for cls in ("DoorCrossmodalKalmanFilter", "DoorUnimodalKalmanFilter", "DoorKalmanFilter"):
            print(json.dumps(eval_ekf_regime(args, dev, cls)), flush=True)
    if "train" in args.only:
        for backend in args.backends.split(","):
            print(json.dumps(train_regime(args, dev, backend)), flush=True)
        if not args.no_cpu:
            print(json.dumps(train_cpu_twin(args)), flush=True)


if __name__ == "__main__":
    main()
