"""The N>1 path on CPU: two ``gloo`` ranks shard the trajectory axis, each filters its own
shard, and the per-sequence squared errors are all-gathered (the only collective of the
path, ``multimodalfilter_amd/distributed.py``).  The HIP engine cannot run here, so the
CPU oracle stands in as the per-rank filter: what is under test is the sharding, the ragged
all-gather and the shard-invariance of the result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodalfilter_amd import distributed, evaluation, synthetic


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _filter_oracle(traj, noise, M, d):
    from oracle import models as om
    from oracle.tf.base import ReplayNoise

    torch.set_num_threads(1)
    f = om.build("PushUnimodalParticleFilter")
    f.load_state_dict(om.seeded_state_dict(f, seed=5, gain=1.0))
    f.eval()
    f.num_particles = M
    eps0, eps, us = noise
    f.noise = ReplayNoise([eps0] + eps, us)
    N = traj["states"].shape[1]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    with torch.no_grad():
        f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
        return f.forward_loop(observations={k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")},
                              controls=traj["controls"][1:])


def _slice_noise(noise, lo, hi):
    eps0, eps, us = noise
    return eps0[lo:hi], [e[lo:hi] for e in eps], [u[lo:hi] for u in us]


def _worker(rank, world, port, N, T, M, d, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = distributed.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=9)
    noise = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=10)
    lo, hi = distributed.shard_bounds(N, rank, world)
    mine = distributed.shard_trajectories(traj, rank, world)
    assert mine["states"].shape[1] == hi - lo
    pred = _filter_oracle(mine, _slice_noise(noise, lo, hi), M, d)
    mse = evaluation.per_trajectory_mse(pred, mine["states"][1:], start=1)
    distributed.barrier()
    gathered = distributed.all_gather_rows(mse)
    if N % world == 0:  # equal shards and a caller that knows the total: ONE collective, same rows
        calls = []
        real_sizes = dist.all_gather
        dist.all_gather = lambda *a, **k: (calls.append(1), real_sizes(*a, **k))[1]
        try:
            assert torch.equal(distributed.all_gather_rows(mse, N), gathered) and not calls
        finally:
            dist.all_gather = real_sizes
    slowest = distributed.max_over_ranks(float(rank + 1), torch.device("cpu"))
    assert slowest == float(world)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), gathered.numpy())
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("N", [5, 4])  # 5: ragged (rank 0 owns 3 trajectories, rank 1 owns 2); 4: equal shards = one collective
def test_two_rank_sharding_matches_single_process(tmp_path, N):
    T, M, d = 3, 16, 2
    port = _free_port()
    mp.spawn(_worker, args=(2, port, N, T, M, d, str(tmp_path)), nprocs=2, join=True)
    gathered = np.load(tmp_path / "gathered.npy")
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=9)
    noise = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=10)
    pred = _filter_oracle(traj, noise, M, d)
    want = evaluation.per_trajectory_mse(pred, traj["states"][1:], start=1).numpy()
    assert gathered.shape == (N, d)
    np.testing.assert_allclose(gathered, want, rtol=1e-5, atol=1e-7)  # PF path is shard-invariant


def test_shard_bounds_cover_the_batch():
    for n in (1, 7, 256, 8192):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_eval_arithmetic_matches_reference_vectors(golden_dir):
    """H1 on the product side: same numbers as the reference's run_eval (tests/golden/eval.npz)."""
    z = np.load(os.path.join(golden_dir, "eval.npz"))
    for tname in ("door", "push"):
        pred = torch.from_numpy(z[f"{tname}/pred"])
        true = torch.from_numpy(z[f"{tname}/true"][1:])
        rmse = evaluation.raw_rmse(evaluation.per_trajectory_mse(pred, true))
        res = evaluation.task_metrics(tname, rmse)
        for k, v in res.items():
            np.testing.assert_allclose(np.asarray(v), z[f"{tname}/{k}"], rtol=1e-6)


def test_synthetic_generator_is_seeded_and_shaped():
    a = synthetic.make_trajectories(state_dim=3, T=4, N=3, seed=1, image_blackout_ratio=0.4)
    b = synthetic.make_trajectories(state_dim=3, T=4, N=3, seed=1, image_blackout_ratio=0.4)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert a["states"].shape == (5, 3, 3) and a["image"].shape == (5, 3, 32, 32)
    assert a["controls"].shape == (5, 3, 7) and set(a["controls"][..., 6].unique().tolist()) <= {-1.0, 1.0}
    assert float(a["image"].abs().max()) <= 1.0
    dark = (a["image"].abs().sum((-1, -2)) == 0)
    assert 0 < int(dark.sum()) < dark.numel()


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 2))
    x = torch.full((3, 5), float(rank + 1))
    net(x).pow(2).sum().backward()
    local = [p.grad.clone() for p in net.parameters()]
    n = distributed.all_reduce_gradients(net)
    assert n == sum(p.numel() for p in net.parameters())
    # round 6: no copy back -- every p.grad is a view of the ONE reduced buffer, in parameter order
    base = next(net.parameters()).grad.untyped_storage().data_ptr()
    off = 0
    for p in net.parameters():
        assert p.grad.untyped_storage().data_ptr() == base and p.grad.storage_offset() == off and p.grad.shape == p.shape
        off += p.numel()
    torch.save({"local": local, "reduced": [p.grad.clone() for p in net.parameters()]},
               os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_gradient_all_reduce_averages_over_ranks(tmp_path):
    """X2: flat all-reduce of the parameter gradients (data-parallel training step)."""
    port = _free_port()
    mp.spawn(_grad_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    for a, b, r0, r1 in zip(g0["local"], g1["local"], g0["reduced"], g1["reduced"]):
        torch.testing.assert_close(r0, (a + b) / 2)
        torch.testing.assert_close(r1, r0)


def _run_bench(extra_env, *argv):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=env,
                          capture_output=True, text=True, timeout=240)


@pytest.mark.timeout(300)
def test_bench_self_launches_n_ranks():
    """``python bench.py --gpus 2`` as the driver invokes it for N > 1 when no launcher wraps it:
    the parent spawns 2 ranks (gloo here), rank 0 prints ONE JSON line carrying ``n_gpus: 2``
    and the world size the process group reported."""
    import json

    r = _run_bench({"MMF_BENCH_DRY": "1", "MMF_DIST_BACKEND": "gloo"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["world_size_seen"] == 2
    assert out["gathered_rows"] == 3 and out["max_over_ranks"] == 1.0  # ragged all-gather: 1 + 2 rows
    assert out["steps"] == 3 and out["warmup"] == 1


@pytest.mark.timeout(300)
def test_bench_launcher_propagates_a_failing_rank():
    r = _run_bench({"MMF_BENCH_DRY": "fail1", "MMF_DIST_BACKEND": "gloo"}, "--gpus", "2")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.timeout(600)
def test_bench_eight_rank_preflight_strong_scaling_shards():
    """The command the driver will run on an 8-GPU node for BASELINE config 4, rehearsed with 8 gloo ranks and no GPU
    work (``MMF_BENCH_DRY``): ``launch_ranks`` spawns 8 children, they rendezvous, and the 8192 trajectories are cut
    into 8 contiguous shards of 1024; the weak-scaling default reports 8 x 256.  (No scaling curve has been
    measured on hardware: the pool gives this builder one GPU -- DESIGN.md section 6.)"""
    import json

    r = _run_bench({"MMF_BENCH_DRY": "1", "MMF_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}, "--gpus", "8", "--steps", "4",
                   "--warmup", "1", "--workload", "door_ekf", "--global-batch", "8192")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["world_size_seen"] == 8 and out["scaling"] == "strong"
    assert out["gathered_rows"] == sum(range(1, 9)) and out["max_over_ranks"] == 7.0
    shards = sorted(out["shards"])
    assert [s[0] for s in shards] == list(range(8))
    assert shards[0][1] == 0 and shards[-1][2] == 8192
    assert all(a[2] == b[1] for a, b in zip(shards, shards[1:])) and all(s[2] - s[1] == 1024 for s in shards)
    r = _run_bench({"MMF_BENCH_DRY": "1", "MMF_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}, "--gpus", "8")
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and out["global_batch"] == 8 * 256 and out["workload"] == "door_pf"


class _OracleCounterNoise:
    """The oracle-side twin of ``multimodalfilter_amd.utils.CounterNoise``: every draw is a pure function of
    (seed, step, GLOBAL trajectory index, particle), restated on the CPU by ``oracle.strict`` (bit for bit the
    generator of ``include/mmf_philox.h``)."""

    def __init__(self, seed, traj_offset=0):
        self.seed, self.traj_offset, self.step_g, self.step_u = seed, traj_offset, 0, 0

    def gaussian(self, shape, *, like=None):
        from oracle import strict

        N, M, d = shape
        out = torch.from_numpy(strict.philox_normals(self.seed, self.step_g, N, M, d, traj0=self.traj_offset))
        self.step_g += 1
        return out

    def uniform(self, shape, *, like=None):
        from oracle import strict

        out = torch.from_numpy(strict.philox_uniforms(self.seed, self.step_u, 1, shape[0], traj0=self.traj_offset))[0]
        self.step_u += 1
        return out


def test_counter_noise_makes_an_eight_way_sharded_filter_equal_the_single_process_run():
    """Weak / strong scaling shards the trajectory axis; with counter-based noise keyed by the GLOBAL trajectory
    index (``CounterNoise(traj_offset=...)``) shard r of an 8-way run draws exactly what trajectories
    ``[lo, hi)`` of the single-process run draw -- the draws themselves are compared bit for bit -- so the sharded
    particle filter reproduces the unsharded one (CPU oracle as the per-rank filter: its torch GEMMs round
    differently for different batch sizes, hence 1e-5 on the estimates; on the HIP engine, whose rows do not
    depend on their batch position, ``tests/test_gpu_strict.py::test_counter_noise_*`` has a shard equal its
    slice of the batch in every bit)."""
    from oracle import models as om
    from oracle import strict

    if not strict.available():
        pytest.skip("oracle/strict needs gcc")
    torch.set_num_threads(2)
    N, T, M, d, world = 11, 3, 24, 2, 8   # ragged: shards of 2 and 1 trajectories
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=19)

    def run(tr, offset):
        f = om.build("PushUnimodalParticleFilter")
        f.load_state_dict(om.seeded_state_dict(f, seed=5, gain=1.0))
        f.eval()
        f.num_particles = M
        f.noise = _OracleCounterNoise(4711, traj_offset=offset)
        n = tr["states"].shape[1]
        cov = (torch.eye(d) * 0.1)[None].expand(n, d, d)
        with torch.no_grad():
            f.initialize_beliefs(mean=tr["states"][0], covariance=cov)
            return f.forward_loop(observations={k: tr[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")},
                                  controls=tr["controls"][1:])

    whole = run(traj, 0)
    parts = []
    for r in range(world):
        lo, hi = distributed.shard_bounds(N, r, world)
        parts.append(run(distributed.shard_trajectories(traj, r, world), lo))
        # the shard's draws ARE the global run's draws for its trajectories: Gaussians of every step, resampling uniforms
        for step in range(T + 1):
            a = strict.philox_normals(4711, step, hi - lo, M, d, traj0=lo)
            b = strict.philox_normals(4711, step, N, M, d, traj0=0)[lo:hi]
            assert np.array_equal(a, b)
        assert np.array_equal(strict.philox_uniforms(4711, 0, T, hi - lo, traj0=lo), strict.philox_uniforms(4711, 0, T, N, traj0=0)[:, lo:hi])
    torch.testing.assert_close(torch.cat(parts, dim=1), whole, rtol=0, atol=1e-5)


# ---- the data-parallel TRAINING step (BASELINE config 5; SURVEY.md 8e "Collective (training, C5)"): replicated weights,
# rank-private subsequence batches, one flat all-reduce of the gradients, optimiser step.  The HIP engine cannot run
# here; the CPU oracle's train-mode particle filter (torch autograd) stands in as the per-rank model.
def _train_step(f, traj, eps0, eps, d, lr, all_reduce):
    from oracle.tf.base import ReplayNoise

    N = traj["states"].shape[1]
    f.noise = ReplayNoise([eps0] + list(eps), [])
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    opt = torch.optim.SGD(f.parameters(), lr=lr)
    opt.zero_grad(set_to_none=True)
    f.initialize_beliefs(mean=traj["states"][0], covariance=cov)
    pred = f.forward_loop(observations={k: traj[k][1:] for k in ("image", "gripper_pos", "gripper_sensors")},
                          controls=traj["controls"][1:])
    loss = torch.mean((pred - traj["states"][1:]) ** 2)
    loss.backward()
    if all_reduce:
        distributed.all_reduce_gradients(f)
    opt.step()
    return float(loss.detach())


def _train_model(M):
    from oracle import models as om

    torch.set_num_threads(1)
    f = om.build("PushUnimodalParticleFilter")
    f.load_state_dict(om.seeded_state_dict(f, seed=5, gain=1.0))
    f.train()
    f.num_particles = M
    return f


def _train_worker(rank, world, port, N, T, M, d, lr, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    distributed.init_from_env(backend="gloo")
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=21)
    g = torch.Generator().manual_seed(22)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    lo, hi = distributed.shard_bounds(N, rank, world)
    f = _train_model(M)
    loss = _train_step(f, distributed.shard_trajectories(traj, rank, world), eps0[lo:hi], [e[lo:hi] for e in eps], d, lr, True)
    torch.save({"loss": loss, "params": {k: v.detach().clone() for k, v in f.named_parameters()}}, os.path.join(out_dir, f"t{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_data_parallel_training_step_equals_the_single_process_step(tmp_path, world):
    """After ONE data-parallel optimiser step (``train.train_filter_step(.., all_reduce=True)``'s sequence: backward,
    ``distributed.all_reduce_gradients``, step) every rank holds the same weights bit for bit, and they equal the
    weights of a single process stepping on the concatenated batch (equal shards: the mean of the shard losses is the
    batch loss) to fp32 summation order."""
    N, T, M, d, lr = 8, 2, 6, 2, 0.05
    port = _free_port()
    mp.spawn(_train_worker, args=(world, port, N, T, M, d, lr, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(tmp_path / f"t{r}.pt") for r in range(world)]
    for r in range(1, world):
        for k, v in got[0]["params"].items():
            assert torch.equal(v, got[r]["params"][k]), (r, k)
    traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=21)
    g = torch.Generator().manual_seed(22)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    f = _train_model(M)
    before = {k: v.detach().clone() for k, v in f.named_parameters()}
    loss = _train_step(f, traj, eps0, eps, d, lr, False)
    assert abs(sum(x["loss"] for x in got) / world - loss) < 1e-5 * max(1.0, abs(loss))
    moved = 0
    for k, v in f.named_parameters():
        step = (v.detach() - before[k]).abs().max()
        moved += int(step > 0)
        torch.testing.assert_close(got[0]["params"][k], v.detach(), rtol=1e-4, atol=1e-6 + 1e-3 * float(step))
    assert moved > 20  # the step changed the networks (a no-op step would pass the comparison trivially)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 8])
def test_bench_training_leg_preflight(world):
    """``bench.py --workload push_train --gpus N`` rehearsed with N gloo ranks and no GPU work (``MMF_BENCH_DRY``): the
    launcher spawns the ranks before anything touches HIP, they rendezvous, and the training leg's one collective -- a
    flat all-reduce of the filter's 696,993 gradient elements -- leaves the exact average on every rank."""
    import json

    r = _run_bench({"MMF_BENCH_DRY": "1", "MMF_DIST_BACKEND": "gloo", "OMP_NUM_THREADS": "1"}, "--gpus", str(world),
                   "--workload", "push_train", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["workload"] == "push_train" and out["n_gpus"] == world and out["world_size_seen"] == world
    assert out["allreduce_elements"] == 696993 and out["allreduce_average_exact_on_every_rank"] is True
    assert out["allreduce_ms"] > 0 and out["global_batch"] == 32 * world
