"""First timed pass after a W=4 warm-up vs later passes (same process), K=32."""
import time
import torch
import multimodalfilter_amd as mmf
from multimodalfilter_amd import evaluation, synthetic, engine

dev = torch.device("cuda:0")
N, M, d, K, W = 256, 4096, 3, 32, 4
f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
f.num_particles = M
def inputs(T, seed):
    traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=seed).items()}
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=seed + 1)
    return traj, (eps0.to(dev), torch.stack(eps).to(dev), torch.stack(us).to(dev))
tw, nw = inputs(W, 1)
tk, nk = inputs(K, 3)
f.reserve(steps=K, batch=N, particles=M)
def whole(traj, nz):
    f.noise = mmf.StackedNoise(*nz)
    pred = evaluation.run_filter(f, traj)
    return evaluation.per_trajectory_mse(pred, traj["states"][1:], start=min(30, traj["states"].shape[0] // 2))
import sys
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    whole(tw, nw)
torch.cuda.synchronize()
for i in range(4):
    st0 = torch.cuda.memory_stats()
    n0 = st0["num_device_alloc"]
    torch.cuda.synchronize(); t0 = time.perf_counter(); whole(tk, nk); torch.cuda.synchronize()
    st1 = torch.cuda.memory_stats()
    for k in ("segment.small_pool.allocated", "segment.large_pool.allocated", "reserved_bytes.small_pool.current",
              "reserved_bytes.large_pool.current", "allocated_bytes.all.peak"):
        if st1[k] != st0[k]:
            print("   ", k, st0[k], "->", st1[k])
    print(f"pass {i}: {(time.perf_counter() - t0) * 1e3:.3f} ms, hipMalloc calls {torch.cuda.memory_stats()['num_device_alloc'] - n0}")
