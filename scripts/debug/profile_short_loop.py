"""Host-side profile of one 20-step forward_loop of the headline workload (what the driver's
`bench.py --steps 20 --warmup 5` times): where the per-loop fixed cost goes."""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import multimodalfilter_amd as mmf
from multimodalfilter_amd import synthetic, evaluation

dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["door_pf"])
K, B, M, d = 20, wl["batch"], wl["particles"], 3
f = bench.build_filter(wl, dev)
synthetic.stabilise_dynamics(f)
_, traj = bench.make_inputs(wl, K, B, 1, dev, d)
f.num_particles = M
nz = synthetic.draw_filter_noise(T=K, N=B, M=M, state_dim=d, seed=78)
nz = (nz[0].to(dev), torch.stack(nz[1]).to(dev), torch.stack(nz[2]).to(dev))
f.reserve(steps=K, batch=B, particles=M)
for _ in range(3):
    bench.run_pf(f, traj, nz, M)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); bench.run_pf(f, traj, nz, M); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.2f} ms, until GPU done {1e3*(t2-t0):.2f} ms  ({1e3*(t2-t0)/K:.3f} ms/step)")
pr = cProfile.Profile(); pr.enable(); bench.run_pf(f, traj, nz, M); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
