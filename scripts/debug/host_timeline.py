"""Host-side wall-clock of the pieces of one 20-step PF pass (what the driver's flags time)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import multimodalfilter_amd as mmf
from multimodalfilter_amd import synthetic, evaluation, filters

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

marks = []
def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); marks.append((name, 1e3 * (time.perf_counter() - t))); return r
    setattr(obj, name, w)
wrap(f, "initialize_beliefs")
wrap(f.measurement_model, "encode_observations")
wrap(f.dynamics_model, "encode_controls")
wrap(f, "_native_loop")
from multimodalfilter_amd import engine
wrap(engine, "check_range")
filters.check_range = engine.check_range
for rep in range(2):
    marks.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pred = bench.run_pf(f, traj, nz, M)
    t1 = time.perf_counter()
    mse = evaluation.per_trajectory_mse(pred, traj["states"][1:], start=10)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"run_pf {1e3*(t1-t0):.2f} ms, + mse & sync {1e3*(t2-t1):.3f} ms;", "; ".join(f"{n} {v:.3f}" for n, v in marks))
