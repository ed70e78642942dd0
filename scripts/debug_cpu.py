import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalfilter_amd import synthetic
from oracle import models as om
from oracle.tf.base import NoiseSource
print("cpu_count", os.cpu_count()); os.system("lscpu | grep -E 'Model name|Socket|Core|Thread' | head -5")
d = 3; M = 4096; N = 32
traj = synthetic.make_trajectories(state_dim=d, T=3, N=N, seed=1)
obs = synthetic.observations_of(traj)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    o = om.build("DoorCrossmodalParticleFilter"); o.eval(); o.num_particles = M; o.noise = NoiseSource(0)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    with torch.no_grad():
        o.initialize_beliefs(mean=traj["states"][0], covariance=cov)
        o(observations={k: v[1] for k, v in obs.items()}, controls=traj["controls"][1])
        t0 = time.perf_counter()
        for t in (2, 3):
            o(observations={k: v[t] for k, v in obs.items()}, controls=traj["controls"][t])
        dt = (time.perf_counter() - t0) / 2
    print(f"threads {th}: {dt*1e3:.1f} ms/step -> {N*M/dt:.3e} particle-steps/s", flush=True)
