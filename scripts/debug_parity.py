import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multimodalfilter_amd as mmf
from multimodalfilter_amd import synthetic, evaluation
from oracle import models as om
from oracle.tf.base import ReplayNoise as OReplay
import bench

dev = torch.device("cuda:0")
wl = bench.WORKLOADS["door_pf"]; d = 3; M = 1024; N = 8; T = 8
torch.manual_seed(0)
f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
f.num_particles = M
traj = synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=4242)
trd = {k: v.to(dev) for k, v in traj.items()}
cal = trd["states"][0][:, None, :] + 0.3 * torch.randn((N, 256, d), device=dev)
s = synthetic.calibrate_measurement_heads(f, {k: trd[k][0] for k in ("image", "gripper_pos", "gripper_sensors")}, cal)
print("head scale", s)
eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=d, seed=4243)
oracle = om.build("DoorCrossmodalParticleFilter")
oracle.load_state_dict({k: v.detach().cpu() for k, v in f.state_dict().items()})
oracle.eval(); oracle.num_particles = M; oracle.noise = OReplay([eps0] + eps, us)
obs = synthetic.observations_of(traj)
cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
f.noise = mmf.ReplayNoise([eps0] + eps, us); f.record_indices = True
f.initialize_beliefs(mean=trd["states"][0], covariance=cov.to(dev))
with torch.no_grad():
    oracle.initialize_beliefs(mean=traj["states"][0], covariance=cov)
    for t in range(1, T + 1):
        o = {k: v[t] for k, v in obs.items()}
        w = oracle(observations=o, controls=traj["controls"][t])
        g = f(observations={k: v.to(dev) for k, v in o.items()}, controls=trd["controls"][t]).cpu()
        same = (f.last_resample_indices.cpu().long() == oracle.last_resample_indices).float().mean().item()
        lw = oracle.particle_log_weights
        dx = (f.particle_states.cpu() - oracle.particle_states).abs().max().item()
        print(f"t={t} est err {float((g - w).abs().max()):.3e} scale {float(w.abs().max()):.2f} idx same {same:.6f} states maxdiff {dx:.3e}")
