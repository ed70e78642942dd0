import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import test_gpu_known_answers as T
import multimodalfilter_amd as mmf
from oracle import tf as otf
from oracle.tf.base import ReplayNoise as OReplay

mode = "systematic"
dev = torch.device("cuda:0")
d = 3
A, B, L, Rt = T._system(d)
N, Tn, M = 3, 4, 16384
g = torch.Generator().manual_seed(2)
us = torch.randn(Tn, N, 7, generator=g)
zs = 0.3 * torch.randn(Tn, N, d, generator=g)
mu0 = 0.2 * torch.randn(N, d, generator=g)
cov0 = (0.1 * torch.eye(d))[None].expand(N, d, d)
eps0 = torch.randn((N, M, d), generator=g)
eps = [torch.randn((N, M, d), generator=g) for _ in range(Tn)]
uu = [torch.rand((N,), generator=g) for _ in range(Tn)]
ODyn, _, OLik = T._user_models(otf.base, A, B, L, Rt, "cpu")
o = otf.filters.ParticleFilter(dynamics_model=ODyn(), measurement_model=OLik(), num_particles=M, resample_mode=mode)
o.eval(); o.noise = OReplay([eps0] + eps, uu)
o.initialize_beliefs(mean=mu0, covariance=cov0)
Dyn, _, Lik = T._user_models(mmf.base, A, B, L, Rt, dev)
f = mmf.filters.ParticleFilter(dynamics_model=Dyn(), measurement_model=Lik(), num_particles=M, resample_mode=mode)
f.eval(); f.record_indices = True
f.noise = mmf.ReplayNoise([eps0] + eps, uu)
f.initialize_beliefs(mean=mu0.to(dev), covariance=cov0.to(dev))
print("init diff", float((f.particle_states.cpu() - o.particle_states).abs().max()))
for t in range(Tn):
    so, wo = o.particle_states, o.particle_log_weights
    se, we = f.particle_states.cpu(), f.particle_log_weights.cpu()
    print(t, "belief diff states", float((se - so).abs().max()), "per traj", (se - so).abs().amax((1, 2)).tolist())
    want = o(observations={"z": zs[t]}, controls=us[t])
    est = f(observations={"z": zs[t].to(dev)}, controls=us[t].to(dev)).cpu()
    idx_e, idx_o = f.last_resample_indices.cpu().long(), o.last_resample_indices
    print(t, "est diff per traj", (est - want).abs().amax(1).tolist(), "idx mismatches per traj", (idx_e != idx_o).sum(1).tolist(),
          "max idx delta", int((idx_e - idx_o).abs().max()))
    # ESS of oracle weights before resampling is not kept; recompute from loglik
