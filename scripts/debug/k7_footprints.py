import sys, torch
sys.path.insert(0, '.')
import multimodalfilter_amd as mmf
from multimodalfilter_amd import trajprog, synthetic
orig=trajprog.TrajProgram.run
seen={}
def run(self, tensors, R):
    key=id(self)
    if key not in seen:
        seen[key]=(len(self._instrs), self._footprint(), R, sorted(tensors))
    return orig(self, tensors, R)
trajprog.TrajProgram.run=run
dev=torch.device("cuda:0")
for name in ("DoorCrossmodalKalmanFilter","DoorCrossmodalParticleFilter"):
    f=mmf.model_types("door")[name]().to(dev).eval()
    traj={k:v.to(dev) for k,v in synthetic.make_trajectories(state_dim=3,T=4,N=64,seed=1).items()}
    if "Particle" in name: f.num_particles=256
    f.initialize_beliefs(mean=traj["states"][0], covariance=(0.1*torch.eye(3,device=dev))[None].expand(64,3,3))
    f.forward_loop(observations={k:traj[k][1:] for k in ("image","gripper_pos","gripper_sensors")}, controls=traj["controls"][1:])
    print(name)
    for v in seen.values(): print("  ", v)
    seen.clear()
