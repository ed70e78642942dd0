"""Which configuration makes the native training recursion differ from the stepwise K6 path?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import multimodalfilter_amd as mmf
from multimodalfilter_amd import engine
from oracle import models as om

dev = torch.device("cuda:0")


def run(tname, cls, N, M, T, chunk_rows, seed=31):
    task = om.TASKS[tname]
    d = task.state_dim
    g = torch.Generator().manual_seed(seed)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1),
           "gripper_pos": torch.randn((T, N, 3), generator=g), "gripper_sensors": torch.randn((T, N, 7), generator=g)}
    ctrl = torch.randn((T, N, 7), generator=g); x0 = torch.randn((N, d), generator=g); target = torch.randn((T, N, d), generator=g)
    eps0 = torch.randn((N, M, d), generator=g); eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    torch.manual_seed(3)
    f = mmf.model_types(tname)[cls]().to(dev).train()
    f.num_particles = M
    engine.set_training_backend("hip"); engine.set_default_precision("f32")
    engine.TRAIN_CHUNK_ROWS = chunk_rows
    res = {}
    for native in (False, True):
        f.use_native_loop = native
        f.zero_grad(set_to_none=True)
        f.noise = mmf.ReplayNoise([eps0] + eps, [])
        f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
        pred = f.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
        loss = torch.mean((pred - target.to(dev)) ** 2)
        loss.backward()
        res[native] = {n: p.grad.detach().clone() for n, p in f.named_parameters() if p.grad is not None}
    top = max(float(v.abs().max()) for v in res[False].values())
    diffs = sorted(((float((res[False][k] - res[True][k]).abs().max()) / max(1e-3 * top, float(res[False][k].abs().max())), k) for k in res[False]), reverse=True)
    print(tname, cls, N, M, T, chunk_rows, "->", [(round(a, 6), b) for a, b in diffs[:3]])


for cfg in [("door", "DoorParticleFilter", 3, 100, 3, 100), ("door", "DoorParticleFilter", 3, 100, 3, 32768),
            ("door", "DoorParticleFilter", 3, 128, 3, 32768), ("door", "DoorParticleFilter", 3, 100, 1, 32768),
            ("door", "DoorCrossmodalParticleFilter", 3, 100, 3, 100), ("door", "DoorUnimodalParticleFilter", 3, 100, 3, 100),
            ("push", "PushParticleFilter", 3, 100, 3, 32768), ("door", "DoorParticleFilter", 3, 100, 3, 32768, 7)]:
    run(*cfg)
