"""Step-by-step vs Python forward_loop vs native C loop: where do the bits diverge?"""
import torch
import multimodalfilter_amd as mmf
from multimodalfilter_amd import filters

dev = torch.device("cuda:0")
d, N, M = 3, 5, 300
for resample, T in [(False, 3), (True, 3)]:
    g = torch.Generator().manual_seed(23)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1).to(dev),
           "gripper_pos": torch.randn((T, N, 3), generator=g).to(dev),
           "gripper_sensors": torch.randn((T, N, 7), generator=g).to(dev)}
    ctrl = torch.randn((T, N, 7), generator=g).to(dev)
    x0 = torch.randn((N, d), generator=g).to(dev)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d).to(dev)
    eps0 = torch.randn((N, M, d), generator=g).to(dev)
    eps = torch.randn((T, N, M, d), generator=g).to(dev)
    us = torch.rand((T, N), generator=g).to(dev)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    f.num_particles = M
    f.resample = resample

    def fresh():
        f.noise = mmf.StackedNoise(eps0, eps, us)
        f.initialize_beliefs(mean=x0, covariance=cov)

    fresh()
    step = torch.stack([f(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t]) for t in range(T)])
    fresh()
    native = f.forward_loop(observations=obs, controls=ctrl)
    fresh()
    keep = filters.ParticleFilter._native_loop
    filters.ParticleFilter._native_loop = lambda self, *a: None
    pyloop = f.forward_loop(observations=obs, controls=ctrl)
    filters.ParticleFilter._native_loop = keep
    print(resample, "step-native", (step - native).abs().amax(dim=(1, 2)).tolist())
    print(resample, "step-pyloop", (step - pyloop).abs().amax(dim=(1, 2)).tolist())
    print(resample, "pyloop-native", (pyloop - native).abs().amax(dim=(1, 2)).tolist())
    # encoders: N rows vs T*N rows
    meas = f.measurement_model
    flat = {k: v.reshape((T * N,) + tuple(v.shape[2:])) for k, v in obs.items()}
    big = meas.encode_observations(flat)
    for t in range(T):
        small = meas.encode_observations({k: v[t] for k, v in obs.items()})
        for k in small:
            diff = (small[k] - big[k][t * N:(t + 1) * N]).abs().max().item()
            if diff:
                print("  encoder ctx", k, "t", t, diff)
