"""Does the N=32, M=300 random-weight filter really leave the f16x3 operand range, and when?"""
import torch
import multimodalfilter_amd as mmf
from multimodalfilter_amd import engine, evaluation, synthetic

dev = torch.device("cuda:0")
N, M, d = 32, 300, 3
torch.manual_seed(0)
f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
f.num_particles = M
for K in (50, 100, 200, 400):
    traj = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=K, N=N, seed=20201025 + 2).items()}
    cal = traj["states"][0][:, None, :] + 0.3 * torch.randn((N, 256, d), device=dev)
    if K == 50:
        synthetic.calibrate_measurement_heads(f, {k: traj[k][0] for k in ("image", "gripper_pos", "gripper_sensors")}, cal)
    eps0, eps, us = synthetic.draw_filter_noise(T=K, N=N, M=M, state_dim=d, seed=78)
    for prec in ("f32", "f16x3"):
        engine.set_default_precision(prec)
        f.noise = mmf.StackedNoise(eps0.to(dev), torch.stack(eps).to(dev), torch.stack(us).to(dev))
        try:
            pred = evaluation.run_filter(f, traj)
            print(K, prec, "ok  max|estimate|", float(pred.abs().max()), "max|particle|", float(f.particle_states.abs().max()))
        except Exception as e:
            engine.range_flag(dev).zero_()
            print(K, prec, "RAISED", str(e)[:60], "max|particle|", float(f.particle_states.abs().max()))
