"""Does train.GraphedFilterStep capture and replay the reference-sized training step, with the eager step's results?"""
import copy, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multimodalfilter_amd as mmf
from multimodalfilter_amd import engine, synthetic, train

dev = torch.device("cuda:0")
N, M, L, d = 32, 30, 16, 3
engine.set_training_backend("hip")
torch.manual_seed(0)
f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).train()
f.num_particles = M
g = copy.deepcopy(f)
batches = [{k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=3 + i).items()} for i in range(4)]
cov = torch.eye(d, device=dev) * 0.1
for opt_name in ("sgd", "adam"):
    mk = (lambda m: torch.optim.SGD(m.parameters(), lr=1e-4)) if opt_name == "sgd" else (lambda m: torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True))
    f2, g2 = copy.deepcopy(f), copy.deepcopy(g)
    of, og = mk(f2), mk(g2)
    f2.noise, g2.noise = mmf.NoiseSource(seed=5), mmf.NoiseSource(seed=5)
    step = train.GraphedFilterStep(g2, og, initial_covariance=cov, noise=g2.noise, eager_steps=2)
    le, lg = [], []
    for i in range(10):
        le.append(train.train_filter_step(f2, batches[i % 4], of, initial_covariance=cov, noise=f2.noise))
        lg.append(step(batches[i % 4]))
    print(opt_name, "eager  ", [round(x, 7) for x in le])
    print(opt_name, "graphed", [round(x, 7) for x in lg])
    worst = max(float((p - q).abs().max()) for p, q in zip(f2.parameters(), g2.parameters()))
    print(opt_name, "max |weight difference| after 10 steps:", worst)
    for name, fn in (("eager", lambda b: train.train_filter_step(f2, b, of, initial_covariance=cov, noise=f2.noise)), ("graphed", step)):
        torch.cuda.synchronize()
        ts = []
        for i in range(12):
            t0 = time.perf_counter(); fn(batches[i % 4]); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print(opt_name, name, "ms per optimiser step: median %.3f best %.3f" % (1e3 * ts[len(ts) // 2], 1e3 * ts[0]))
