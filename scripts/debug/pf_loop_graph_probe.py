#!/usr/bin/env python3
"""Would a hipGraph of the particle-filter step loop (4 launches per step) shorten the gaps between its dependent launches?
The recursion call of one forward_loop (mmf_pf_forward_loop on fixed buffers) timed as plain launches and as a graph replay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import multimodalfilter_amd as mmf  # noqa: E402
from multimodalfilter_amd import _abi, engine, synthetic  # noqa: E402

dev = torch.device("cuda:0")
engine.PF_PERSISTENT = False
for N, M, T in ((256, 4096, 64), (256, 1024, 64), (32, 4096, 64)):
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).eval()
    synthetic.stabilise_dynamics(f)
    traj = bench.to_device(synthetic.make_trajectories(state_dim=3, T=T, N=N, seed=5), dev)
    eps0, eps, us = synthetic.draw_filter_noise(T=T, N=N, M=M, state_dim=3, seed=6)
    noise = (eps0.to(dev), torch.stack(eps).to(dev), torch.stack(us).to(dev))
    result = {}
    real = _abi.pf_forward_loop

    def probe(a, like, *r, **k):  # measured INSIDE the call: the operands of `a` are alive only here
        out = real(a, like, *r, **k)
        if a.T < 32 or a.events:
            return out
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            def plain(reps=5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    real(a, like, None, 1)
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps
            plain(2)
            tp = min(plain() for _ in range(3))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                real(a, like, None, 1)

            def replay(reps=5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    g.replay()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps
            replay(2)
            tg = min(replay() for _ in range(3))
        torch.cuda.current_stream().wait_stream(s)
        result["line"] = f"{a.N} x {a.M}, {a.T} steps: launches {1e6 * tp / a.T:.2f} us/step, graph replay {1e6 * tg / a.T:.2f} us/step ({100 * (tp - tg) / tp:+.1f} %)"
        return out

    _abi.pf_forward_loop = probe
    try:
        bench.run_pf(f, traj, noise, M)
    finally:
        _abi.pf_forward_loop = real
    print(result.get("line", "no loop of >= 32 steps seen"), flush=True)
