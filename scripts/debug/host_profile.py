"""Host-side cost of one 20-step forward_loop pass (cProfile of `bench.run_pf`), debug only.

    python scripts/debug/host_profile.py [steps]
"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    from multimodalfilter_amd import synthetic

    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    torch.set_num_threads(16)
    wl = dict(bench.WORKLOADS["door_pf"])
    B, M, d = wl["batch"], wl["particles"], 3
    f = bench.build_filter(wl, dev)
    synthetic.stabilise_dynamics(f)
    _, traj = bench.make_inputs(wl, K, B, 1, dev, d)
    nz = synthetic.draw_filter_noise(T=K, N=B, M=M, state_dim=d, seed=78)
    nz = (nz[0].to(dev), torch.stack(nz[1]).to(dev), torch.stack(nz[2]).to(dev))
    f.num_particles = M
    f.reserve(steps=K, batch=B, particles=M)
    for _ in range(5):
        bench.run_pf(f, traj, nz, M)
    torch.cuda.synchronize()
    # enqueue time alone (no synchronisation inside): what the host spends before the GPU could be done
    ts = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bench.run_pf(f, traj, nz, M)
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
    print("host enqueue ms per pass:", [round(1e3 * t, 3) for t in ts])
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        bench.run_pf(f, traj, nz, M)
        torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
