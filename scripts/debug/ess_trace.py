"""ESS/M of the engine's own weights over a long run, after bench.calibrate_to_band: is the tracking regime stationary?
    python scripts/debug/ess_trace.py [door_pf|push_pf] [M] [steps]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from multimodalfilter_amd import synthetic  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "door_pf"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS[name])
d = 3 if wl["task"] == "door" else 2
f = bench.build_filter(wl, dev)
synthetic.stabilise_dynamics(f)
f.num_particles = M
trace = bench.calibrate_to_band(f, wl, dev, d, M)
print(json.dumps({"calibration": trace}))
for N, seed in ((32, 1), (64, 2)):
    traj = bench.to_device(synthetic.make_trajectories(state_dim=d, T=T, N=N, seed=seed), dev)
    for Mi in (M, M // 4):
        f.num_particles = Mi
        run = bench.FilterRun(f, traj, ("philox", 5 + seed), particles=Mi)
        ess = bench.engine_ess(run, [(0, T)])[0]
        spread = []
        print(json.dumps({"N": N, "M": Mi, "ess_batch_mean_every_8th_step": [round(float(x), 3) for x in ess.mean(1)[::8]],
                          "ess_quartiles_last_step": [round(float(x), 3) for x in torch.quantile(ess[-1], torch.tensor([0.1, 0.5, 0.9]))]}))
