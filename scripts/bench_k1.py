#!/usr/bin/env python3
"""K1 alone (reweight + resample + gather) on resident inputs: us per launch and fraction of the
8 TB/s HBM peak for the SURVEY 8d shapes.

``--batch-sweep`` (round 4, verdict item 6: "de-phase the workgroups"): the same kernel at M = 4096 for N = 64 ..
2048 trajectories.  One workgroup serves one trajectory and a 1024-thread workgroup with ~97 KB of LDS leaves room
for ONE per CU, so up to N = 256 every workgroup has a CU to itself and all of them walk the same phases at the same
time (load, reduce, store): the launch takes one workgroup's latency chain whatever N.  Beyond 256 a CU runs its
workgroups one after the other -- de-phased against the other CUs' -- and the time per TRAJECTORY falls: that is
what de-phasing buys, and it needs more trajectories than CUs."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodalfilter_amd import _abi  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    shapes = ((256, 4096, 3), (256, 1024, 3), (1024, 4096, 2), (32, 4096, 3), (32, 300, 3), (256, 8192, 2))
    if "--batch-sweep" in sys.argv:
        shapes = tuple((n, 4096, 3) for n in (64, 128, 256, 384, 512, 768, 1024, 2048))
    for N, M, d in shapes:
        ll = torch.randn((N, M), device=dev, generator=g) * 1.2
        lw = torch.full((N, M), -float(torch.log(torch.tensor(float(M)))), device=dev)
        x = torch.randn((N, M, d), device=dev, generator=g)
        u = torch.rand((N,), device=dev, generator=g)
        est = torch.empty((N, d), device=dev)
        xo, lo = torch.empty_like(x), torch.empty_like(lw)
        for _ in range(5):
            _abi.pf_reweight_resample(ll, lw, x, u, est, xo, lo, None, 1)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        s.record()
        for _ in range(reps):
            _abi.pf_reweight_resample(ll, lw, x, u, est, xo, lo, None, 1)
        e.record()
        torch.cuda.synchronize()
        us = 1e3 * s.elapsed_time(e) / reps
        nbytes = N * M * 4.0 * (2 + 2 * d)
        print(json.dumps({"N": N, "M": M, "d": d, "us": round(us, 2), "us_per_256_trajectories": round(us * 256.0 / N, 2),
                          "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 3)}), flush=True)


if __name__ == "__main__":
    main()
