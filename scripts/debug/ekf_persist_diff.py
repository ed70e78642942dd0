#!/usr/bin/env python3
"""Persistent EKF loop against the loop of launches: where do the bits part?  (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multimodalfilter_amd as mmf  # noqa: E402
from multimodalfilter_amd import engine  # noqa: E402

dev = torch.device("cuda:0")
for cls, tname, prec in (("DoorKalmanFilter", "door", "f16x3"), ("DoorKalmanFilter", "door", "f32"), ("DoorCrossmodalKalmanFilter", "door", "f16x3"),
                         ("PushKalmanFilter", "push", "f16x3")):
    engine.set_default_precision(prec)
    d = 3 if tname == "door" else 2
    N, T = 7, 5
    g = torch.Generator().manual_seed(31)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1).to(dev),
           "gripper_pos": torch.randn((T, N, 3), generator=g).to(dev), "gripper_sensors": torch.randn((T, N, 7), generator=g).to(dev)}
    ctrl = torch.randn((T, N, 7), generator=g).to(dev)
    x0 = torch.randn((N, d), generator=g).to(dev)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d).to(dev)
    torch.manual_seed(0)
    f = mmf.model_types(tname)[cls]().to(dev).eval()
    outs = {}
    for persistent in (False, True):
        engine.EKF_PERSISTENT = persistent
        f.initialize_beliefs(mean=x0, covariance=cov)
        outs[persistent] = f.forward_loop(observations=obs, controls=ctrl)
        subs = list(f.filter_models) if hasattr(f, "filter_models") else [f]
        outs[(persistent, "S")] = torch.stack([m._belief_covariance for m in subs])
    a, b = outs[False], outs[True]
    print(cls, prec, "estimates differ in", int((a.view(torch.int32) != b.view(torch.int32)).sum()), "of", a.numel(), "values; per step:",
          [int((a[t].view(torch.int32) != b[t].view(torch.int32)).sum()) for t in range(T)], "max abs", float((a - b).abs().max()),
          "cov differ", int((outs[(False, 'S')].view(torch.int32) != outs[(True, 'S')].view(torch.int32)).sum()))
    if not torch.equal(a, b):
        t0 = [t for t in range(T) if not torch.equal(a[t], b[t])][0]
        print("  first differing step", t0, "\n  launches", a[t0].flatten()[:9].tolist(), "\n  persist ", b[t0].flatten()[:9].tolist())
