#!/usr/bin/env python3
"""The 8192 -> 64 linear layer + ResLinear tail behind the resident K4 kernel: one launch (fc_fused_f16x3_kernel) against
partial sums + tail (MMF_K4_FC_TWO_LAUNCHES=1).  Saves the features of a fixed batch to argv[1] (bit comparison across the
two processes) and prints the time of the encoder launch sequence."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodalfilter_amd import engine, layers  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
res = {}
for n_img, nets in ((4096, 2), (1000, 3), (37, 1), (1, 2)):
    encs = [layers.image_encoder(64).to(dev) for _ in range(nets)]
    img = (torch.randn((n_img, 32, 32), device=dev) * 0.5).clamp(-1, 1)
    for _ in range(3):
        out = engine.encode_images(encs, img)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        out = engine.encode_images(encs, img)
    e.record()
    torch.cuda.synchronize()
    res[f"{n_img}x{nets}"] = torch.stack(out).cpu()
    print(f"two_launches={os.environ.get('MMF_K4_FC_TWO_LAUNCHES', '0')} {n_img} x {nets}: {s.elapsed_time(e) / 20:.4f} ms per launch sequence", flush=True)
torch.save(res, sys.argv[1])
