#!/usr/bin/env python3
"""K4 alone: image encoders on a resident batch, fused (default) against the per-layer kernels
(MMF_K4_UNFUSED=1).  Prints one JSON line per configuration; compares the two paths' outputs."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodalfilter_amd import engine, layers  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    reps = int(os.environ.get("REPS", "10"))
    shapes = ((4096, 2), (4096, 3), (2048, 2), (1024, 3), (256, 2), (32, 3))
    if os.environ.get("K4_SHAPES"):  # e.g. K4_SHAPES=4096x2,1024x3
        shapes = tuple(tuple(int(v) for v in sh.split("x")) for sh in os.environ["K4_SHAPES"].split(","))
    modes = tuple(os.environ.get("K4_MODES", "fused,bf16,unfused").split(","))
    for n_img, nets in shapes:
        encs = [layers.image_encoder(64).to(dev) for _ in range(nets)]
        img = (torch.randn((n_img, 32, 32), device=dev) * 0.5).clamp(-1, 1)
        outs = {}
        for mode in modes:
            engine.set_image_encoder_precision("bf16" if mode == "bf16" else None)
            if mode == "unfused":
                os.environ["MMF_K4_UNFUSED"] = "1"
            else:
                os.environ.pop("MMF_K4_UNFUSED", None)
            for _ in range(3):
                out = engine.encode_images(encs, img)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                out = engine.encode_images(encs, img)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / reps
            outs[mode] = torch.stack(out)
            flops = engine.image_encoder_flops(n_img) * nets
            print(json.dumps({"mode": mode, "images": n_img, "nets": nets, "ms": round(ms, 4),
                              "ns_per_image_encoder": round(1e6 * ms / (n_img * nets), 1),
                              "algorithmic_tflops": round(flops / ms / 1e9, 1)}), flush=True)
        if "unfused" not in outs or "fused" not in outs or "bf16" not in outs:
            continue
        scale = max(1.0, float(outs["unfused"].abs().max()))
        print(json.dumps({"images": n_img, "nets": nets,
                          "fused_vs_unfused_max_rel": float((outs["fused"] - outs["unfused"]).abs().max()) / scale,
                          "bf16_vs_unfused_max_rel": float((outs["bf16"] - outs["unfused"]).abs().max()) / scale}), flush=True)
    os.environ.pop("MMF_K4_UNFUSED", None)


if __name__ == "__main__":
    main()
