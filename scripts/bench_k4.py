#!/usr/bin/env python3
"""K4 alone: image encoders on a resident batch, the fused kernels in f16x3 (default) and bf16 against the exact-fp32
per-layer kernels (``f32``: the mode the golden vectors pin).  Prints one JSON line per configuration; compares outputs."""
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
    modes = tuple(os.environ.get("K4_MODES", "fused,bf16,f32").split(","))
    for n_img, nets in shapes:
        encs = [layers.image_encoder(64).to(dev) for _ in range(nets)]
        img = (torch.randn((n_img, 32, 32), device=dev) * 0.5).clamp(-1, 1)
        outs = {}
        for mode in modes:
            engine.set_image_encoder_precision("bf16" if mode == "bf16" else ("f32" if mode == "f32" else None))
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
        if "f32" not in outs or "fused" not in outs or "bf16" not in outs:
            continue
        scale = max(1.0, float(outs["f32"].abs().max()))
        print(json.dumps({"images": n_img, "nets": nets,
                          "fused_f16x3_vs_f32_max_rel": float((outs["fused"] - outs["f32"]).abs().max()) / scale,
                          "bf16_vs_f32_max_rel": float((outs["bf16"] - outs["f32"]).abs().max()) / scale}), flush=True)
    engine.set_image_encoder_precision(None)


if __name__ == "__main__":
    main()
