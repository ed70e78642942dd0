"""Experiment: K4 launch sequences of consecutive image chunks on ONE stream against TWO streams (the tail of a chunk's
persistent grids under the head of the next chunk's).  python scripts/debug/k4_two_streams.py [images per chunk] [nets]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodalfilter_amd import _abi, engine, layers  # noqa: E402


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    nets = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    chunks = 8
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    encs = [layers.image_encoder(64).to(dev) for _ in range(nets)]
    packs = [engine.PackedImageEncoder(e).blob() for e in encs]
    img = [(torch.randn((n_img, 32, 32), device=dev) * 0.5).clamp(-1, 1) for _ in range(chunks)]
    feat = [torch.empty((nets, n_img, 64), device=dev) for _ in range(chunks)]
    need = _abi.image_encoder_workspace_bytes(n_img, nets)
    ws = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(2)]
    flag = engine.range_flag(dev)
    prec = engine.image_encoder_precision_code()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def run(two):
        if not two:
            for c in range(chunks):
                _abi.image_encoder(packs, img[c], feat[c], ws[0], flag, prec, _abi.ENCODER_DEFAULT)
            return
        cur = torch.cuda.current_stream()
        for s in streams:
            s.wait_stream(cur)
        for c in range(chunks):
            with torch.cuda.stream(streams[c & 1]):
                _abi.image_encoder(packs, img[c], feat[c], ws[c & 1], flag, prec, _abi.ENCODER_DEFAULT)
        for s in streams:
            cur.wait_stream(s)

    ref = None
    for two in (False, True, False, True):
        for _ in range(2):
            run(two)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            run(two)
        e.record()
        torch.cuda.synchronize()
        out = torch.stack(feat).clone()
        if ref is None:
            ref = out
        print(f"{'two streams' if two else 'one stream '}: {s.elapsed_time(e) / 5 / chunks:.4f} ms per chunk of {n_img} x {nets}; "
              f"identical to the first run: {bool(torch.equal(out, ref))}")


if __name__ == "__main__":
    main()
