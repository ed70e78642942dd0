#!/usr/bin/env python3
"""The backward data path of the native training recursion on f16x3 (per-tile power-of-two scaling) against the same
kernel with exact fp32 products, on the door filter's own networks: d_states and the stored dz (f16 x tile scale)
for gradients of ordinary size and for gradients of 1e-9 (which plain f16 operands would flush)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multimodalfilter_amd as mmf
from multimodalfilter_amd import _abi, engine

lib = _abi.load()
bwd = getattr(lib, "_Z29mmf_internal_train_backward_hPKfiS0_iiPKjS0_PvPfS4_iiS3_")
bwd.restype = ctypes.c_int
bwd.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
torch.manual_seed(0)
f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev)
g = torch.Generator(device=dev).manual_seed(1)
d = 3
for name, net, kind in (("dynamics", f.dynamics_model._net, 0), ("measurement", f.measurement_model.measurement_models[0]._net, 1)):
    NL, n_out = 3 + 2 * net.n_res, net.n_out
    head_w = net._sources()[-2].detach().float().contiguous()
    for R, mag in ((960, 1.0), (4096, 1e-9), (100000, 1e-4)):
        mask = torch.randint(-2**31, 2**31 - 1, (NL + 1, R, 2), device=dev, generator=g, dtype=torch.int64).to(torch.int32)
        rowscale = torch.exp(torch.randn((R, 1), device=dev, generator=g) * 2)  # rows of one tile two orders of magnitude apart
        d_out = torch.randn((R, n_out), device=dev, generator=g) * mag * rowscale
        res = {}
        for prec in (_abi.PREC_F32, _abi.PREC_F16X3):
            blob = engine._transposed_blob(net, prec)
            dz = torch.zeros((NL + 1, R, 64), dtype=torch.float16, device=dev)
            sc = torch.zeros((NL + 1, R), device=dev)
            ds = torch.zeros((R, d), device=dev)
            rc = bwd(blob.data_ptr(), prec, head_w.data_ptr(), net.n_res, kind, mask.data_ptr(), d_out.data_ptr(), dz.data_ptr(),
                     sc.data_ptr(), ds.data_ptr(), R, d, None)
            assert rc == 0, rc
            torch.cuda.synchronize()
            res[prec] = (ds.double(), dz.double() * sc.double()[:, :, None])
        (s0, z0), (s1, z1) = res[_abi.PREC_F32], res[_abi.PREC_F16X3]
        # per ROW: a row's error against that row's own largest entry (rows of a tile share a scale)
        row = lambda a, b: float(((a - b).abs().amax(-1) / a.abs().amax(-1).clamp_min(1e-300)).max())
        print(f"{name:12s} R={R:6d} |d_out|~{mag:g}: d_states max rel err {float((s0 - s1).abs().max() / s0.abs().max()):.2e} "
              f"(worst row {row(s0, s1):.2e}); dz {float((z0 - z1).abs().max() / z0.abs().max()):.2e} (worst row {row(z0, z1):.2e}); finite {bool(torch.isfinite(s1).all())}")
