#!/usr/bin/env python3
"""weight_grad_h_kernel (f16 recompute buffers, tile-scaled dz, f16 MFMA) against an fp64 product of the same data."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multimodalfilter_amd import _abi

lib = _abi.load()
fn = getattr(lib, "_Z27mmf_internal_weight_grads_hPKvPKfS0_PfS3_iiiiPv")
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for R, S, NL1 in ((120, 1, 3), (960, 15, 8), (4096, 64, 8), (100000, 128, 2)):
    dz = torch.randn((NL1, R, 64), device=dev, generator=g)
    ntile = (R + 31) // 32
    tile_scale = torch.rand((NL1, ntile), device=dev, generator=g) * 10 + 0.1
    sc = tile_scale.repeat_interleave(32, dim=1)[:, :R].contiguous()
    dz_rel = (dz / dz.abs().amax(dim=(1, 2), keepdim=True)).to(torch.float16)
    stash = torch.relu(torch.randn((NL1, R, 64), device=dev, generator=g)).to(torch.float16)
    pw = torch.zeros((NL1, S, 64, 64), device=dev)
    pb = torch.zeros((NL1, S, 64), device=dev)
    for acc in (0, 1):
        rc = fn(dz_rel.data_ptr(), sc.data_ptr(), stash.data_ptr(), pw.data_ptr(), pb.data_ptr(), NL1, R, S, acc, None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    a = dz_rel.double() * sc.double()[:, :, None]
    want_w = 2 * torch.einsum("lro,lri->loi", a, stash.double())
    want_b = 2 * a.sum(1)
    got_w, got_b = pw.sum(1).double(), pb.sum(1).double()
    print(R, S, "dW max rel err", float((got_w - want_w).abs().max() / want_w.abs().max()), "db", float((got_b - want_b).abs().max() / want_b.abs().max()))
