"""When do the workgroups of the K2 measurement kernel finish, by XCD?  Builds a library of its own with
-DMMF_K2_WG_STAMPS (multimodalfilter_amd/libmmf_hip_stamps.so), runs a few headline-sized measurement launches through the
engine and reads the last launch's per-workgroup stamps.

    python scripts/debug/k2_wg_spread.py --build        # here (hipcc cross-compiles)
    python scripts/debug/k2_wg_spread.py                # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "multimodalfilter_amd", "libmmf_hip_stamps.so")


def build():
    sys.path.insert(0, ROOT)
    from multimodalfilter_amd import build as b

    objs = []
    out = os.path.join(b.CSRC, "_obj_stamps")
    os.makedirs(out, exist_ok=True)
    for src in b.SOURCES:
        obj = os.path.join(out, src.replace(".hip", ".o"))
        subprocess.run(["/opt/rocm/bin/hipcc", *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), "-DMMF_K2_WG_STAMPS", "-c", "-o", obj,
                        os.path.join(b.CSRC, src)], check=True)
        objs.append(obj)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    print(LIB)


def main():
    if "--build" in sys.argv:
        return build()
    os.environ["MMF_LIB_PATH"] = LIB
    sys.path.insert(0, ROOT)
    import torch
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, engine

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    N, M = 256, 4096
    model = mmf.door_models.DoorMeasurementModel(modalities={"pos", "sensors"}).to(dev).eval()
    states = torch.randn((N, M, 3), device=dev)
    bias = torch.randn((N, 64), device=dev)
    ll = torch.empty((N, M), device=dev)
    for _ in range(30):
        engine.run_measure(model._net, states, bias, None, 0, ll, False)
    torch.cuda.synchronize()
    lib = _abi.load()
    st = (ctypes.c_longlong * 1024)()
    xcc = (ctypes.c_int * 512)()
    assert lib.mmf_debug_k2_stamps(st, xcc) == 0
    n = 256
    t0 = min(st[i] for i in range(n))
    t1 = max(st[512 + i] for i in range(n))
    print(f"measurement kernel, {N} x {M}: first start -> last end {(t1 - t0) * 0.01:.1f} us over {n} workgroups")
    for x in range(8):
        ends = [(st[512 + i] - t0) * 0.01 for i in range(n) if xcc[i] == x]
        if ends:
            print(f"  XCD {x}: {len(ends):3d} workgroups end {min(ends):6.1f} .. {max(ends):6.1f} us (idle after its last: {100 * (t1 - t0 - max(ends) / 0.01) / (t1 - t0):4.1f} %)")


if __name__ == "__main__":
    main()
