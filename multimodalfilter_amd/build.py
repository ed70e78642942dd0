"""Compile the HIP sources in ``csrc/`` into ``libmmf_hip.so`` (gfx950 only, in-tree).

    python -m multimodalfilter_amd.build [--force]

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``.so`` is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmf_hip.so")
SOURCES = ["abi.hip", "pf_resample.hip", "pf_init.hip", "ekf.hip", "ukf.hip", "particle_net.hip", "particle_net_fused.hip", "image_encoder.hip", "traj_program.hip", "traj_train.hip", "pf_loop.hip", "ekf_loop.hip", "pf_train_loop.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# particle_net.hip: with the SLP vectoriser on, the f16x3 operand split (x - float(hi) -> f16)
# becomes cvt + cvt + v_pk_add_f32 + cvt; without it the same source selects
# v_fma_mixlo_f16 / v_fma_mixhi_f16 (2 instructions per pair fewer; K2 is VALU-issue bound)
# particle_net_fused.hip: its kernels run one wave per SIMD with > 256 registers; by default hipcc then puts every MFMA
# result in AGPRs and copies it to VGPRs for the VALU work between layers (~170 v_accvgpr_* per layer and tile);
# with VGPR-form MFMAs the chain stays in VGPRs and only the parked state (weight-gradient accumulators, f16 copies of
# the layer inputs) moves through AGPRs
EXTRA_FLAGS = {"particle_net.hip": ["-fno-slp-vectorize"], "particle_net_fused.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"], "image_encoder.hip": ["-fno-slp-vectorize"]}
OBJ = os.path.join(CSRC, "_obj")


def _headers():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "mmf.h"))
    files.append(os.path.abspath(__file__))
    return files


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + _headers()


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > built for f in _deps())


def _compile(hipcc, src, verbose):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + _headers()
    if os.path.exists(obj) and all(os.path.getmtime(f) <= os.path.getmtime(obj) for f in deps):
        return obj
    cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", "-o", obj, os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(lambda src: _compile(hipcc, src, verbose), SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
