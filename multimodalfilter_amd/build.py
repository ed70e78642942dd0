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
SOURCES = ["abi.hip", "pf_resample.hip", "ekf.hip", "particle_net.hip", "image_encoder.hip", "traj_program.hip", "pf_loop.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def _deps():
    files = [os.path.join(CSRC, s) for s in SOURCES]
    files += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    files.append(os.path.join(os.path.dirname(HERE), "include", "mmf.h"))
    return files


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > built for f in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, *FLAGS, "-o", LIB + ".tmp", *[os.path.join(CSRC, s) for s in SOURCES]]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
