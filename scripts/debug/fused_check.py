"""GPU check of mmf_particle_net_train_fused against the exact-fp32 step kernels (engine.ParticleNetFunction):
every output of one fused network call -- d states, the compact rows of the narrow reductions, the weight / bias
partials -- per tensor, relative to the tensor's largest entry.

    python scripts/debug/fused_check.py [--sizes 3x40,9x1000,32x8192]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from _fused_case import run_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="3x40,2x64,5x7,9x1000,32x30,32x8192")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--timing", action="store_true")
    args = ap.parse_args()
    worst = 0.0
    for size in args.sizes.split(","):
        N, M = (int(v) for v in size.split("x"))
        for task in ("door", "push"):
            for kind in ("measure", "dynamics"):
                worst = max(worst, run_case(task, kind, N, M, verbose=not args.quiet, timing=args.timing and N * M >= 100000))
    print(f"worst over all cases: {worst:.2e}")


if __name__ == "__main__":
    main()
