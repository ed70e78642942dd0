"""Where do the device-to-device copies of a bench.py run come from?

rocprofv3's kernel stats of the headline command list ``__amd_rocclr_copyBuffer`` (hipMemcpyAsync D2D: what
``Tensor.copy_`` / ``.clone()`` / ``.contiguous()`` of same-dtype dense tensors become).  This runs ``bench.main()``
under a ``TorchDispatchMode`` that sees every ``aten::copy_`` / ``clone`` / ``_to_copy`` between CUDA tensors and books
it to the innermost Python frame of this repository.

    python scripts/debug/find_copies.py [bench.py flags ...]  2> gpurun_out/find_copies.txt
"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

SITES = collections.defaultdict(lambda: [0, 0])


class CopyFinder(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if name.split(".")[0] in ("copy_", "clone", "_to_copy", "contiguous"):
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            if ts and all(t.is_cuda for t in ts):
                site = "?"
                for fr in reversed(traceback.extract_stack()[:-1]):
                    if fr.filename.startswith(ROOT) and "find_copies" not in fr.filename:
                        site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} ({fr.name})"
                        break
                same = len(ts) < 2 or (ts[0].dtype == ts[1].dtype)
                rec = SITES[(name, site, "memcpy-shaped" if same and all(t.is_contiguous() for t in ts) else "kernel")]
                rec[0] += 1
                rec[1] += ts[0].numel() * ts[0].element_size()
        return func(*args, **(kwargs or {}))


def main():
    import bench

    sys.argv = ["bench.py"] + (sys.argv[1:] or ["--steps", "20", "--warmup", "5"])
    with CopyFinder():
        bench.main()
    rows = sorted(SITES.items(), key=lambda kv: -kv[1][0])
    print(f"{'calls':>7} {'MB':>10}  op  kind  site", file=sys.stderr)
    for (name, site, kind), (n, b) in rows[:80]:
        print(f"{n:7d} {b / 1e6:10.2f}  {name}  {kind}  {site}", file=sys.stderr)


if __name__ == "__main__":
    main()
