"""Where do the device-to-device copies of a bench.py run come from?

rocprofv3's kernel stats of the headline command list ``__amd_rocclr_copyBuffer`` (hipMemcpyAsync D2D,
what ``Tensor.copy_`` / ``.clone()`` / ``.contiguous()`` of a strided view become).  This runs ``bench.main()``
under ``torch.profiler`` with Python stacks and prints every ``aten::copy_`` / ``aten::clone`` that spent device
time, grouped by the innermost frame of this repository.

    python scripts/debug/find_copies.py [bench.py flags ...]  > gpurun_out/find_copies.txt
"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    import bench

    sys.argv = ["bench.py"] + (sys.argv[1:] or ["--steps", "20", "--warmup", "5"])
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        bench.main()
    by_site = collections.defaultdict(lambda: [0, 0.0, 0])
    for ev in prof.events():
        if ev.name not in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::_to_copy"):
            continue
        dev_us = getattr(ev, "device_time_total", 0.0) or getattr(ev, "cuda_time_total", 0.0)
        if dev_us <= 0:
            continue
        site = "?"
        for fr in ev.stack or ():
            if ROOT in fr or "bench" in fr or "multimodalfilter_amd" in fr:
                site = fr.replace(ROOT + "/", "")
                break
        rec = by_site[(ev.name, site)]
        rec[0] += 1
        rec[1] += dev_us
    rows = sorted(by_site.items(), key=lambda kv: -kv[1][1])
    print(f"{'calls':>7} {'device us':>12}  op  site", file=sys.stderr)
    for (name, site), (n, us, _) in rows[:60]:
        print(f"{n:7d} {us:12.1f}  {name}  {site}", file=sys.stderr)
    kinds = collections.Counter()
    for ev in prof.events():
        if "Memcpy" in ev.name or "copyBuffer" in ev.name:
            kinds[ev.name] += 1
    print("memcpy-like device activities:", dict(kinds), file=sys.stderr)


if __name__ == "__main__":
    main()
