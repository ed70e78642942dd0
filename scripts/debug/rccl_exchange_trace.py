"""What does ONE data-parallel gradient exchange launch?  (one collective + <= 2 kernels.)

One-rank RCCL group on the one GPU of the box (RCCL refuses two ranks on one device): the filter's real parameters
(~140 tensors) with gradients set, ``distributed.all_reduce_gradients`` five times.  Run plainly it counts the device
kernels of each exchange with torch.profiler; under ``rocprofv3 --kernel-trace --stats`` the kernel stats of the same
process are the trace the verdict asks for (profiles/r06/rccl_exchange_kernel_stats.csv).

    MMF_DIST_FORCE_COLLECTIVES=1 python scripts/debug/rccl_exchange_trace.py
"""
import os
import sys

os.environ.setdefault("MMF_DIST_FORCE_COLLECTIVES", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29757")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import distributed

    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    f = mmf.push_models.PushUnimodalParticleFilter().cuda()
    params = [p for p in f.parameters() if p.requires_grad]
    for p in params:
        p.grad = torch.randn_like(p)
    for _ in range(2):
        distributed.all_reduce_gradients(f)   # RCCL's first call builds its communicator
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    per = []
    for _ in range(5):
        for p in params:
            p.grad = torch.randn_like(p)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            n = distributed.all_reduce_gradients(f)
            torch.cuda.synchronize()
        per.append([e.name for e in prof.events() if e.device_type.name != "CPU" and "Memcpy" not in e.name and "Memset" not in e.name])
    print(f"parameters: {len(params)} tensors, {n} elements; device kernels per exchange:")
    for names in per:
        print(f"  {len(names)}: " + " | ".join(x[:70] for x in names))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
