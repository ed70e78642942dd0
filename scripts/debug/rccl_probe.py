"""Does RCCL run on this box: 1 rank, and 2 ranks sharing the one GPU?  (debug probe)"""
import os
import subprocess
import sys

import torch


def worker():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    x = torch.full((4, 3), float(rank + 1), device="cuda:0")
    out = [torch.empty_like(x) for _ in range(world)]
    dist.all_gather(out, x)
    y = torch.ones(1000, device="cuda:0") * (rank + 1)
    dist.all_reduce(y)
    torch.cuda.synchronize()
    print("rank", rank, "of", world, "all_gather", [float(o[0, 0]) for o in out], "all_reduce", float(y[0]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if os.environ.get("RCCL_PROBE_WORKER"):
        worker()
        sys.exit(0)
    for world in (1, 2):
        procs = []
        for r in range(world):
            env = dict(os.environ, RCCL_PROBE_WORKER="1", RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(29731 + world), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, __file__], env=env))
        rcs = []
        for p in procs:
            try:
                rcs.append(p.wait(timeout=120))
            except subprocess.TimeoutExpired:
                p.kill()
                rcs.append("timeout")
        print("world", world, "exit codes", rcs, flush=True)
