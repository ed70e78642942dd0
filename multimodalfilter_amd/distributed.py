"""Multi-GPU layout: independent trajectories shard across ranks, one process per GPU.

Nothing in the filter recursions mixes batch elements, so the data path has
no collective: rank ``r`` owns trajectories ``[r*N/P, (r+1)*N/P)`` for all ``T`` and the
model weights are replicated.  The only exchange is the evaluation statistic: an
all-gather of the per-sequence squared-error partials ``(N_local, d)`` (RCCL over xGMI when
the backend is ``"nccl"``; KB-scale, latency-bound).  Particles of one trajectory never
split across GPUs, so the resampling prefix sum stays inside one workgroup.
"""
import os
from typing import Dict, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; no-op for one process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm; MMF_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
            backend = os.environ.get("MMF_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _single() -> bool:
    """True when there is nothing to exchange.  ``MMF_DIST_FORCE_COLLECTIVES=1`` sends a one-rank group through
    the collectives anyway (the one-GPU box's RCCL test: RCCL refuses two ranks on one device)."""
    if not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("MMF_DIST_FORCE_COLLECTIVES") != "1"


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced ownership of the trajectory axis (first ranks get the remainder)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_trajectories(traj: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Slice every ``(T, N, ...)`` tensor along ``N`` (dim 1)."""
    n = next(iter(traj.values())).shape[1]
    lo, hi = shard_bounds(n, rank, world)
    return {k: v[:, lo:hi].contiguous() for k, v in traj.items()}


def all_gather_rows(x: torch.Tensor, total_rows: int = None) -> torch.Tensor:
    """Concatenate per-rank ``(N_local, ...)`` tensors along dim 0 (ragged shards allowed).

    ``total_rows``: the global row count, when the caller knows it (the evaluation path does: the shards are
    ``shard_bounds(total_rows, rank, world)``).  If it divides evenly every rank holds ``total_rows / world`` rows and
    knows that the others do -- ONE collective; otherwise the shard sizes are exchanged first (two)."""
    if _single():
        return x
    world = dist.get_world_size()
    # gloo moves host memory; RCCL moves device memory over xGMI
    comm_dev = x.device if dist.get_backend() == "nccl" else torch.device("cpu")
    if total_rows is not None and total_rows % world == 0:
        assert x.shape[0] == total_rows // world, (x.shape, total_rows, world)
        mine = x.to(comm_dev).contiguous()
        out = torch.empty((total_rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=comm_dev)
        dist.all_gather_into_tensor(out, mine)
        return out.to(x.device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=comm_dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([x.shape[0]], dtype=torch.int64, device=comm_dev))
    sizes = [int(s) for s in sizes]
    pad = max(sizes)
    buf = torch.zeros((pad,) + tuple(x.shape[1:]), dtype=x.dtype, device=comm_dev)
    buf[: x.shape[0]] = x.to(comm_dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0).to(x.device)


def max_over_ranks(value: float, device) -> float:
    if _single():
        return value
    comm_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=comm_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t)


def barrier():
    if not _single():
        dist.barrier()


def all_reduce_gradients(module: torch.nn.Module, average: bool = True, marks: list = None) -> int:
    """Data-parallel gradient exchange (X2): ONE flat all-reduce of every parameter gradient
    (<= 1.43 M fp32 = 5.7 MB for the largest filter; a ring over xGMI is per-link bound,
    2(P-1)/P * S / 153 GB/s ~ 65 us at P = 8 -- nothing to overlap at this size).

    Device work per exchange: one pack kernel (``torch.cat`` of the gradients into a fresh flat buffer), the in-place
    collective on that buffer, one scale kernel -- and NO copy back: every ``p.grad`` is re-pointed at its slice of the
    reduced buffer (views; ``optimizer.step()`` reads them, ``zero_grad(set_to_none=True)`` drops them).  Round 5 copied
    the result back with one ``copy_`` launch per parameter tensor (100+ per step around a 30-65 us collective).
    ``marks``: a list that receives the CUDA events ``(start, packed, reduced, end)`` of this call (``bench.py
    --workload push_train`` reports the collective apart from the pack / scale around it).
    Returns the number of elements reduced."""
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return 0
    if _single():
        return sum(p.numel() for p in params)
    dev = params[0].device
    ev = None
    if marks is not None and dev.type == "cuda":
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    if ev:
        ev[1].record()
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)      # in place, device memory, RCCL over xGMI
        if ev:
            ev[2].record()
    else:                                                # gloo moves host memory
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
        if ev:
            ev[2].record()
    if average:
        flat.div_(dist.get_world_size())
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    if ev:
        ev[3].record()
        marks.append(tuple(ev))
    return off
