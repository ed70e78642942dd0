"""Builder / runner for per-trajectory MLP programs (K7, ``csrc/traj_program.hip``).

A model describes its N-row network once as a list of LOAD / LINEAR / STORE instructions
over LDS vector slots; ``run`` is then ONE HIP launch, whatever the number of layers.
Weights are gathered (transposed, padded) from the owning ``nn.Module`` parameters into one
device blob that is rebuilt lazily when a parameter changes.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _abi
from .layers import ResLinear

Src = Tuple[int, int, int]  # (slot, first feature, width)


class TrajProgram:
    def __init__(self):
        self._instrs: List[_abi.MmfTrajInstr] = []
        self._blob_parts = []      # (kind, parameter, cols, out_pad)
        self._blob_floats = 0
        self._io: Dict[str, int] = {}
        self._free = list(range(_abi.TRAJ_SLOTS))
        self._prog_dev = None
        self._blob = None
        self._stamp = None

    # ---------------------------------------------------------------- slots / io
    def alloc(self) -> int:
        assert self._free, "out of LDS vector slots"
        return self._free.pop(0)

    def free(self, slot: int):
        assert slot not in self._free
        self._free.insert(0, slot)

    def _io_index(self, name: str) -> int:
        if name not in self._io:
            assert len(self._io) < _abi.TRAJ_MAX_IO, "too many program inputs/outputs"
            self._io[name] = len(self._io)
        return self._io[name]

    def _emit(self, **kw) -> _abi.MmfTrajInstr:
        I = _abi.MmfTrajInstr()
        for i in range(4):
            I.src[i], I.src_off[i], I.src_dim[i] = -1, 0, 0
        I.res, I.b_off, I.act, I.fparam = -1, -1, _abi.ACT_NONE, 0.0
        for k, v in kw.items():
            setattr(I, k, v)
        self._instrs.append(I)
        return I

    # ---------------------------------------------------------------- instructions
    def load(self, name: str, dim: int, stride: Optional[int] = None, off: int = 0) -> int:
        slot = self.alloc()
        self._emit(op=_abi.TRAJ_LOAD, dst=slot, out_dim=dim, io=self._io_index(name),
                   io_stride=dim if stride is None else stride, io_off=off)
        return slot

    def linear(self, srcs: Sequence[Src], lin: nn.Linear, act: int = _abi.ACT_NONE,
               res: Optional[int] = None, dst: Optional[int] = None,
               cols: Optional[Tuple[int, int]] = None, bias: bool = True) -> int:
        """``dst = act(W[:, cols] cat(srcs) + b (+ res))``; ``cols`` selects input columns of ``lin``."""
        in_total = sum(w for _, _, w in srcs)
        c0, c1 = cols if cols is not None else (0, lin.in_features)
        assert c1 - c0 == in_total, (c0, c1, in_total)
        out_dim = lin.out_features
        assert out_dim <= 128 and len(srcs) <= 4
        out_pad = 64 if out_dim <= 64 else 128
        w_off = self._blob_floats
        dims = [w for _, _, w in srcs]
        self._blob_parts.append(("wT", lin.weight, (c0, c1, dims), out_pad))
        self._blob_floats += sum(-(-w // 16) * 16 for w in dims) * out_pad
        b_off = -1
        if bias and lin.bias is not None:
            b_off = self._blob_floats
            self._blob_parts.append(("b", lin.bias, None, 128))
            self._blob_floats += 128
        slot = self.alloc() if dst is None else dst
        I = self._emit(op=_abi.TRAJ_LINEAR, dst=slot, out_dim=out_dim, w_off=w_off, b_off=b_off,
                       res=-1 if res is None else res, act=act)
        for i, (s, o, w) in enumerate(srcs):
            assert o % 4 == 0
            I.src[i], I.src_off[i], I.src_dim[i] = s, o, w
        return slot

    def res_linear(self, block: ResLinear, slot: int, width: int) -> int:
        """In place: ``slot = relu(block2(relu(block1(slot))) + slot)``."""
        h = self.linear([(slot, 0, width)], block.block1, _abi.ACT_RELU)
        self.linear([(h, 0, width)], block.block2, _abi.ACT_RELU, res=slot, dst=slot)
        self.free(h)
        return slot

    def vector_encoder(self, seq: nn.Sequential, src: int, in_dim: int) -> int:
        """``Linear, ReLU, ResLinear`` (``layers.vector_encoder``) -> new slot (64 wide)."""
        assert isinstance(seq[0], nn.Linear) and isinstance(seq[2], ResLinear)
        x = self.linear([(src, 0, in_dim)], seq[0], _abi.ACT_RELU)
        return self.res_linear(seq[2], x, seq[0].out_features)

    def store(self, name: str, slot: int, dim: int, stride: Optional[int] = None, off: int = 0,
              act: int = _abi.ACT_NONE, fparam: float = 0.0, src_off: int = 0, diag: bool = False):
        I = self._emit(op=_abi.TRAJ_STORE_DIAG if diag else _abi.TRAJ_STORE, out_dim=dim,
                       io=self._io_index(name), io_off=off, act=act, fparam=float(fparam),
                       io_stride=(dim * dim if diag else dim) if stride is None else stride)
        I.src[0], I.src_off[0], I.src_dim[0] = slot, src_off, dim

    # ---------------------------------------------------------------- run
    def _refresh(self, device):
        params = [p for _, p, _, _ in self._blob_parts]
        stamp = tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        if self._blob is not None and stamp == self._stamp:
            return
        parts = []
        for kind, p, cols, out_pad in self._blob_parts:
            t = p.detach().to(torch.float32)
            if kind == "wT":
                # MFMA A fragments of v_mfma_f32_16x16x4_f32 (csrc/traj_program.hip): per source, per group of
                # 4 k-steps, per 16-output tile, lane (i, q) holds W[16 mt + i][16 g + 4 ks + q], ks = 0..3
                c0, _c1, dims = cols
                MT = out_pad // 16
                col = c0
                for dim in dims:
                    groups = -(-dim // 16)
                    Wp = torch.zeros((out_pad, 16 * groups), dtype=torch.float32, device=t.device)
                    Wp[: t.shape[0], :dim] = t[:, col:col + dim]
                    frag = Wp.view(MT, 16, groups, 4, 4).permute(2, 0, 4, 1, 3)    # (g, mt, q, i, ks)
                    parts.append(frag.reshape(-1))
                    col += dim
            else:
                pad = torch.zeros(128, dtype=torch.float32, device=t.device)
                pad[: t.numel()] = t
                parts.append(pad)
        self._blob = torch.cat(parts).to(device).contiguous()
        self._stamp = stamp
        if self._prog_dev is None or self._prog_dev.device != self._blob.device:
            n = len(self._instrs)
            raw = (_abi.MmfTrajInstr * n)(*self._instrs)
            host = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).clone()
            self._prog_dev = host.to(self._blob.device)

    def _footprint(self):
        """(slots used, vector width 64 | 128): the launch sizes its LDS slot file for these."""
        slots, width = 1, 0
        for I in self._instrs:
            if I.op in (_abi.TRAJ_LOAD, _abi.TRAJ_LINEAR):
                slots = max(slots, I.dst + 1)
                width = max(width, I.out_dim)
            for k in range(4):
                if I.src[k] >= 0:
                    slots = max(slots, I.src[k] + 1)
                    width = max(width, I.src_off[k] + I.src_dim[k])
            if I.res >= 0:
                slots = max(slots, I.res + 1)
        return slots, (64 if width <= 64 else 128)

    def run(self, tensors: Dict[str, torch.Tensor], R: int):
        """``tensors``: every input and (pre-allocated) output by the names used in the program."""
        assert set(tensors) == set(self._io), (sorted(tensors), sorted(self._io))
        any_t = next(iter(tensors.values()))
        self._refresh(any_t.device)
        io = [None] * len(self._io)
        for name, idx in self._io.items():
            io[idx] = tensors[name]
        _abi.traj_program(self._prog_dev, len(self._instrs), self._blob, io, R, *self._footprint())
