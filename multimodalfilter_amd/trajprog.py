"""Builder / runner for per-trajectory MLP programs (K7, ``csrc/traj_program.hip``).

A model describes its N-row network once as a list of LOAD / LINEAR / STORE instructions
over LDS vector slots; ``run`` is then ONE HIP launch, whatever the number of layers.
Weights are gathered (transposed, padded) on the DEVICE from the owning ``nn.Module`` parameters
where they lie (``mmf_traj_pack``: one launch whenever a parameter has changed).

The reference evaluates these networks as chains of ``nn.Linear`` / ReLU / add launches
(``door_models/layers.py:11-40,66-95``, ``crossmodal_pf.py:74-106``, ``kf.py:81-126``,
``crossmodal_kf.py:134-167``) and differentiates them with torch autograd inside
``torchfilter.train.train_filter`` (``train_helpers.py:124-162``).  Round 5: ``run_autograd`` is that
differentiation in HIP -- the forward program stashes every vector it forms, ``_TrainPlan`` derives a
REVERSE program from the instruction list (same kernel, same slot file), ``mmf_traj_weight_grads`` forms
the parameter gradients from the two stashes.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _abi
from .layers import ResLinear

Src = Tuple[int, int, int]  # (slot, first feature, width)


def _new_instr(**kw) -> _abi.MmfTrajInstr:
    I = _abi.MmfTrajInstr()
    for i in range(4):
        I.src[i], I.src_off[i], I.src_dim[i] = -1, 0, 0
    I.res, I.b_off, I.act, I.fparam, I.dst_off = -1, -1, _abi.ACT_NONE, 0.0, 0
    for k, v in kw.items():
        setattr(I, k, v)
    return I


def _copy_instr(I: _abi.MmfTrajInstr) -> _abi.MmfTrajInstr:
    return _abi.MmfTrajInstr.from_buffer_copy(bytes(I))


def _pack_table(descs, device) -> torch.Tensor:
    raw = (_abi.MmfTrajPackDesc * len(descs))(*descs)
    return torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).clone().to(device)


def _to_device(instrs, device) -> torch.Tensor:
    raw = (_abi.MmfTrajInstr * len(instrs))(*instrs)
    return torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).clone().to(device)


def _footprint_of(instrs):
    """(slots used, vector width 64 | 128): the launch sizes its LDS slot file for these."""
    slots, width = 1, 0
    for I in instrs:
        if I.op in (_abi.TRAJ_LOAD, _abi.TRAJ_LINEAR, _abi.TRAJ_MASK, _abi.TRAJ_ADD, _abi.TRAJ_ZERO, _abi.TRAJ_LOAD_ADD):
            slots = max(slots, I.dst + 1)
            width = max(width, I.dst_off + I.out_dim)
        for k in range(4):
            if I.src[k] >= 0:
                slots = max(slots, I.src[k] + 1)
                width = max(width, I.src_off[k] + I.src_dim[k])
        if I.res >= 0:
            slots = max(slots, I.res + 1)
    return slots, (64 if width <= 64 else 128)


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
        self._train = None         # reverse-mode plan (built on first use)
        self._pack = None          # (parameter addresses, device table of MmfTrajPackDesc, count)
        self._lin_info = {}        # LINEAR instruction index -> (weight, first column, source widths, bias | None)

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
        I = _new_instr(**kw)
        self._instrs.append(I)
        self._train = None
        return I

    # ---------------------------------------------------------------- instructions
    def load(self, name: str, dim: int, stride: Optional[int] = None, off: int = 0, act: int = _abi.ACT_NONE) -> int:
        slot = self.alloc()
        self._emit(op=_abi.TRAJ_LOAD, dst=slot, out_dim=dim, io=self._io_index(name),
                   io_stride=dim if stride is None else stride, io_off=off, act=act)
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
        self._lin_info[len(self._instrs)] = (lin.weight, c0, dims, lin.bias if b_off >= 0 else None)
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
        """Weight blob of the current parameter values: ONE pack launch when a parameter has changed (their addresses
        do not: optimisers update in place, so the descriptor table is built once)."""
        params = [p for _, p, _, _ in self._blob_parts]
        stamp = tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        if self._blob is not None and stamp == self._stamp:
            return
        device = torch.device(device)
        # parameters that do not live on the device as contiguous fp32 (a host-resident user model) are staged: the table
        # then follows the copies' addresses and is rebuilt with them
        srcs = [p if (p.dtype == torch.float32 and p.is_contiguous() and p.device == device)
                else p.detach().to(device=device, dtype=torch.float32).contiguous() for p in params]
        self._pack_keep = srcs
        addr = tuple((p.data_ptr(), str(p.device)) for p in srcs)
        if self._pack is None or self._pack[0] != addr:
            descs, off = [], 0
            for (kind, _p, cols, out_pad), p in zip(self._blob_parts, srcs):
                if kind == "wT":
                    c0, _c1, dims = cols
                    col = c0
                    for dim in dims:
                        descs.append(_abi.MmfTrajPackDesc(src=p.data_ptr(), kind=_abi.TRAJ_PACK_LAYER, rows=p.shape[0], ld=p.shape[1],
                                                          col0=col, dim=dim, out_pad=out_pad, dst_off=off))
                        off += -(-dim // 16) * 16 * out_pad
                        col += dim
                else:
                    descs.append(_abi.MmfTrajPackDesc(src=p.data_ptr(), kind=_abi.TRAJ_PACK_BIAS, rows=p.numel(), ld=0, col0=0, dim=0,
                                                      out_pad=128, dst_off=off))
                    off += 128
            assert off == self._blob_floats
            self._pack = (addr, _pack_table(descs, device), len(descs))
            self._blob = torch.empty(max(off, 4), dtype=torch.float32, device=device)
        _abi.traj_pack(self._pack[1], self._pack[2], self._blob)
        self._stamp = stamp
        if self._prog_dev is None or self._prog_dev.device != self._blob.device:
            self._prog_dev = _to_device(self._instrs, self._blob.device)

    def _footprint(self):
        return _footprint_of(self._instrs)

    def run(self, tensors: Dict[str, torch.Tensor], R: int):
        """``tensors``: every input and (pre-allocated) output by the names used in the program."""
        assert set(tensors) == set(self._io), (sorted(tensors), sorted(self._io))
        any_t = next(iter(tensors.values()))
        self._refresh(any_t.device)
        io = [None] * len(self._io)
        for name, idx in self._io.items():
            io[idx] = tensors[name]
        _abi.traj_program(self._prog_dev, len(self._instrs), self._blob, io, R, *self._footprint())


    # ---------------------------------------------------------------- reverse mode (K6 for the N-row networks)
    def run_autograd(self, inputs: Dict[str, torch.Tensor], outputs: Dict[str, int], R: int) -> Dict[str, torch.Tensor]:
        """Differentiable ``run``: ``inputs`` by name (those with ``requires_grad`` receive gradients), ``outputs`` =
        ``{name: width}`` of the STOREs to return.  Forward = this program plus a stash of every vector it forms;
        backward = its reverse program and ``mmf_traj_weight_grads`` -- three launches, no library call."""
        assert set(inputs) | set(outputs) == set(self._io), (sorted(inputs), sorted(outputs), sorted(self._io))
        names_in, names_out = tuple(sorted(inputs)), tuple(sorted(outputs))
        grad_in = tuple(n for n in names_in if inputs[n].requires_grad)
        plan = self._training_plan(grad_in, names_out)
        outs = TrajProgramFunction.apply(self, plan, R, names_in, names_out, tuple(outputs[n] for n in names_out),
                                         *[inputs[n] for n in names_in], *plan.params)
        return dict(zip(names_out, outs))

    def _training_plan(self, grad_inputs, out_names):
        key = (tuple(grad_inputs), tuple(out_names))
        if self._train is None:
            self._train = {}
        if key not in self._train:
            self._train[key] = _TrainPlan(self, grad_inputs, out_names)
        return self._train[key]


def _ceil4(n: int) -> int:
    return -(-n // 4) * 4


class _TrainPlan:
    """Forward-with-stash and reverse instruction lists of a ``TrajProgram`` plus the descriptors of its parameter
    gradients.  Gradient vectors live in the slot of the value they belong to; a slot's gradient is complete when the
    reverse walk reaches the instruction that defined the value (every use lies behind it)."""

    def __init__(self, prog: TrajProgram, grad_inputs, out_names):
        F = prog._instrs
        io_name = {idx: name for name, idx in prog._io.items()}
        self.prog = prog
        # ---- values and their stash columns
        cur: Dict[int, int] = {}          # slot -> value id
        vals = []                          # value id -> dict(width, col, grad)
        snap = []                          # per instruction: value ids it reads / defines
        S = 0
        for k, I in enumerate(F):
            if I.op == _abi.TRAJ_LOAD:
                vals.append(dict(width=I.out_dim, col=S, grad=io_name[I.io] in grad_inputs))
                S += _ceil4(I.out_dim)
                cur[I.dst] = len(vals) - 1
                snap.append(dict(dst=cur[I.dst]))
            elif I.op == _abi.TRAJ_LINEAR:
                srcs = [cur[I.src[j]] for j in range(4) if I.src[j] >= 0]
                res = cur[I.res] if I.res >= 0 else None
                assert I.act in (_abi.ACT_NONE, _abi.ACT_RELU), "reverse mode covers ReLU / identity layers"
                assert all(I.src[j] != I.dst for j in range(4)), "in-place LINEAR over its own source"
                vals.append(dict(width=I.out_dim, col=S, grad=True))
                S += _ceil4(I.out_dim)
                cur[I.dst] = len(vals) - 1
                snap.append(dict(srcs=srcs, res=res, dst=cur[I.dst]))
            else:
                assert I.op == _abi.TRAJ_STORE, "reverse mode covers plain STOREs"
                snap.append(dict(src=cur[I.src[0]]))
        self.stash_ld = max(S, 4)
        # ---- forward with stash
        fio = dict(prog._io)
        fio["__stash"] = len(fio)
        assert len(fio) <= _abi.TRAJ_MAX_IO
        fwd = []
        for k, I in enumerate(F):
            fwd.append(_copy_instr(I))
            if I.op in (_abi.TRAJ_LOAD, _abi.TRAJ_LINEAR):
                v = vals[snap[k]["dst"]]
                st = _new_instr(op=_abi.TRAJ_STORE, out_dim=I.out_dim, io=fio["__stash"], io_stride=self.stash_ld, io_off=v["col"])
                st.src[0], st.src_off[0], st.src_dim[0] = I.dst, 0, I.out_dim
                fwd.append(st)
            elif io_name[I.io] in out_names:
                assert I.act == _abi.ACT_NONE, "differentiable outputs are stored as they are"
        self.fwd, self.fwd_io = fwd, fio
        # ---- reverse program
        bio: Dict[str, int] = {}

        def io(name):
            if name not in bio:
                bio[name] = len(bio)
                assert len(bio) <= _abi.TRAJ_MAX_IO, "too many inputs / outputs for the reverse program"
            return bio[name]

        io("__stash"), io("__dz")
        bwd = []
        live = set()                       # slots whose gradient vector has been started
        parts = []                         # transposed-layer blob
        blob_floats = 0
        self.desc = []
        regions = []                       # (parameter, offset into the flat gradient)
        n_grads = 0
        Z = 0

        def region(p):
            nonlocal n_grads
            regions.append((p, n_grads))
            n_grads += _ceil4(p.numel())
            return regions[-1][1]

        def emit(**kw):
            srcs = kw.pop("srcs", None)
            I = _new_instr(**kw)
            if srcs:
                for j, (sl, off, dim) in enumerate(srcs):
                    I.src[j], I.src_off[j], I.src_dim[j] = sl, off, dim
            bwd.append(I)
            return I

        def accumulate_from_io(slot, off, dim, width, name, stride, io_off):
            if slot in live:
                emit(op=_abi.TRAJ_LOAD_ADD, dst=slot, dst_off=off, out_dim=dim, io=io(name), io_stride=stride, io_off=io_off)
                return
            if off != 0 or dim != width:
                emit(op=_abi.TRAJ_ZERO, dst=slot, out_dim=width)
                emit(op=_abi.TRAJ_LOAD_ADD, dst=slot, dst_off=off, out_dim=dim, io=io(name), io_stride=stride, io_off=io_off)
            else:
                emit(op=_abi.TRAJ_LOAD, dst=slot, out_dim=dim, io=io(name), io_stride=stride, io_off=io_off)
            live.add(slot)

        for k in range(len(F) - 1, -1, -1):
            I, sn = F[k], snap[k]
            if I.op == _abi.TRAJ_STORE:
                name = io_name[I.io]
                if name in out_names and vals[sn["src"]]["grad"]:
                    accumulate_from_io(I.src[0], I.src_off[0], I.out_dim, vals[sn["src"]]["width"], "d:" + name,
                                       I.io_stride, I.io_off)
            elif I.op == _abi.TRAJ_LINEAR:
                v = vals[sn["dst"]]
                if I.dst not in live:      # an output nothing downstream depends on
                    emit(op=_abi.TRAJ_ZERO, dst=I.dst, out_dim=I.out_dim)
                    live.add(I.dst)
                if I.act == _abi.ACT_RELU:
                    emit(op=_abi.TRAJ_MASK, dst=I.dst, out_dim=I.out_dim, io=io("__stash"), io_stride=self.stash_ld, io_off=v["col"])
                dz_col = Z
                Z += _ceil4(I.out_dim)
                emit(op=_abi.TRAJ_STORE, out_dim=I.out_dim, io=io("__dz"), io_stride=0, io_off=dz_col,
                     srcs=[(I.dst, 0, I.out_dim)])          # io_stride patched below, once Z is known
                weight, c0, dims, bias = prog._lin_info[k]
                w_region = region(weight)
                b_region = region(bias) if bias is not None else -1
                col = c0
                for j, vid in enumerate(sn["srcs"]):
                    sl, off, dim = I.src[j], I.src_off[j], I.src_dim[j]
                    xv = vals[vid]
                    self.desc.append(_abi.MmfTrajGradDesc(x_col=xv["col"] + off, x_dim=dim, dz_col=dz_col, out_dim=I.out_dim,
                                                          grad_off=w_region + col, grad_ld=weight.shape[1],
                                                          bias_off=b_region if j == 0 else -1, reserved=0))
                    if xv["grad"]:          # g[source] += W_s^T dz: a LINEAR over the transposed block
                        out_pad = 64 if dim <= 64 else 128
                        # the kernel's LINEAR epilogue stores whole padded tiles (out_pad columns from dst_off, all four
                        # waves): a sub-range source must sit on a 64-column boundary and its padded tile must stay inside
                        # the vector it belongs to, or the store would run into the next row of the slot
                        assert off % 64 == 0 and off + out_pad <= -(-xv["width"] // 64) * 64, (
                            f"reverse LINEAR into columns {off}..{off + dim} of a {xv['width']}-wide vector: sub-range sources "
                            "that receive a gradient must start at a multiple of 64 and their 64 / 128-column tile must fit the vector")
                        parts.append((weight, col, dim, I.out_dim, out_pad))
                        started = sl in live
                        if not started and (off != 0 or dim != xv["width"]):
                            emit(op=_abi.TRAJ_ZERO, dst=sl, out_dim=xv["width"])
                            started = True
                        emit(op=_abi.TRAJ_LINEAR, dst=sl, dst_off=off, out_dim=dim, w_off=blob_floats, b_off=-1,
                             res=sl if started else -1, srcs=[(I.dst, 0, I.out_dim)])
                        blob_floats += -(-I.out_dim // 16) * 16 * out_pad
                        live.add(sl)
                    col += dim
                if I.res >= 0 and I.res != I.dst:
                    if I.res in live:
                        emit(op=_abi.TRAJ_ADD, dst=I.res, out_dim=I.out_dim, srcs=[(I.dst, 0, I.out_dim)])
                    else:
                        rw = vals[sn["res"]]["width"]
                        emit(op=_abi.TRAJ_ZERO, dst=I.res, out_dim=rw)
                        emit(op=_abi.TRAJ_ADD, dst=I.res, out_dim=I.out_dim, srcs=[(I.dst, 0, I.out_dim)])
                        live.add(I.res)
                if I.res != I.dst:          # in place over its residual: dz IS the gradient of the value it replaced
                    live.discard(I.dst)
            else:  # LOAD
                v = vals[sn["dst"]]
                if v["grad"]:
                    if I.dst not in live:
                        emit(op=_abi.TRAJ_ZERO, dst=I.dst, out_dim=I.out_dim)
                    if I.act == _abi.ACT_RELU:
                        emit(op=_abi.TRAJ_MASK, dst=I.dst, out_dim=I.out_dim, io=io("__stash"), io_stride=self.stash_ld, io_off=v["col"])
                    else:
                        assert I.act == _abi.ACT_NONE
                    emit(op=_abi.TRAJ_STORE, out_dim=I.out_dim, io=io("g:" + io_name[I.io]), io_stride=I.io_stride, io_off=I.io_off,
                         srcs=[(I.dst, 0, I.out_dim)])
                live.discard(I.dst)
        self.dz_ld = max(Z, 4)
        for I in bwd:
            if I.op == _abi.TRAJ_STORE and I.io == bio["__dz"]:
                I.io_stride = self.dz_ld
        self.bwd, self.bwd_io, self.bwd_parts = bwd, bio, parts
        self.n_grads = max(n_grads, 4)
        self.regions = regions
        # one gradient per distinct parameter, in first-use order
        self.params = []
        for p, _off in regions:
            if all(p is not q for q in self.params):
                self.params.append(p)
        self.grad_inputs, self.out_names = tuple(grad_inputs), tuple(out_names)
        self._dev = None
        self._tpack = None

    def device_state(self, device):
        """(forward program, reverse program, descriptors) on ``device`` + the transposed blob of the current weights."""
        if self._dev is None or self._dev[0].device != torch.device(device):
            raw = (_abi.MmfTrajGradDesc * len(self.desc))(*self.desc)
            desc = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).clone().to(device)
            self._dev = (_to_device(self.fwd, device), _to_device(self.bwd, device), desc)
        return self._dev

    def transposed_blob(self, device) -> torch.Tensor:
        """The reverse program's layers from the current parameter values: one pack launch."""
        device = torch.device(device)
        if not self.bwd_parts:
            return torch.zeros(4, dtype=torch.float32, device=device)
        addr = tuple((w.data_ptr(), str(w.device)) for w, *_ in self.bwd_parts)
        if self._tpack is None or self._tpack[0] != addr:
            descs, off = [], 0
            for weight, col, dim, out_dim, out_pad in self.bwd_parts:
                assert weight.dtype == torch.float32 and weight.is_contiguous() and weight.device == device
                descs.append(_abi.MmfTrajPackDesc(src=weight.data_ptr(), kind=_abi.TRAJ_PACK_TRANSPOSED, rows=dim, ld=weight.shape[1],
                                                  col0=col, dim=out_dim, out_pad=out_pad, dst_off=off))
                off += -(-out_dim // 16) * 16 * out_pad
            self._tpack = (addr, _pack_table(descs, device), len(descs), torch.empty(off, dtype=torch.float32, device=device))
        _abi.traj_pack(self._tpack[1], self._tpack[2], self._tpack[3])
        return self._tpack[3]


class TrajProgramFunction(torch.autograd.Function):
    """``TrajProgram.run_autograd``: the program forward (stashing every vector), its reverse program and the
    parameter-gradient kernel backward.  Replaces torch autograd over ~15 ``nn.Linear`` per model
    (``train_helpers.py:124-162`` through ``door_models/layers.py:11-40``, ``crossmodal_pf.py:74-106``)."""

    @staticmethod
    def forward(ctx, prog, plan, R, names_in, names_out, out_dims, *tensors):
        inputs = [t.detach().to(torch.float32).contiguous() for t in tensors[:len(names_in)]]
        dev = inputs[0].device
        prog._refresh(dev)
        fwd, _bwd, _desc = plan.device_state(dev)
        outs = [torch.empty((R, d), dtype=torch.float32, device=dev) for d in out_dims]
        stash = torch.empty((R, plan.stash_ld), dtype=torch.float32, device=dev)
        io = [None] * len(plan.fwd_io)
        for n, t in zip(names_in, inputs):
            io[plan.fwd_io[n]] = t
        for n, t in zip(names_out, outs):
            io[plan.fwd_io[n]] = t
        io[plan.fwd_io["__stash"]] = stash
        _abi.traj_program(fwd, len(plan.fwd), prog._blob, io, R, *_footprint_of(plan.fwd))
        ctx.prog, ctx.plan, ctx.R, ctx.names_in, ctx.names_out, ctx.out_dims = prog, plan, R, names_in, names_out, out_dims
        ctx.in_shapes = [t.shape for t in tensors[:len(names_in)]]
        # the reverse program multiplies by the transposed blob, which is packed from the parameters' CURRENT values at
        # backward time: the parameters are not saved tensors (they are read in place), so autograd's own version check does
        # not cover them -- keep their version counters and refuse a backward across an in-place update, as torch would
        ctx.param_versions = [p._version for p in plan.params]
        ctx.save_for_backward(stash)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        plan, R = ctx.plan, ctx.R
        (stash,) = ctx.saved_tensors
        dev = stash.device
        for p, v in zip(plan.params, ctx.param_versions):
            if p._version != v:
                raise RuntimeError(
                    "one of the variables needed for gradient computation has been modified by an inplace operation: a "
                    f"parameter of shape {tuple(p.shape)} of a per-trajectory program is at version {p._version}; expected "
                    f"version {v} (the reverse program reads the weights in place: call backward before optimizer.step())")
        _fwd, bwd, desc = plan.device_state(dev)
        io = [None] * len(plan.bwd_io)
        io[plan.bwd_io["__stash"]] = stash
        dz = torch.empty((R, plan.dz_ld), dtype=torch.float32, device=dev)
        io[plan.bwd_io["__dz"]] = dz
        for n, g, d in zip(ctx.names_out, gouts, ctx.out_dims):
            if "d:" + n in plan.bwd_io:
                io[plan.bwd_io["d:" + n]] = (torch.zeros((R, d), dtype=torch.float32, device=dev) if g is None
                                             else g.to(torch.float32).contiguous())
        d_in = {}
        for n, shape in zip(ctx.names_in, ctx.in_shapes):
            if "g:" + n in plan.bwd_io:
                d_in[n] = torch.empty(shape, dtype=torch.float32, device=dev)
                io[plan.bwd_io["g:" + n]] = d_in[n]
        _abi.traj_program(bwd, len(plan.bwd), plan.transposed_blob(dev), io, R, *_footprint_of(plan.bwd))
        grads = torch.zeros(plan.n_grads, dtype=torch.float32, device=dev)
        n_slices = 1 if R <= 64 else min(64, -(-R // 64))   # the kernel is a chain of loads per 4 rows: short runs, summed in order
        # zeros: the columns of a layer that the program does not multiply (the per-particle half of a join layer) are summed too
        partials = torch.zeros((n_slices, plan.n_grads), dtype=torch.float32, device=dev) if n_slices > 1 else None
        _abi.traj_weight_grads(desc, len(plan.desc), stash, dz, grads, partials, n_slices, R)
        per_param = []
        for p in plan.params:
            g = None
            for q, off in plan.regions:
                if q is p:
                    piece = grads[off:off + p.numel()].view(p.shape)
                    g = piece if g is None else g + piece
            per_param.append(g)
        return (None,) * 6 + tuple(d_in.get(n) for n in ctx.names_in) + tuple(per_param)
