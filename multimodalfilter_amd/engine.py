"""Bridges ``nn.Module`` parameters to the HIP kernels: fragment-ordered weight blobs for
the per-particle networks, and the launches of K2 / K5 (``csrc/particle_net.hip``).
"""
import ctypes
import os
from typing import List, Sequence

import torch
import torch.nn as nn

from . import _abi
from .layers import ResLinear


# MAC per row of each per-particle network: enc 64*(d) [+bias via the MFMA],
# 2 + 1 + 2*n_res layers of 64x64, head 64*n_out.
def particle_net_macs(d_in: int, n_res: int, n_out: int) -> int:
    return 64 * d_in + (3 + 2 * n_res) * 64 * 64 + 64 * n_out


class KernelTimer:
    """HIP-event timing of individual kernel launches on the launching stream (used by
    ``bench.py`` for the roofline figure; off by default, no cost when off)."""

    def __init__(self, prealloc: int = 512, loop_stride: int = 1):
        self.records = {}
        # native step loops record events on every loop_stride-th step only: each record is a
        # barrier packet between two kernels (~6 % of the headline step when taken everywhere)
        self.loop_stride = loop_stride
        # hipEventCreate is slow the first time (~0.1 ms each): build the pool up front, outside
        # any timed region, and touch every event once
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * prealloc)]
        for e in self._pool:
            e.record()
        torch.cuda.synchronize()

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def launch(self, name: str, flops: float, nbytes: float, fn):
        start = self._event()
        end = self._event()
        start.record()
        fn()
        end.record()
        self.records.setdefault(name, []).append((start, end, flops, nbytes))

    def loop_events(self, count: int):
        """``count`` created events for a native step loop to record (see ``pf_loop.hip``)."""
        return [self._event() for _ in range(count)]

    def add_loop_records(self, names, work, events):
        """Register the (start, end) pairs a native loop recorded: ``names`` / ``work`` describe
        one step's launches in order; ``events`` is the flat list handed to the loop."""
        per_step = 2 * len(names)
        for t in range(len(events) // per_step):
            for j, (name, (flops, nbytes)) in enumerate(zip(names, work)):
                s, e = events[t * per_step + 2 * j], events[t * per_step + 2 * j + 1]
                self.records.setdefault(name, []).append((s, e, flops, nbytes))

    def summary(self):
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _, _ in recs]
            out[name] = {"launches": len(recs), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms),
                         "flops_per_launch": sum(r[2] for r in recs) / len(recs),
                         "bytes_per_launch": sum(r[3] for r in recs) / len(recs)}
        return out


_TIMER = None


def set_kernel_timer(timer):
    global _TIMER
    _TIMER = timer


def kernel_timer():
    return _TIMER


def _timed(name, flops, nbytes, fn):
    if _TIMER is None:
        fn()
    else:
        _TIMER.launch(name, flops, nbytes, fn)


def reserve_memory(device, nbytes: int):
    """Grow PyTorch's caching allocator by ``nbytes`` ahead of time (allocate + free one block).
    A first ``forward_loop`` at a new size otherwise pays several ``hipMalloc`` calls, each of
    which drains the stream (measured: 7 segment allocations = +24 ms on the first 32-step
    loop of the headline workload)."""
    if nbytes > 0:
        block = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        # requests under 1 MB come from the allocator's separate pool of 2 MB segments
        # (estimates, per-step scalars, program outputs): grow that one as well
        small = [torch.empty(512 << 10, dtype=torch.uint8, device=device) for _ in range(32)]
        del block, small


def require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _abi.MmfError(
            f"{what}: tensors must live on the MI355X (got {t.device}); the filter hot path has "
            "no CPU implementation -- move the module and its inputs to 'cuda'"
        )


# Arithmetic of the per-particle 64x64 layers: "f32" (exact fp32 products on the f32 MFMA) or
# "f16x3" (operands split into two halves, three f16 MFMA products, fp32 accumulate; ~2^-22).
DEFAULT_PRECISION = os.environ.get("MMF_PRECISION", "f16x3")


def set_default_precision(name: str):
    global DEFAULT_PRECISION
    assert name in _abi.PRECISIONS, name
    DEFAULT_PRECISION = name


# Arithmetic of the image encoders (K4): None follows DEFAULT_PRECISION; "bf16" = single bf16
# products with fp32 accumulation in the two fused convolution kernels (BASELINE config 5's "bf16
# measurement CNN on MFMA"; a reduced-precision mode, ~1e-2 relative on the features -- opt-in).
IMAGE_ENCODER_PRECISION = os.environ.get("MMF_K4_PRECISION") or None


def set_image_encoder_precision(name):
    global IMAGE_ENCODER_PRECISION
    assert name is None or name in _abi.IMAGE_PRECISIONS, name
    IMAGE_ENCODER_PRECISION = name


def image_encoder_precision_code() -> int:
    return _abi.IMAGE_PRECISIONS[IMAGE_ENCODER_PRECISION or DEFAULT_PRECISION]


def training_image_precision_code() -> int:
    """Arithmetic of the image encoders' TRAINING forward (``ImageConvsFunction``): exact fp32 products (per-layer kernels;
    the reference trains in fp32, and the engine-wide default does not reach training) unless the image-encoder precision
    was set explicitly -- ``set_image_encoder_precision("bf16")`` is BASELINE config 5, ``"f16x3"`` the inference
    arithmetic; both run the resident K4 kernel with every activation the backward reads kept on the way (round 6: one launch,
    84 us per 512 images against 359 us for the five per-layer kernels, profiles/r06/train_refsize_fwd_ab.txt)."""
    return _abi.IMAGE_PRECISIONS[IMAGE_ENCODER_PRECISION] if IMAGE_ENCODER_PRECISION else _abi.PREC_F32


# Small particle-filter loops (mmf_pf_persistent_plan > 0: e.g. the reference's 32 x 300 evaluation) as ONE persistent
# launch per forward_loop: role-specialised workgroups keep one network's weights in LDS for all T steps and hand the
# particles over through L2 (csrc/pf_persistent.inc); bit-identical to the launch-per-step loop.  "0": A/B, off.
PF_PERSISTENT = os.environ.get("MMF_PF_PERSISTENT", "1") not in ("", "0")
# The EKF step loop likewise (mmf_ekf_persistent_plan > 0; csrc/ekf_persistent.inc): a wave owns 8 trajectories of one
# sub-filter for all T steps, K > 1 sub-filters meet once per step through L2; bit-identical to the 2 T launches.
EKF_PERSISTENT = os.environ.get("MMF_EKF_PERSISTENT", "1") not in ("", "0")


# Training is opt-in: nothing switches paths silently, and eval() always means the forward-only HIP
# path.  With a training backend set, modules in train() mode evaluate differentiably:
#   "hip"       the N*M-row work (per-particle dynamics / measurement networks: ParticleNetFunction;
#               reweight + estimate: ReweightEstimateFunction) and every EKF's Kalman algebra
#               (EkfStepFunction) run forward AND backward in HIP (K6); the per-trajectory networks
#               (encoders, CNNs, weight models, Jacobians) through torch autograd
#   "autograd"  torch ops throughout (the reference's own formulation; the cross-check)
TRAINING_BACKEND = os.environ.get("MMF_TRAINING_BACKEND") or None


def set_training_backend(name):
    """``None``: forward-only HIP path everywhere.  ``"autograd"``: modules in ``train()`` mode
    evaluate through differentiable torch ops.  ``"hip"``: as ``"autograd"``, but the per-particle
    networks (the ``N*M``-row work) go through ``ParticleNetFunction`` (K6: HIP forward with
    stash + HIP backward data path)."""
    global TRAINING_BACKEND
    assert name in (None, "autograd", "hip"), name
    TRAINING_BACKEND = name


def use_autograd(module: nn.Module) -> bool:
    return TRAINING_BACKEND in ("autograd", "hip") and module.training and torch.is_grad_enabled()


def use_hip_backward() -> bool:
    return TRAINING_BACKEND == "hip"


_RANGE_FLAGS = {}


def range_flag(device) -> torch.Tensor:
    """Per-device sticky int32: the f16x3 kernels OR 1 into it when an activation left the
    range in which the two-half split is exact."""
    key = str(device)
    if key not in _RANGE_FLAGS:
        _RANGE_FLAGS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _RANGE_FLAGS[key]


_PERSISTENT_WARNED = False


def persistent_loop_gave_up(device) -> bool:
    """After a persistent particle-filter loop: did it abort (bit 2 of the range flag: a hand-off's bounded spin ran
    out because a workgroup of the launch was not resident, e.g. another process shares the GPU)?  Clears the bit,
    switches the persistent form off for the rest of the process and warns once; the caller re-runs the loop as a
    loop of launches.  One 4-byte device->host read."""
    global PF_PERSISTENT, EKF_PERSISTENT, _PERSISTENT_WARNED
    flag = _RANGE_FLAGS.get(str(device))
    if flag is None:
        return False
    bits = int(flag.item())
    if not bits & 4:
        return False
    flag.bitwise_and_(~4)
    PF_PERSISTENT = EKF_PERSISTENT = False
    if not _PERSISTENT_WARNED:
        _PERSISTENT_WARNED = True
        import warnings
        warnings.warn("a persistent filter loop gave up waiting for a hand-off (a workgroup of its launch was not "
                      "resident -- is another process using this GPU?); this forward_loop is re-run as a loop of launches and the "
                      "persistent forms are switched off for this process (MMF_PF_PERSISTENT=0 MMF_EKF_PERSISTENT=0 do so from the start)")
    return True


def run_ekf_loop(a, mu: torch.Tensor, Sigma: torch.Tensor):
    """``mmf_ekf_forward_loop`` on filled ``MmfEkfLoopArgs`` (``mu`` / ``Sigma``: its in / out belief tensors): as ONE
    persistent launch where the problem is eligible (``EKF_PERSISTENT``), with the loop of launches as the fallback if a
    hand-off of that launch timed out (the belief is restored first)."""
    import ctypes

    dev = mu.device
    keep = None
    if (EKF_PERSISTENT and not CAPTURING and a.T > 0 and a.d in (2, 3) and a.n_res_dyn == 3
            and _abi.ekf_persistent_plan(a.N, a.K) > 0):
        n_words = _abi.ekf_persistent_sync_words(a.N, a.K, a.d)
        sync = torch.empty(n_words, dtype=torch.int32, device=dev)  # tagged granules of the hand-offs (zeroed by the call)
        keep = (sync, mu.clone(), Sigma.clone()) if a.K > 1 else (sync, None, None)  # (one sub-filter: no hand-offs, nothing can time out)
        a.persistent, a.n_sync_words = 1, n_words
        a.sync_words = ctypes.c_void_p(_abi.ptr(sync, dtype=torch.int32))
        a.range_flag = ctypes.c_void_p(_abi.ptr(range_flag(dev), dtype=torch.int32))  # bit 2: "a hand-off timed out"
    _abi.ekf_forward_loop(a, mu)
    if a.persistent and a.K > 1 and persistent_loop_gave_up(dev):
        mu.copy_(keep[1])
        Sigma.copy_(keep[2])
        a.persistent = 0
        _abi.ekf_forward_loop(a, mu)


# While a training step is being captured into a hipGraph (train.GraphedFilterStep) nothing may read the device: the checks
# that cost a host read -- the range flag at the end of a forward_loop, "covariance not positive definite" in
# initialize_beliefs -- only accumulate in the range flag (bits 1 / 16) and are read ONCE after every replay.
CAPTURING = False


def check_range(device):
    """Raise if any f16x3 launch since the last check saturated its operand split (one
    4-byte device->host read; filters call it once per ``forward_loop`` / on demand)."""
    flag = _RANGE_FLAGS.get(str(device))
    bits = 0 if flag is None else int(flag.item())
    if bits & 16:
        flag.zero_()
        raise ValueError("initialize_beliefs: covariance is not positive definite (reported by a captured training step)")
    if bits & 4:
        flag.zero_()
        raise _abi.MmfError(
            "the persistent particle-filter loop gave up waiting for a hand-off (a workgroup of its launch was not "
            "resident -- is another process using this GPU?): results of this forward_loop are invalid; "
            "MMF_PF_PERSISTENT=0 selects the launch-per-step loop")
    if bits != 0:
        flag.zero_()
        raise _abi.MmfError(
            "an activation exceeded the f16x3 operand range (|x| >= 65504): results of the last "
            "filter steps are invalid; set MMF_PRECISION=f32 / engine.set_default_precision('f32')")


import threading

_CHECK = threading.local()  # .depth[device]: > 0 while a checked forward_loop / forward is running ON THIS THREAD for that device


def _checked(fn, *, step: bool):
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = None
        for p in self.parameters():
            dev = p.device
            break
        # a bare training step runs the differentiable path (torch ops / exact-fp32 K6 kernels): no launch of it writes
        # the f16x3 flag, so there is nothing to clear or to read back (one blocking .item() per step otherwise) -- UNLESS
        # the image encoders' training forward was given a reduced precision (ImageConvsFunction then splits operands and
        # hands the flag to mmf_image_convs_train_forward), in which case the step is checked like any other
        if dev is None or dev.type != "cuda" or (step and use_autograd(self) and training_image_precision_code() == _abi.PREC_F32):
            return fn(self, *args, **kwargs)
        if CAPTURING:  # a hipGraph capture: no host read; the flag accumulates and is read after the replay
            return fn(self, *args, **kwargs)
        depth = getattr(_CHECK, "depth", None)
        if depth is None:
            depth = _CHECK.depth = {}
        key = str(dev)
        if depth.get(key, 0) > 0:  # an enclosing checked call on this thread and device owns the check
            return fn(self, *args, **kwargs)
        clear_range(dev)
        depth[key] = 1
        try:
            out = fn(self, *args, **kwargs)
        finally:
            depth[key] = 0
        check_range(dev)
        return out

    return wrapper


def checked_loop(fn):
    """Decorator for ``forward_loop`` methods: the f16x3 range flag (raised by K2 / K4 launches whose
    operands left the f16 range) is cleared on entry and checked on the way out, so a loop reports
    its own launches -- one 4-byte device->host read per loop."""
    return _checked(fn, step=False)


def checked_step(fn):
    """Decorator for a filter's ``forward``: a bare step (no ``forward_loop`` around it) checks the range
    flag itself, so an out-of-range f16x3 operand raises at the step that produced it instead of handing
    saturated numbers to the caller -- one 4-byte device->host read per step of a caller-driven loop;
    steps inside a checked ``forward_loop`` (or inside another filter's step) on the same thread and device leave
    the check to it; a training step on the differentiable path launches nothing that writes the flag and skips it."""
    return _checked(fn, step=True)


def clear_range(device):
    """Forget range reports of earlier work (another filter's failed step, say): a loop's check
    speaks about its own launches."""
    flag = _RANGE_FLAGS.get(str(device))
    if flag is not None:
        flag.zero_()


class PackedParticleNet:
    """One per-particle network (``enc -> join -> res* -> head``, see ``include/mmf.h``)
    packed into MFMA-fragment order on the device.

    The blob is rebuilt lazily whenever a source parameter changed (``Tensor._version``),
    moved, or was replaced, so optimiser steps and ``load_state_dict`` are picked up.
    """

    def __init__(self, *, encoder: nn.Sequential, join: nn.Linear, join_state_off: int,
                 res_blocks: Sequence[ResLinear], head: nn.Linear, relu_after_join: bool):
        enc_in, enc_res = encoder[0], encoder[2]
        assert isinstance(enc_in, nn.Linear) and isinstance(enc_res, ResLinear)
        assert enc_in.out_features == _abi.MMF_UNITS and join.out_features == _abi.MMF_UNITS
        assert len(res_blocks) <= _abi.MMF_MAX_RES
        self.d_in = enc_in.in_features
        self.n_res = len(res_blocks)
        self.n_out = head.out_features
        self.relu_after_join = bool(relu_after_join)
        self.join = join
        self.join_state_off = int(join_state_off)
        self._enc_in, self._enc_res = enc_in, enc_res
        self._res = list(res_blocks)
        self._head = head
        self._blobs = {}       # precision code -> (stamp, blob)
        self.precision = None  # None -> engine.DEFAULT_PRECISION at call time

    def precision_code(self) -> int:
        return _abi.PRECISIONS[self.precision or DEFAULT_PRECISION]

    def _sources(self) -> List[torch.Tensor]:
        ts = [self._enc_in.weight, self._enc_in.bias,
              self._enc_res.block1.weight, self._enc_res.block1.bias,
              self._enc_res.block2.weight, self._enc_res.block2.bias, self.join.weight]
        for rb in self._res:
            ts += [rb.block1.weight, rb.block1.bias, rb.block2.weight, rb.block2.bias]
        ts += [self._head.weight, self._head.bias]
        return ts

    def blob(self, precision: int = None) -> torch.Tensor:
        src = self._sources()
        prec = self.precision_code() if precision is None else precision
        stamp = tuple((t.data_ptr(), t._version, str(t.device)) for t in src)
        cached = self._blobs.get(prec)
        if cached is not None and cached[0] == stamp:
            return cached[1]
        dev = src[0].device
        require_device(src[0], "PackedParticleNet")
        keep = [t.detach().to(torch.float32).contiguous() for t in src]
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        d = _abi.MmfParticleNetDesc()
        d.d_in, d.n_res, d.relu_after_join, d.n_out = self.d_in, self.n_res, int(self.relu_after_join), self.n_out
        d.join_in, d.join_state_off = self.join.in_features, self.join_state_off
        d.w_in, d.b_in = P(keep[0]), P(keep[1])
        d.w_enc[0], d.b_enc[0], d.w_enc[1], d.b_enc[1] = P(keep[2]), P(keep[3]), P(keep[4]), P(keep[5])
        d.w_join = P(keep[6])
        for i in range(self.n_res):
            base = 7 + 4 * i
            d.w_res[2 * i], d.b_res[2 * i] = P(keep[base]), P(keep[base + 1])
            d.w_res[2 * i + 1], d.b_res[2 * i + 1] = P(keep[base + 2]), P(keep[base + 3])
        d.w_head, d.b_head = P(keep[-2]), P(keep[-1])
        blob = torch.empty(_abi.particle_net_floats(self.n_res), dtype=torch.float32, device=dev)
        _abi.pack_particle_net(d, blob, prec)
        self._blobs[prec] = (stamp, blob)
        return blob


def _transposed_blob(net: PackedParticleNet, precision: int = _abi.PREC_F32) -> torch.Tensor:
    """Blob of the TRANSPOSED 64x64 layers (K6 backward data path: ``dx = W^T dz`` is a forward
    layer with ``W^T``) in the given arithmetic's fragment order; cached like the forward blobs."""
    src = net._sources()
    stamp = tuple((t.data_ptr(), t._version, str(t.device)) for t in src)
    key = "transposed" if precision == _abi.PREC_F32 else f"transposed_{precision}"
    cached = net._blobs.get(key)
    if cached is not None and cached[0] == stamp:
        return cached[1]
    dev = src[0].device
    f32 = lambda t: t.detach().to(torch.float32)
    T = lambda t: f32(t).t().contiguous()
    zeros = torch.zeros(_abi.MMF_UNITS, dtype=torch.float32, device=dev)
    off = net.join_state_off
    keep = [f32(src[0]).contiguous(), f32(src[1]).contiguous(), T(src[2]), T(src[4]),
            f32(src[6])[:, off:off + _abi.MMF_UNITS].t().contiguous()]
    for i in range(net.n_res):
        keep += [T(src[7 + 4 * i]), T(src[9 + 4 * i])]
    keep += [f32(src[-2]).contiguous(), f32(src[-1]).contiguous(), zeros]
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    d = _abi.MmfParticleNetDesc()
    d.d_in, d.n_res, d.relu_after_join, d.n_out = net.d_in, net.n_res, int(net.relu_after_join), net.n_out
    d.join_in, d.join_state_off = _abi.MMF_UNITS, 0
    d.w_in, d.b_in = P(keep[0]), P(keep[1])
    d.w_enc[0], d.b_enc[0], d.w_enc[1], d.b_enc[1] = P(keep[2]), P(zeros), P(keep[3]), P(zeros)
    d.w_join = P(keep[4])
    for i in range(2 * net.n_res):
        d.w_res[i], d.b_res[i] = P(keep[5 + i]), P(zeros)
    d.w_head, d.b_head = P(keep[-3]), P(keep[-2])
    blob = torch.empty(_abi.particle_net_floats(net.n_res), dtype=torch.float32, device=dev)
    _abi.pack_particle_net(d, blob, precision)
    net._blobs[key] = (stamp, blob)
    return blob


class ParticleNetFunction(torch.autograd.Function):
    """K6: differentiable evaluation of a per-particle network on ``N*M`` rows.

    ``apply(net, kind, N, M, states (R, d), traj_bias (N, 64), *net._sources())`` returns the
    head outputs ``(R, n_out)`` (bias included, before the gate / log-weight epilogue).  The
    forward kernel stashes every layer's input; the backward kernel runs the transposed
    residual network over the masks and writes the pre-activation gradients; the reductions
    over particles (weight, bias, per-trajectory-bias and state gradients) are GEMMs over the
    two stashes.  Exact fp32 (f32 MFMA)."""

    @staticmethod
    def forward(ctx, net, kind, N, M, states, traj_bias, *params):
        require_device(states, "ParticleNetFunction")
        R, d = states.shape
        assert R == N * M and traj_bias.shape == (N, _abi.MMF_UNITS)
        NL = 3 + 2 * net.n_res
        st = states.detach().to(torch.float32).contiguous()
        tb = traj_bias.detach().to(torch.float32).contiguous()
        stash = torch.empty((NL + 1, R, _abi.MMF_UNITS), dtype=torch.float32, device=states.device)
        mask = torch.empty((NL + 1, R, 2), dtype=torch.int32, device=states.device)  # ReLU sign bits of the stash
        out = torch.empty((R, net.n_out), dtype=torch.float32, device=states.device)
        _abi.particle_net_train_forward(net.blob(_abi.PREC_F32), net.n_res, kind, st, tb, stash, mask, out, N, M, d)
        ctx.net, ctx.kind, ctx.N, ctx.M = net, kind, N, M
        ctx.save_for_backward(st, stash, mask, *[p.detach() for p in params])
        return out

    @staticmethod
    def backward(ctx, d_out):
        net, kind, N, M = ctx.net, ctx.kind, ctx.N, ctx.M
        st, stash, mask, *params = ctx.saved_tensors
        R, d = st.shape
        NL = 3 + 2 * net.n_res
        d_out = d_out.to(torch.float32).contiguous()
        head_w = params[-2].to(torch.float32).contiguous()
        dz = torch.empty_like(stash)
        d_states = torch.empty_like(st)
        _abi.particle_net_train_backward(_transposed_blob(net), head_w, net.n_res, kind, mask, d_out, dz,
                                         d_states, R, d)
        grads = [None] * len(params)
        U = _abi.MMF_UNITS
        # every reduction over the R particles of the 64x64 layers in one launch: dW_l = dz_l^T
        # stash_l and db_l = column sums, as per-slice partials (slot NL of dW pairs unrelated
        # tensors and is ignored; its db is the first layer's bias gradient)
        S = max(1, min(128, R // 64))  # >= 64 rows per slice; 32 x 8192 particles: 64 slices 51.8 ms / training step, 128 49.9, 256 49.9
        pw = torch.empty((NL + 1, S, U, U), dtype=torch.float32, device=st.device)
        pb = torch.empty((NL + 1, S, U), dtype=torch.float32, device=st.device)
        _abi.particle_net_weight_grads(dz, stash, pw, pb, NL + 1, R, S)
        dW, db = pw.sum(1), pb.sum(1)                    # (NL + 1, 64, 64), (NL + 1, 64)
        quad = kind == _abi.KIND_JACOBIAN
        n_out = d_out.shape[1]
        if quad:
            # tangent rows carry no bias: bias gradients come from the primal row of each group of four
            # (N rows in all: plain torch reductions)
            db = dz[:, 0::4, :].sum(1)
            g_first = torch.stack([(dz[NL] * st[:, i:i + 1]).sum(0) for i in range(d)], dim=1)
            g_head = torch.stack([(stash[NL] * d_out[:, o:o + 1]).sum(0) for o in range(n_out)], dim=0)
            g_head_b = d_out[0::4].sum(0)
            d_traj_bias = dz[2][0::4]
        else:
            # the narrow reductions -- (64 x R) @ (R x d), (n_out x R) @ (R x 64), column sums of d_out and
            # the per-trajectory sums of the join layer's dz -- in one pass over the rows
            # (mmf_particle_net_small_grads; as GEMMs they are the library's worst shapes, as torch
            # reductions five passes over (R, 64) tensors)
            SL = max(1, min(16, M // 256))
            pf = torch.empty((N * SL, U, 4), dtype=torch.float32, device=st.device)
            ph = torch.empty((N * SL, 4, U), dtype=torch.float32, device=st.device)
            pd = torch.empty((N * SL, 4), dtype=torch.float32, device=st.device)
            pt = torch.empty((N * SL, U), dtype=torch.float32, device=st.device)
            _abi.particle_net_small_grads(dz[NL], dz[2], stash[NL], st, d_out, pf, ph, pd, pt, N, M, SL)
            g_first = pf.sum(0)[:, :d]
            g_head = ph.sum(0)[:n_out]
            g_head_b = pd.sum(0)[:n_out]
            d_traj_bias = pt.view(N, SL, U).sum(1)
        grads[0] = g_first                               # first layer (64, d)
        grads[1] = db[NL]
        grads[2], grads[3], grads[4], grads[5] = dW[0], db[0], dW[1], db[1]  # encoder residual block
        gj = torch.zeros_like(params[6], dtype=torch.float32)  # join: only the state columns are ours
        off = net.join_state_off
        gj[:, off:off + U] = dW[2]
        grads[6] = gj
        for i in range(net.n_res):
            for k in range(2):
                layer = 3 + 2 * i + k
                grads[7 + 4 * i + 2 * k] = dW[layer]
                grads[8 + 4 * i + 2 * k] = db[layer]
        grads[-2] = g_head
        grads[-1] = g_head_b
        return (None, None, None, None, d_states, d_traj_bias, *grads)


# Rows of one backward chunk of the native training recursion (whole trajectories): bounds the two
# recompute buffers (stash + dz: 2 x (NL + 1) x rows x 256 B = 1.3 GB at the default) whatever N, M, T.
# Measured at 32 x 8192 x 16 (push unimodal PF, ms per optimiser step): 16,384 rows 113, 32,768 (both
# buffers inside the 256 MiB Infinity Cache) 65.7, 65,536 56.1, 262,144 45.0 -- on-die hand-offs do not pay
# for the small launches (150 KB of weights staged per workgroup for 4 tiles, half-empty grids)
TRAIN_CHUNK_ROWS = int(os.environ.get("MMF_TRAIN_CHUNK_ROWS", "262144"))
# The backward of the native training recursion, per network call of a step (round 6: two forms, not six):
#  * fused (default when the engine's mode is f16x3 and the networks have the reference's depths): recompute in the forward
#    pass's own three-product f16 arithmetic + backward data path + weight gradients as ONE kernel
#    (csrc/particle_net_fused.hip) -- layer inputs and pre-activation gradients never reach HBM, the weight gradients
#    accumulate in registers across the launch (one partial per workgroup).  Numerics: the data path (d_states, every
#    gradient that flows to earlier steps and to the per-trajectory networks) keeps 22 bits per element down to 2^-14 of its
#    own particle's largest gradient (measured against the exact-fp32 backward: 1.6e-6 overall, <= 5.5e-5 in the worst row).
#  * three passes with EXACT fp32 products over f16 recompute buffers (activations as f16, pre-activation gradients as f16
#    relative to the largest magnitude of their 32-row tile: only the PARAMETER gradients see that rounding): the training
#    path of the f32 engine mode (the reference trains in fp32: MMF_PRECISION=f32) and the cross-check of the fused kernel
#    (MMF_TRAIN_FUSED=0, tests/test_gpu_training.py).
# Rounds 3-4's other forms (fp32 buffers; f16x3 recompute / backward in three passes) are deleted; their timings are in
# profiles/r05/bench_train_fused_ab.txt.
TRAIN_FUSED = os.environ.get("MMF_TRAIN_FUSED", "1") != "0"
TRAIN_FUSED_SLOTS = 256  # weight-gradient partials per layer = the largest grid of the fused kernel
# the measurement networks of a step share one launch (MmfPfTrainArgs.fused_sets); False: one launch each (what a filter with
# ONE measurement network runs anyway; tests compare the two bit for bit)
TRAIN_FUSED_MERGE = True


class PfTrainLoopFunction(torch.autograd.Function):
    """K6, whole recursion: ``T`` train-mode (no resampling) particle-filter steps forward in ONE C call
    (``mmf_pf_train_forward``) and backward in one (``mmf_pf_train_backward``: activations recomputed per
    chunk of trajectories into a cache-resident stash, weight gradients accumulated on the device).

    ``apply(nets, T, N, M, states0 (N, M, d), logw0 (N, M), eps (T, N, M, d), tril (d, d), dyn_bias (T*N, 64),
    beta (T*N, K_all) | empty, *meas_biases (T*N, 64), *params)`` -> ``(estimates (T, N, d), states_T, logw_T)``
    with ``nets = (dyn_net, [(meas_net, beta column | None)], K_all)`` and ``params`` the concatenated
    ``_sources()`` of the dynamics network and of every measurement network."""

    @staticmethod
    def forward(ctx, nets, T, N, M, states0, logw0, eps, tril, dyn_bias, beta, *rest):
        dyn_net, meas, K_all = nets
        K = len(meas)
        meas_biases, params = rest[:K], rest[K:]
        dev = states0.device
        require_device(states0, "PfTrainLoopFunction")
        d = states0.shape[-1]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        U = _abi.MMF_UNITS
        R = N * M
        states = torch.empty((T + 1, N, M, d), dtype=torch.float32, device=dev)
        logw = torch.empty((T + 1, N, M), dtype=torch.float32, device=dev)
        states[0].copy_(states0.detach())
        logw[0].copy_(logw0.detach())
        est = torch.empty((T, N, d), dtype=torch.float32, device=dev)
        keep = dict(states=states, logw=logw, est=est, eps=f32(eps), tril=f32(tril), dyn_bias=f32(dyn_bias),
                    meas_biases=[f32(b) for b in meas_biases], beta=f32(beta) if beta.numel() else None,
                    loglik=torch.empty((N, M), dtype=torch.float32, device=dev),
                    ll_steps=torch.empty((T, K, N, M), dtype=torch.float32, device=dev))
        a = _abi.MmfPfTrainArgs()
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        a.T, a.N, a.M, a.d, a.n_meas = T, N, M, d, K
        a.n_res_dyn, a.n_res_meas, a.logw_stride = dyn_net.n_res, meas[0][0].n_res, K_all
        # forward pass in the engine's arithmetic mode (f16x3 by default: the inference kernels); the backward
        # recomputes the SAME particle sets' activations with exact fp32 products
        a.precision = dyn_net.precision_code()
        blobs = [dyn_net.blob(_abi.PREC_F32)] + [m.blob(_abi.PREC_F32) for m, _ in meas]
        fwd_blobs = [dyn_net.blob()] + [m.blob() for m, _ in meas]
        a.dyn.packed, a.dyn.packed_f32 = P(fwd_blobs[0]), P(blobs[0])
        for k, (m, col) in enumerate(meas):
            a.meas[k].packed, a.meas[k].packed_f32 = P(fwd_blobs[1 + k]), P(blobs[1 + k])
            a.meas_bias[k] = P(keep["meas_biases"][k])
            if keep["beta"] is not None and col is not None:
                a.meas_logw[k] = ctypes.c_void_p(keep["beta"].data_ptr() + 4 * col)
        a.dyn_bias, a.noise, a.scale_tril = P(keep["dyn_bias"]), P(keep["eps"]), P(keep["tril"])
        a.states, a.logw, a.estimates = P(states), P(logw), P(est)
        a.loglik, a.ll_steps = P(keep["loglik"]), P(keep["ll_steps"])
        a.range_flag = ctypes.c_void_p(range_flag(dev).data_ptr())
        _abi.pf_train_forward(a, states)
        ctx.nets, ctx.shape, ctx.keep, ctx.blobs = nets, (T, N, M, d), keep, blobs
        ctx.fwd_blobs, ctx.fwd_precision = fwd_blobs, int(a.precision)
        ctx.save_for_backward(*[p.detach() for p in params])
        sT, lT = states[T], logw[T]
        ctx.mark_non_differentiable(sT, lT)
        return est, sT, lT

    @staticmethod
    def backward(ctx, g_est, _g_states, _g_logw):
        dyn_net, meas, K_all = ctx.nets
        T, N, M, d = ctx.shape
        keep = ctx.keep
        params = ctx.saved_tensors
        K = len(meas)
        dev = keep["states"].device
        U = _abi.MMF_UNITS
        chunk_traj = max(1, min(N, TRAIN_CHUNK_ROWS // M))
        C = chunk_traj * M
        S = max(1, min(128, C // 64))  # weight-gradient slices: >= 64 rows each (960 rows -> 15 workgroups per layer)
        SL = max(1, min(16, M // 256))
        nets = [dyn_net] + [m for m, _ in meas]
        NLmax = max(3 + 2 * n.n_res for n in nets)
        E = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        P = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        a = _abi.MmfPfTrainArgs()
        a.T, a.N, a.M, a.d, a.n_meas = T, N, M, d, K
        a.n_res_dyn, a.n_res_meas, a.logw_stride = dyn_net.n_res, meas[0][0].n_res, K_all
        a.chunk_traj, a.n_splits, a.n_slices = chunk_traj, S, SL
        # fused: recompute in the forward pass's own arithmetic (f16x3 blobs); otherwise three passes of exact fp32 products
        fused = bool(ctx.fwd_precision == _abi.PREC_F16X3 and TRAIN_FUSED and all(n.n_res == (3 if i == 0 else 2) for i, n in enumerate(nets)))
        a.fused = int(fused)
        a.precision = _abi.PREC_F16X3 if fused else _abi.PREC_F32
        if fused:
            S = min(TRAIN_FUSED_SLOTS, max(1, -(-C // 128)))  # one partial per workgroup (4 tiles of 32 rows per pass)
            a.n_splits = S
        sets = 1
        bufs = []
        head_ws, tblobs = [], []
        n_par = [len(n._sources()) for n in nets]
        offs = [0]
        for c in n_par:
            offs.append(offs[-1] + c)
        for i, net in enumerate(nets):
            NL = 3 + 2 * net.n_res
            tn = a.dyn if i == 0 else a.meas[i - 1]
            Z = (lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)) if fused else E  # fused: accumulated in place
            b = dict(pw=Z(NL + 1, S, U, U), pb=Z(NL + 1, S, U), p_first=E(T, N * SL, U, 4), p_head=E(T, N * SL, 4, U),
                     p_dout=E(T, N * SL, 4), p_traj=E(T, N * SL, U))
            bufs.append(b)
            head_ws.append(params[offs[i + 1] - 2].to(torch.float32).contiguous())
            tblobs.append(None if fused else _transposed_blob(net, _abi.PREC_F32))
            if fused:
                tn.packed_dual = P(net.blob(_abi.PREC_F16X3_DUAL))
            tn.packed_f32 = P(ctx.blobs[i])
            tn.packed = P(ctx.fwd_blobs[i]) if fused else tn.packed_f32
            tn.packed_t, tn.head_w = P(tblobs[i]), P(head_ws[i])
            tn.pw, tn.pb, tn.p_first, tn.p_head, tn.p_dout, tn.p_traj = (P(b[k]) for k in ("pw", "pb", "p_first", "p_head", "p_dout", "p_traj"))
        for k in range(K):
            a.meas_bias[k] = P(keep["meas_biases"][k])
            col = meas[k][1]
            if keep["beta"] is not None and col is not None:
                a.meas_logw[k] = ctypes.c_void_p(keep["beta"].data_ptr() + 4 * col)
        g_est = g_est.to(torch.float32).contiguous()
        A = lambda *shape: torch.empty(shape, dtype=torch.float16, device=dev)   # the recompute buffers are f16 (ABI 39)
        if fused:  # only the three (C, 64) row slots of the narrow reductions + the dynamics' encoder hand-offs
            fs = max(1, K)  # one set of row slots per measurement network: a step's networks share a launch
            scratch = dict(stash=A(fs, C, U), mask=None, dz=A(fs, 2, C, U), dz_scale=E(fs, 2, C), raw=None, d_raw=E(K + 8, C), ga=E(N, M, d), gb=E(N, M, d),
                           la=E(N, M), lb=E(N, M), d_tmp=E(fs, C, d), d_states0=E(N, M, d), d_logw0=E(N, M), act=E(C, U), g_act=E(C, U))
            a.fused_act, a.fused_g_act, a.fused_sets = P(scratch["act"]), P(scratch["g_act"]), (fs if TRAIN_FUSED_MERGE else 1)
        else:
            scratch = dict(stash=A(sets, NLmax + 1, C, U), mask=torch.empty((sets, NLmax + 1, C, 2), dtype=torch.int32, device=dev),
                           dz=A(sets, NLmax + 1, C, U), dz_scale=E(sets, NLmax + 1, C), raw=E(sets, C, 8), d_raw=E(K + 8, C), ga=E(N, M, d), gb=E(N, M, d),
                           la=E(N, M), lb=E(N, M), d_tmp=E(sets, C, d), d_states0=E(N, M, d), d_logw0=E(N, M))
        a.dyn_bias, a.noise, a.scale_tril, a.g_estimates = P(keep["dyn_bias"]), P(keep["eps"]), P(keep["tril"]), P(g_est)
        a.states, a.logw, a.estimates = P(keep["states"]), P(keep["logw"]), P(keep["est"])
        a.loglik, a.ll_steps = P(keep["loglik"]), P(keep["ll_steps"])
        a.stash, a.mask, a.dz, a.raw, a.d_raw, a.d_tmp = (P(scratch[k]) for k in ("stash", "mask", "dz", "raw", "d_raw", "d_tmp"))
        a.dz_scale = P(scratch["dz_scale"])
        a.g_states_a, a.g_states_b, a.g_logw_a, a.g_logw_b = P(scratch["ga"]), P(scratch["gb"]), P(scratch["la"]), P(scratch["lb"])
        a.d_states0, a.d_logw0 = P(scratch["d_states0"]), P(scratch["d_logw0"])
        _abi.pf_train_backward(a, g_est)
        # the partial sums -> parameter gradients in _sources() order, d hoisted terms, d modality log-weights: two launches
        # per network (mmf_pf_train_finalize)
        param_grads, bias_grads, d_beta = [], [], None
        if keep["beta"] is not None:
            d_beta = torch.zeros((T * N, K_all), dtype=torch.float32, device=dev)
        for i, net in enumerate(nets):
            b = bufs[i]
            own = params[offs[i]:offs[i + 1]]
            flat = torch.empty(sum(p.numel() for p in own), dtype=torch.float32, device=dev)
            # mmf_pf_train_finalize writes the gradients of (first layer, encoder block, join, residual blocks, head) in
            # _sources() order, derived from (d, n_res, join_in, n_out) alone: hold the parameters' layout to that count
            assert flat.numel() == U * d + U + (2 + 2 * net.n_res) * (U * U + U) + U * net.join.in_features + net.n_out * U + net.n_out, \
                "a per-particle network whose _sources() layout differs from (d, n_res, join_in, n_out) cannot use mmf_pf_train_finalize"
            bias_grads.append(E(T * N, U))
            fa = _abi.MmfPfTrainFinalizeArgs()
            fa.T, fa.N, fa.SL, fa.S, fa.n_res, fa.d, fa.n_out = T, N, SL, int(a.n_splits), net.n_res, d, net.n_out
            fa.join_in, fa.join_state_off, fa.fused = net.join.in_features, net.join_state_off, int(fused)
            col = meas[i - 1][1] if i > 0 else None
            fa.beta_stride, fa.beta_col = K_all, (col if col is not None else 0)
            fa.pw, fa.pb, fa.p_first, fa.p_head, fa.p_dout, fa.p_traj = (P(b[k]) for k in ("pw", "pb", "p_first", "p_head", "p_dout", "p_traj"))
            fin_scratch = E(32 * 516)
            fa.grads, fa.bias_grad, fa.scratch = P(flat), P(bias_grads[-1]), P(fin_scratch)
            fa.d_beta = P(d_beta) if (d_beta is not None and col is not None) else None
            _abi.pf_train_finalize(fa, flat)
            o = 0
            for p_ in own:
                param_grads.append(flat[o:o + p_.numel()].view(p_.shape))
                o += p_.numel()
        none7 = (None, None, None, None)
        return (*none7, scratch["d_states0"], scratch["d_logw0"], None, None, bias_grads[0],
                d_beta if d_beta is not None else None, *bias_grads[1:], *param_grads)


def dynamics_with_jacobian_autograd(net: PackedParticleNet, x: torch.Tensor, traj_bias: torch.Tensor):
    """K6 for K5: differentiable ``(x' (N, d), A (N, d, d))`` of a dynamics network
    ``x' = x + dir(x) * sigmoid(gate(x))``.  The primal and the ``d`` tangent columns ``e_j`` of every
    trajectory go through ``ParticleNetFunction`` as one group of four rows (tangents: no biases,
    the primal's ReLU masks), whose raw head outputs are ``(dir, gate)`` and the columns of their
    Jacobians; the sigmoid-gate algebra on top is a few element-wise torch ops on ``N`` rows.
    Gradients reach the weights and ``traj_bias`` through both ``x'`` and ``A``, ``x`` through ``x'``."""
    N, d = x.shape
    assert d <= 3, "groups are {primal, e_1, e_2, e_3}"
    # x' keeps its dependence on x ...
    out = ParticleNetFunction.apply(net, _abi.KIND_DYNAMICS, N, 1, x, traj_bias, *net._sources())
    x_next = x + out[:, :d] * torch.sigmoid(out[:, d:])
    # ... the Jacobian is evaluated AT x, as upstream's default ``DynamicsModel.jacobian`` does (it
    # differentiates a detached copy of the states: gradients reach the weights and the controls
    # through A, not the belief mean)
    eye = torch.zeros((N, 3, d), dtype=x.dtype, device=x.device)
    eye[:, :d, :] = torch.eye(d, dtype=x.dtype, device=x.device)
    rows = torch.cat([x.detach()[:, None, :], eye], dim=1).reshape(4 * N, d)
    raw = ParticleNetFunction.apply(net, _abi.KIND_JACOBIAN, N, 4, rows, traj_bias, *net._sources()).view(N, 4, d + 1)
    dirp, gate = raw[:, 0, :d], raw[:, 0, d:]
    sg = torch.sigmoid(gate)
    J_dir = raw[:, 1:1 + d, :d].transpose(1, 2)          # [n, i, j] = d dir_i / d x_j
    J_gate = raw[:, 1:1 + d, d]                          # [n, j]    = d gate / d x_j
    A = torch.eye(d, dtype=x.dtype, device=x.device) + sg[:, :, None] * J_dir \
        + (dirp * sg * (1.0 - sg))[:, :, None] * J_gate[:, None, :]
    return x_next, A


class ReweightEstimateFunction(torch.autograd.Function):
    """K6 (K1, no-resample path): ``(loglik, logw_in, states) -> (estimate, logw_out)`` with
    ``logw_out = logw_in + loglik - logsumexp`` and ``estimate = sum_m exp(logw_out) x_m``;
    forward = K1 mode 0, backward = ``mmf_pf_reweight_backward``."""

    @staticmethod
    def forward(ctx, loglik, logw_in, states):
        require_device(states, "ReweightEstimateFunction")
        N, M, d = states.shape
        st = states.detach().to(torch.float32).contiguous()
        est = torch.empty((N, d), dtype=torch.float32, device=states.device)
        logw_out = torch.empty((N, M), dtype=torch.float32, device=states.device)
        _abi.pf_reweight_resample(loglik.detach().to(torch.float32).contiguous(),
                                  logw_in.detach().to(torch.float32).contiguous(), st, None, est, None,
                                  logw_out, None, 0)
        ctx.save_for_backward(logw_out, st)
        return est, logw_out

    @staticmethod
    def backward(ctx, g_est, g_logw):
        logw_out, st = ctx.saved_tensors
        d_a = torch.empty_like(logw_out)
        d_states = torch.empty_like(st)
        g_est = torch.zeros_like(st[:, 0, :]) if g_est is None else g_est.to(torch.float32).contiguous()
        g_logw = None if g_logw is None else g_logw.to(torch.float32).contiguous()
        _abi.pf_reweight_backward(logw_out, st, g_est, g_logw, d_a, d_states)
        return d_a, d_a, d_states


class EkfStepFunction(torch.autograd.Function):
    """K6 (K3): one sub-filter's predict + correct, ``(A (N,d,d), mu_pred (N,d), q_tril (d,d), z (N,d),
    r_tril (N,d,d), Sigma (N,d,d)) -> (mu (N,d), Sigma' (N,d,d))``; forward = ``mmf_ekf_step`` (fusion
    0), backward = ``mmf_ekf_step_backward`` (closed-form adjoints, one trajectory per lane).  The
    dynamics noise ``q_tril`` is a fixed parameter of the reference's models: no gradient."""

    @staticmethod
    def forward(ctx, A, mu_pred, q_tril, z, r_tril, Sigma):
        require_device(mu_pred, "EkfStepFunction")
        N, d = mu_pred.shape
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        A_, mp_, q_, z_, r_, S_ = f32(A).view(1, N, d, d), f32(mu_pred).view(1, N, d), f32(q_tril).view(1, d, d), \
            f32(z).view(1, N, d), f32(r_tril).view(1, N, d, d), f32(Sigma).view(1, N, d, d)
        mu = torch.empty((1, N, d), dtype=torch.float32, device=mu_pred.device)
        S_out = S_.clone()
        _abi.ekf_step(A_, mp_, q_, z_, r_, None, mu, S_out, None, None, fusion=0, feedback=0)
        ctx.save_for_backward(A_, mp_, q_, z_, r_, S_)
        return mu[0], S_out[0]

    @staticmethod
    def backward(ctx, g_mu, g_Sigma):
        A_, mp_, q_, z_, r_, S_ = ctx.saved_tensors
        _, N, d = mp_.shape
        c = lambda g, shape: None if g is None else g.to(torch.float32).contiguous().view(shape)
        g_A, g_r, g_S = torch.empty_like(A_), torch.empty_like(r_), torch.empty_like(S_)
        g_mp, g_z = torch.empty_like(mp_), torch.empty_like(z_)
        _abi.ekf_step_backward(A_, mp_, q_, z_, r_, S_, c(g_mu, (1, N, d)), c(g_Sigma, (1, N, d, d)),
                               g_A, g_mp, g_z, g_r, g_S)
        return g_A[0], g_mp[0], None, g_z[0], g_r[0], g_S[0]


def run_dynamics(net: PackedParticleNet, states: torch.Tensor, traj_bias: torch.Tensor,
                 noise, scale_tril, out: torch.Tensor = None) -> torch.Tensor:
    """``states`` ``(N, M, d)`` or ``(R, d)`` with ``traj_bias`` ``(N, 64)``."""
    require_device(states, "run_dynamics")
    d = states.shape[-1]
    N = traj_bias.shape[0]
    R = states.numel() // d
    assert R % N == 0
    states = states.contiguous()
    out = torch.empty_like(states) if out is None else out
    blob = net.blob()
    noise_c = None if noise is None else noise.contiguous()
    tril_c = None if noise is None else scale_tril.contiguous()
    _timed("particle_net_dynamics", 2.0 * R * particle_net_macs(d, net.n_res, net.n_out),
           R * 4.0 * (2 * d + (d if noise is not None else 0)),
           lambda: _abi.pf_dynamics(blob, net.n_res, net.precision_code(), states, traj_bias, noise_c,
                                    tril_c, out, range_flag(states.device), N, R // N, d))
    return out


def run_measure(net: PackedParticleNet, states: torch.Tensor, traj_bias: torch.Tensor,
                modality_logw, logw_stride: int, loglik: torch.Tensor, combine: bool):
    require_device(states, "run_measure")
    d = states.shape[-1]
    N = traj_bias.shape[0]
    R = states.numel() // d
    assert R % N == 0 and loglik.numel() == R
    blob = net.blob()
    states = states.contiguous()
    _timed("particle_net_measure", 2.0 * R * particle_net_macs(d, net.n_res, net.n_out),
           R * 4.0 * (d + 1 + (1 if combine else 0)),
           lambda: _abi.pf_measure(blob, net.n_res, net.precision_code(), states, traj_bias,
                                   modality_logw, logw_stride, loglik, combine,
                                   range_flag(states.device), N, R // N, d))
    return loglik


def run_jacobian(net: PackedParticleNet, states: torch.Tensor, traj_bias: torch.Tensor):
    """``(N, d)`` -> ``(x' (N, d), J (N, d, d))`` by forward-mode tangents (K5)."""
    require_device(states, "run_jacobian")
    N, d = states.shape
    out = torch.empty_like(states)
    jac = torch.empty((N, d, d), dtype=torch.float32, device=states.device)
    prec = net.precision_code()
    flag = range_flag(states.device) if prec != _abi.PREC_F32 else None
    _abi.dynamics_jacobian(net.blob(prec), net.n_res, prec, states.contiguous(), traj_bias, out, jac, flag, N, d)
    return out, jac


# ------------------------------------------------------------------------------ K4 image encoders
def _image_encoder_variant(seq):
    """``_abi.ENCODER_*`` of an ``observation_image_layers`` stack K4 implements, else ``None``."""
    from .layers import DualSpanningAvgPool, ResConv2d

    try:
        if not (len(seq) == 10 and isinstance(seq[0], nn.Conv2d) and seq[0].weight.shape == (32, 1, 5, 5)
                and isinstance(seq[2], ResConv2d) and seq[3].weight.shape == (16, 32, 3, 3)
                and isinstance(seq[5], nn.Conv2d) and isinstance(seq[7], nn.Linear)
                and isinstance(seq[9], ResLinear)):
            return None
        if seq[5].weight.shape == (8, 16, 3, 3) and seq[7].weight.shape == (64, 8192):
            return _abi.ENCODER_DEFAULT
        if (seq[5].weight.shape == (2, 16, 3, 3) and isinstance(seq[6], DualSpanningAvgPool)
                and seq[7].weight.shape == (64, 64)):
            return _abi.ENCODER_SPANNING_POOL
    except (TypeError, AttributeError, IndexError):
        pass
    return None


def _is_default_image_encoder(seq) -> bool:
    return _image_encoder_variant(seq) is not None


class PackedImageEncoder:
    """Fragment-ordered device copy of one default ``observation_image_layers`` stack
    (``layers.image_encoder``), rebuilt lazily when a source parameter changes."""

    def __init__(self, seq: nn.Sequential):
        self.variant = _image_encoder_variant(seq)
        assert self.variant is not None
        self.seq = seq
        self._blob = None
        self._stamp = None

    def _sources(self):
        q = self.seq
        return [q[0].weight, q[2].block1.weight, q[2].block2.weight, q[3].weight, q[5].weight,
                q[0].bias, q[2].block1.bias, q[2].block2.bias, q[3].bias, q[5].bias,
                q[7].weight, q[7].bias, q[9].block1.weight, q[9].block2.weight,
                q[9].block1.bias, q[9].block2.bias]

    def blob(self) -> torch.Tensor:
        src = self._sources()
        stamp = tuple((t.data_ptr(), t._version, str(t.device)) for t in src)
        if self._blob is not None and stamp == self._stamp:
            return self._blob
        require_device(src[0], "PackedImageEncoder")
        keep = [t.detach().to(torch.float32).contiguous() for t in src]
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        d = _abi.MmfImageEncoderDesc()
        for i in range(5):
            d.conv_w[i], d.conv_b[i] = P(keep[i]), P(keep[5 + i])
        d.fc_w, d.fc_b = P(keep[10]), P(keep[11])
        d.res_w[0], d.res_w[1], d.res_b[0], d.res_b[1] = P(keep[12]), P(keep[13]), P(keep[14]), P(keep[15])
        d.variant = self.variant
        blob = torch.empty(_abi.image_encoder_floats(), dtype=torch.float32, device=src[0].device)
        _abi.pack_image_encoder(d, blob)
        self._blob, self._stamp = blob, stamp
        return blob


class ImageConvsFunction(torch.autograd.Function):
    """K6 for the image encoder: the convolution stack of a default ``observation_image_layers``
    (``door_models/layers.py:43-58``: stem, ResConv, 32->16, 16->8) forward with every activation kept
    (``mmf_image_convs_train_forward``: exact fp32 products, or -- ``set_image_encoder_precision("bf16")``,
    BASELINE config 5 -- bf16 products in the two 32->32 convolutions), backward data path on transposed + flipped weights with the
    ReLU masks fused (``mmf_image_convs_train_backward``) and the 3x3 weight gradients as split-K MFMA
    correlations (``mmf_conv_weight_grads``): no MIOpen kernel in a training step.  Exact fp32.
    ``apply(seq, images (N, 32, 32), *conv weights and biases) -> (N, 8, 32, 32)``; the flatten + linear
    + ResLinear behind it stay torch modules (library GEMMs); bias gradients come out of the same kernels."""

    @staticmethod
    def forward(ctx, seq, images, *params):
        require_device(images, "ImageConvsFunction")
        if not hasattr(seq, "_mmf_packed"):
            object.__setattr__(seq, "_mmf_packed", PackedImageEncoder(seq))
        blob = seq._mmf_packed.blob()
        img = images.detach().to(torch.float32).contiguous()
        N = img.shape[0]
        mk = lambda c: torch.empty((N, c, 32, 32), dtype=torch.float32, device=img.device)
        a1, h, a2, a3, a4 = mk(32), mk(32), mk(32), mk(16), mk(8)
        prec = training_image_precision_code()
        _abi.image_convs_train_forward(blob, img, a1, h, a2, a3, a4,
                                       range_flag(img.device) if prec != _abi.PREC_F32 else None, prec)
        ctx.seq = seq
        ctx.save_for_backward(img, a1, h, a2, a3)
        return a4

    @staticmethod
    def backward(ctx, g_a4):
        seq = ctx.seq
        img, a1, h, a2, a3 = ctx.saved_tensors
        N = img.shape[0]
        src = seq._mmf_packed._sources()
        stamp = tuple((t.data_ptr(), t._version) for t in src[1:5])
        cached = getattr(seq, "_mmf_packed_bwd", None)
        if cached is None or cached[0] != stamp:
            keep = [t.detach().to(torch.float32).contiguous() for t in src[:5]]
            d = _abi.MmfImageEncoderDesc()
            for i in range(5):
                d.conv_w[i] = ctypes.c_void_p(keep[i].data_ptr())
            d.variant = _abi.ENCODER_DEFAULT
            blob_b = torch.empty(_abi.image_convs_backward_floats(), dtype=torch.float32, device=img.device)
            _abi.pack_image_convs_backward(d, blob_b)
            cached = (stamp, blob_b)
            object.__setattr__(seq, "_mmf_packed_bwd", cached)
        g_a4 = g_a4.to(torch.float32).contiguous()
        g1, gh, g2 = torch.empty_like(a1), torch.empty_like(a1), torch.empty_like(a1)
        g3 = torch.empty_like(a3)
        gmax = None   # f16x3 backward: max |g3|, |g2|, |gh|, |g_a4| as device scalars (the operand splits' scales)
        if training_image_precision_code() != _abi.PREC_F32:   # round 6: the 32-output-channel layers on the f16 matrix pipe
            gmax = torch.empty(4, dtype=torch.float32, device=img.device)
            _abi.image_convs_train_backward_h(cached[1], a1, h, a2, a3, g_a4, g1, gh, g2, g3, gmax)
        else:
            _abi.image_convs_train_backward(cached[1], a1, h, a2, a3, g_a4, g1, gh, g2, g3)
        blocks = max(1, min(2 * N, 256))  # a workgroup walks half-images; one partial per workgroup
        partial = torch.empty((blocks, 9, 32, 32), dtype=torch.float32, device=img.device)
        partial_b = torch.empty((blocks, 32), dtype=torch.float32, device=img.device)
        E = lambda *shape: torch.empty(shape, dtype=torch.float32, device=img.device)

        # round 6: with a reduced precision selected for the training forward (f16x3 / bf16: training_image_precision_code) the
        # 3x3 layers' weight gradients run on the f16 matrix pipe with three products per product (mmf_conv_weight_grads_h:
        # 5x fewer matrix-pipe cycles than the exact-fp32 kernel); the 5x5 stem (25 taps of one channel) stays exact
        reduced = training_image_precision_code() != _abi.PREC_F32

        def wgrad(g, act, k=3, slot=None):  # partial slots, then their sum in nn.Conv2d's layout: two launches per layer
            co, ci = g.shape[1], act.shape[1]
            dw, db = E(co, ci, k, k), E(co)
            if reduced and k == 3:
                _abi.conv_weight_grads_h(g, act, gmax[slot:slot + 1], partial, partial_b, range_flag(g.device), blocks, dw, db)
            else:
                _abi.conv_weight_grads(g, act, partial, partial_b, blocks, dw, db)
            return dw, db

        (gw4, gb4), (gw3, gb3), (gw2b, gb2b), (gw2a, gb2a) = wgrad(g_a4, a3, slot=3), wgrad(g3, a2, slot=0), wgrad(g2, h, slot=1), wgrad(gh, a1, slot=2)
        gw1, gb1 = wgrad(g1, img[:, None], 5)                                            # the 5x5 stem
        return (None, None, gw1, gw2a, gw2b, gw3, gw4, gb1, gb2a, gb2b, gb3, gb4)


class Fc64Function(torch.autograd.Function):
    """K6 for the image encoder's ``nn.Linear(8 * 32 * 32, 64)`` (``door_models/layers.py:59-60``): ``x w^T + b``
    forward, ``g w`` / ``g^T x`` / column sums backward, exact fp32 on the matrix cores in a fixed order
    (``mmf_fc64_train_forward`` / ``_backward``, ``csrc/traj_train.hip``) -- no library GEMM."""

    @staticmethod
    def forward(ctx, x, w, b):
        require_device(x, "Fc64Function")
        x = x.detach().to(torch.float32).contiguous()
        y = torch.empty((x.shape[0], 64), dtype=torch.float32, device=x.device)
        _abi.fc64_train_forward(x, w.detach().contiguous(), b.detach().contiguous(), y)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, db = torch.empty_like(w), torch.empty(64, dtype=torch.float32, device=x.device)
        _abi.fc64_train_backward(g.to(torch.float32).contiguous(), x, w.detach().contiguous(), dx, dw, db)
        return dx, dw, db


# round 5: the per-trajectory networks of a training step (vector encoders, weight model, hoisted join halves, the linear
# tail of the image encoder) run forward AND backward in HIP (trajprog.TrajProgram.run_autograd, Fc64Function) under the
# "hip" training backend.  The torch modules + autograd (library GEMMs) ARE the "autograd" backend; this flag lets
# tests/test_gpu_training.py run them under the "hip" backend too, as the cross-check (no environment switch any more).
TRAIN_TRAJ_PROGRAMS = True


def use_traj_program_backward(*tensors) -> bool:
    return TRAINING_BACKEND == "hip" and TRAIN_TRAJ_PROGRAMS and all(t.is_cuda for t in tensors)


def image_features_autograd(seq, images: torch.Tensor, pre_activation: bool = False) -> torch.Tensor:
    """Differentiable ``observation_image_layers(images[:, None])``: with the "hip" training backend
    and a default stack the convolutions and the 8192 -> 64 linear layer run forward and backward in HIP
    (``ImageConvsFunction``, ``Fc64Function``); otherwise the torch module as it stands.  ``pre_activation``: stop behind
    the linear layer (the caller's K7 program applies the ReLU and the ResLinear; default stacks on the device only)."""
    if use_hip_backward() and images.is_cuda and _image_encoder_variant(seq) == _abi.ENCODER_DEFAULT:
        params = PackedImageEncoder(seq)._sources()[:10] if not hasattr(seq, "_mmf_packed") else seq._mmf_packed._sources()[:10]
        a4 = ImageConvsFunction.apply(seq, images, *params)
        x = a4.flatten(1)
        if TRAIN_TRAJ_PROGRAMS and seq[7].weight.dtype == torch.float32 and seq[7].out_features == 64:
            x = Fc64Function.apply(x, seq[7].weight, seq[7].bias)
            if pre_activation:
                return x
            for layer in list(seq)[8:]:
                x = layer(x)
            return x
        assert not pre_activation
        for layer in list(seq)[7:]:
            x = layer(x)
        return x
    assert not pre_activation
    return seq(images[:, None, :, :])


def image_tail_in_program(seq, images: torch.Tensor) -> bool:
    """Whether ``image_features_autograd(seq, images, pre_activation=True)`` is available for this encoder."""
    return (use_traj_program_backward(images) and _image_encoder_variant(seq) == _abi.ENCODER_DEFAULT
            and seq[7].weight.dtype == torch.float32 and seq[7].out_features == 64)


_IMAGE_WORKSPACES = {}
_MAX_NETS = 4
# images per K4 launch sequence (workspace ~1.6 GB per encoder).  Measured round 3 on the EKF step (1024 trajectories,
# three encoders) / the PF step: 512 images 0.482 / 0.612 ms, 1024 0.446 / 0.599, 2048 0.435 / 0.596, 4096 0.420 / 0.588,
# 8192 .. 32768 no further gain: the persistent kernels' fill / drain per launch is what a chunk amortises; keeping
# the intermediates inside the 256 MiB Infinity Cache (small chunks) buys nothing.
_IMAGE_CHUNK = int(os.environ.get("MMF_IMAGE_CHUNK", "4096"))


def _image_workspace(device, n_images: int, n_nets: int) -> torch.Tensor:
    # sized for a full chunk as soon as more than one step's worth of images shows up, so a
    # long forward_loop never re-allocates (hipMalloc of ~1 GB stalls the stream)
    if n_images > 256:
        n_images = max(n_images, _IMAGE_CHUNK)
    need = _abi.image_encoder_workspace_bytes(n_images, n_nets)
    key = str(device)
    ws = _IMAGE_WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=device)
        _IMAGE_WORKSPACES[key] = ws
    return ws


def image_encoder_flops(n_images: int) -> float:
    return 2.0 * 26_124_288 * n_images  # MAC per image of the default stack


def encode_images(encoders, images: torch.Tensor):
    """Run several image encoders on the same ``(N, 32, 32)`` batch: default stacks go
    (``door_models/layers.py:43-63``) and the push virtual sensor's spanning-pool stacks
    (``push_models/layers.py:77-90``) go through K4, one batched launch sequence per
    architecture; anything else runs its own torch module.  Returns one ``(N, 64)`` per encoder."""
    require_device(images, "encode_images")
    images = images.to(torch.float32).contiguous()
    N = images.shape[0]
    out = [None] * len(encoders)
    variants = [_image_encoder_variant(e) for e in encoders]
    for i, e in enumerate(encoders):
        if variants[i] is None:  # a user-defined architecture: its own torch module
            out[i] = e(images[:, None, :, :])
    groups = []  # encoders of one architecture share a launch sequence, _MAX_NETS at a time
    for v in (_abi.ENCODER_DEFAULT, _abi.ENCODER_SPANNING_POOL):
        same = [i for i, vi in enumerate(variants) if vi == v]
        groups += [(v, same[lo:lo + _MAX_NETS]) for lo in range(0, len(same), _MAX_NETS)]
    for variant, grp in groups:
        packs = []
        for i in grp:
            e = encoders[i]
            if not hasattr(e, "_mmf_packed"):
                object.__setattr__(e, "_mmf_packed", PackedImageEncoder(e))
            packs.append(e._mmf_packed.blob())
        pieces = []  # per chunk: (nets, n, 64), written by the launch sequence itself
        # bounded workspace: at most _IMAGE_CHUNK images per launch sequence, in EQUAL chunks (round 5: 5,120 images run
        # as 2 x 2,560, not 4,096 + 1,024 -- the short chunk's persistent grids spent a third of their launch filling
        # and draining, profiles/r04: 0.265 PF against 0.31), a multiple of 256 so every workgroup of the persistent
        # grids gets the same number of images
        n_chunks = -(-N // _IMAGE_CHUNK)
        per = max(1, min(_IMAGE_CHUNK, -(-(-(-N // n_chunks)) // 256) * 256) if N > 256 else N)  # N == 0: an empty loop
        for c0 in range(0, N, per):
            n = min(per, N - c0)
            chunk = images[c0:c0 + n]
            feat = torch.empty((len(grp), n, 64), dtype=torch.float32, device=images.device)
            ws = _image_workspace(images.device, n, len(grp))
            prec = image_encoder_precision_code()
            flag = range_flag(images.device)
            _timed("image_encoder", image_encoder_flops(n) * len(grp), 0.0,
                   lambda: _abi.image_encoder(packs, chunk, feat, ws, flag, prec, variant))
            pieces.append(feat)
        # round 6: no copy per chunk and network (round 5's `feats[k][c0:c0 + n] = feat[k]` was the bench's 1,500
        # `copyBuffer` launches, scripts/debug/find_copies.py): one chunk = views of what the kernels wrote, several
        # chunks = one concatenation per network
        for k, i in enumerate(grp):
            if not pieces:
                out[i] = torch.empty((0, 64), dtype=torch.float32, device=images.device)
            else:
                out[i] = pieces[0][k] if len(pieces) == 1 else torch.cat([p[k] for p in pieces], dim=0)
    return out


def encode_observation_images(models, observations):
    """Image features for every model in ``models`` that owns an image encoder (``None`` for
    the others, and for models that do not accept handed-over features): all default
    stacks of a step run as ONE batched K4 launch sequence."""
    idx = [i for i, m in enumerate(models)
           if m is not None and getattr(m, "accepts_image_feat", False)
           and "image" in getattr(m, "modalities", ())]
    out = [None] * len(models)
    if idx:
        with torch.no_grad():
            feats = encode_images([models[i].observation_image_layers for i in idx], observations["image"])
        for i, f in zip(idx, feats):
            out[i] = f
    return out


def call_with_image_feat(fn, image_feat, **kwargs):
    """Call ``fn(**kwargs)``, adding ``image_feat=`` when features were pre-computed."""
    if image_feat is None:
        return fn(**kwargs)
    return fn(image_feat=image_feat, **kwargs)
