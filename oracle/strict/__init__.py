"""Bit-exact CPU twin of the engine's exact-fp32 ("strict") arithmetic mode.  TEST INFRASTRUCTURE.

``mmf_strict.c`` restates, operation for operation, what the HIP kernels compute in
``MMF_PREC_F32`` mode (k-ordered ``fmaf`` chains, the shared deterministic transcendentals of
``include/mmf_detmath.h``, K1's reduction tree); this module compiles it with gcc, binds it
with ctypes, and walks the ORACLE's torch modules (``oracle/models.py`` -- same ``state_dict``
as the reference's classes) with it.  ``StrictParticleFilter`` is ``oracle.tf.filters.
ParticleFilter.forward`` (torchfilter's step order) on that arithmetic.

Chain of evidence: reference -> (golden vectors) -> torch oracle -> (<= 2e-6, tests/
test_strict_cpu.py) -> this twin -> (every bit, tests/test_gpu_strict.py) -> HIP engine.
"""
import ctypes
import math
import os
import subprocess
from ctypes import POINTER, c_float, c_int, c_int32, c_long, c_uint32, c_uint64, c_void_p

import numpy as np
import torch
import torch.nn as nn

from .. import resample as _rs

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "mmf_strict.c")
LIB = os.path.join(HERE, "_build", "libmmf_strict.so")
HEADERS = [os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", h) for h in ("mmf_detmath.h", "mmf_philox.h")]
# -ffp-contract=off: the only fused operations are the explicit fmaf calls.  -mavx2 -mfma (x86-64-v3: the build
# container and the GPU box's EPYC hosts have them) only make fmaf one instruction; without them libm's software
# fmaf gives the same bits, slower -- so a host without AVX2 / FMA (or a gcc without OpenMP) still builds the twin
CFLAGS = ["-O3", "-ffp-contract=off", "-shared", "-fPIC", "-std=c11"]
_FAST = ["-mavx2", "-mfma"]


def _host_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as fh:
            flags = next((l for l in fh if l.startswith("flags")), "")
        return " avx2" in flags and " fma" in flags
    except OSError:
        return False

_F = np.float32
_lib = None


def build(force: bool = False) -> str:
    stale = force or not os.path.exists(LIB) or any(
        os.path.getmtime(f) > os.path.getmtime(LIB) for f in [SRC] + HEADERS)
    if stale:
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        fast = _FAST if _host_has_fma() else []
        attempts = [[*CFLAGS, *fast, "-fopenmp"], [*CFLAGS, *fast], [*CFLAGS]]
        last = None
        for flags in attempts:
            try:
                subprocess.run(["gcc", *flags, "-o", LIB + ".tmp", SRC, "-lm"], check=True, capture_output=True)
                break
            except (subprocess.CalledProcessError, FileNotFoundError) as e:  # no libgomp / no gcc at all
                last = e
        else:
            raise RuntimeError(f"cannot build the strict twin (test infrastructure): {last}")
        os.replace(LIB + ".tmp", LIB)
    return LIB


def available() -> bool:
    """Can the checker be built / loaded on this host?  (``__graft_entry__`` treats a missing gcc as a skipped
    check of the strict mode, not as a failure of the framework.)"""
    try:
        build()
        return True
    except (RuntimeError, OSError):
        return False


class _Net(ctypes.Structure):
    _fields_ = [("d_in", c_int32), ("n_res", c_int32), ("relu_after_join", c_int32), ("n_out", c_int32),
                ("join_in", c_int32), ("join_state_off", c_int32),
                ("w_in", c_void_p), ("b_in", c_void_p), ("w_enc", c_void_p * 2), ("b_enc", c_void_p * 2),
                ("w_join", c_void_p), ("w_res", c_void_p * 6), ("b_res", c_void_p * 6), ("w_head", c_void_p)]


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        P = c_void_p
        for name in ("strict_det_exp_nonpos", "strict_det_log", "strict_det_sigmoid"):
            getattr(L, name).argtypes = [P, P, c_long]
        L.strict_det_logaddexp.argtypes = [P, P, P, c_long]
        L.strict_linear.argtypes = [P, c_long, c_int, P, c_int, c_int, P, P, c_int, c_int, c_float, P]
        L.strict_conv.argtypes = [P, P, P, P, c_long, c_int, c_int, c_int, c_int, P]
        L.strict_fc_tail.argtypes = [P, P, P, P, P, P, P, c_long, P]
        L.strict_particle_net.argtypes = [POINTER(_Net), P, P, c_long, c_int, P]
        L.strict_measure_epilogue.argtypes = [P, c_float, P, c_int, c_long, c_int, c_int, P]
        L.strict_dynamics_epilogue.argtypes = [P, P, P, P, P, c_long, c_int, P]
        L.strict_estimate.argtypes = [P, P, c_long, c_int, c_int, c_int, P]
        L.strict_philox_normals.argtypes = [c_uint64, c_uint32, c_uint32, c_long, c_int, c_int, P]
        L.strict_philox_uniforms.argtypes = [c_uint64, c_uint32, c_uint32, c_int, c_int, P]
        L.strict_philox_raw.argtypes = [c_uint32] * 6 + [P]
        L.strict_philox_raw.restype = None
        for f in ("strict_det_exp_nonpos", "strict_det_log", "strict_det_sigmoid", "strict_det_logaddexp",
                  "strict_linear", "strict_conv", "strict_fc_tail", "strict_particle_net",
                  "strict_measure_epilogue", "strict_dynamics_epilogue", "strict_estimate",
                  "strict_philox_normals", "strict_philox_uniforms"):
            getattr(L, f).restype = None
        _lib = L
    return _lib


def _a(x) -> np.ndarray:
    if torch.is_tensor(x):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=_F)


def _p(x):
    return None if x is None else c_void_p(x.ctypes.data)


# ------------------------------------------------------------------ scalar functions
def _unary(name, x):
    x = _a(x)
    y = np.empty_like(x)
    getattr(lib(), name)(_p(x), _p(y), x.size)
    return y


def det_exp_nonpos(x):
    return _unary("strict_det_exp_nonpos", x)


def det_log(x):
    return _unary("strict_det_log", x)


def det_sigmoid(x):
    return _unary("strict_det_sigmoid", x)


def det_logaddexp(a, b):
    a, b = _a(a), _a(b)
    y = np.empty_like(a)
    lib().strict_det_logaddexp(_p(a), _p(b), _p(y), a.size)
    return y


# ------------------------------------------------------------------ counter-based noise
def philox_normals(seed: int, step: int, N: int, M: int, d: int, traj0: int = 0) -> np.ndarray:
    """``(N, M, d)`` standard normals of counter step ``step`` (``include/mmf_philox.h``): what
    ``mmf_pf_dynamics_philox`` generates in its epilogue / ``mmf_philox_normals`` materialises."""
    out = np.empty((N, M, d), dtype=_F)
    lib().strict_philox_normals(seed, step, traj0, N, M, d, _p(out))
    return out


def philox_raw(key, counter):
    """Philox4x32-10 block: ``key`` (2 u32), ``counter`` (4 u32) -> 4 u32."""
    out = np.empty(4, dtype=np.uint32)
    lib().strict_philox_raw(*[int(k) for k in key], *[int(c) for c in counter], _p(out))
    return [int(x) for x in out]


def philox_uniforms(seed: int, step0: int, T: int, N: int, traj0: int = 0) -> np.ndarray:
    out = np.empty((T, N), dtype=_F)
    lib().strict_philox_uniforms(seed, step0, traj0, T, N, _p(out))
    return out


# ------------------------------------------------------------------ layers
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SQRT_SQ_PLUS = 0, 1, 2, 3


def linear(x, lin: nn.Linear, *, cols=None, act=ACT_NONE, res=None, bias=True, fparam=0.0):
    """``act(W[:, cols] x + b (+ res))`` as K7's LINEAR computes it (one fma chain per output)."""
    x = _a(x)
    W = _a(lin.weight)
    c0, c1 = cols if cols is not None else (0, W.shape[1])
    assert x.shape[1] == c1 - c0
    b = _a(lin.bias) if (bias and lin.bias is not None) else None
    res = None if res is None else _a(res)
    y = np.empty((x.shape[0], W.shape[0]), dtype=_F)
    lib().strict_linear(_p(x), x.shape[0], x.shape[1], _p(W), W.shape[1], c0, _p(b), _p(res), W.shape[0],
                        act, fparam, _p(y))
    return y


def res_linear(block, x):
    """fannypack ``resblocks.Linear``: ``relu(block2(relu(block1(x))) + x)``."""
    return linear(linear(x, block.block1, act=ACT_RELU), block.block2, act=ACT_RELU, res=x)


def vector_encoder(seq, x):
    """``Linear, ReLU, ResLinear`` (``door_models/layers.py:11-40,66-95``)."""
    return res_linear(seq[2], linear(x, seq[0], act=ACT_RELU))


def conv(x, c: nn.Conv2d, *, relu: bool, skip=None):
    x = _a(x)
    N, cin = x.shape[:2]
    w, b = _a(c.weight), _a(c.bias)
    cout, ks = w.shape[0], w.shape[2]
    out = np.empty((N, cout, 32, 32), dtype=_F)
    lib().strict_conv(_p(x), _p(w), _p(b), _p(None if skip is None else _a(skip)), N, cin, cout, ks, int(relu), _p(out))
    return out


def image_encoder(seq, images):
    """The default 32x32 image encoder (``door_models/layers.py:43-63``), f32 path of K4."""
    assert isinstance(seq[7], nn.Linear) and seq[7].in_features == 8192, "default (non-spanning) encoder only"
    x = _a(images).reshape(-1, 1, 32, 32)
    a = conv(x, seq[0], relu=True)
    b = conv(a, seq[2].block1, relu=True)
    c = conv(b, seq[2].block2, relu=True, skip=a)
    d = conv(c, seq[3], relu=True)
    e = conv(d, seq[5], relu=False)
    N = x.shape[0]
    feat = np.empty((N, 64), dtype=_F)
    fc, r = seq[7], seq[9]
    args = [_a(e).reshape(N, 8192), _a(fc.weight), _a(fc.bias), _a(r.block1.weight), _a(r.block1.bias),
            _a(r.block2.weight), _a(r.block2.bias)]
    lib().strict_fc_tail(*[_p(t) for t in args], N, _p(feat))
    return feat


# ------------------------------------------------------------------ per-particle networks
class ParticleNet:
    """One K2 network assembled from the oracle's modules (what ``engine.PackedParticleNet`` packs)."""

    def __init__(self, *, encoder, join, join_state_off, res_blocks, head, relu_after_join):
        self.keep = []

        def ptr(t):
            arr = _a(t)
            self.keep.append(arr)
            return c_void_p(arr.ctypes.data)

        n = _Net()
        n.d_in, n.n_res, n.relu_after_join = encoder[0].in_features, len(res_blocks), int(relu_after_join)
        n.n_out, n.join_in, n.join_state_off = head.out_features, join.in_features, join_state_off
        n.w_in, n.b_in = ptr(encoder[0].weight), ptr(encoder[0].bias)
        n.w_enc[0], n.b_enc[0] = ptr(encoder[2].block1.weight), ptr(encoder[2].block1.bias)
        n.w_enc[1], n.b_enc[1] = ptr(encoder[2].block2.weight), ptr(encoder[2].block2.bias)
        n.w_join = ptr(join.weight)
        for i, blk in enumerate(res_blocks):
            n.w_res[2 * i], n.b_res[2 * i] = ptr(blk.block1.weight), ptr(blk.block1.bias)
            n.w_res[2 * i + 1], n.b_res[2 * i + 1] = ptr(blk.block2.weight), ptr(blk.block2.bias)
        n.w_head = ptr(head.weight)
        self.net = n
        self.b_head = _a(head.bias)

    def raw(self, states, traj_bias, M):
        states, traj_bias = _a(states), _a(traj_bias)
        R = states.shape[0]
        out = np.empty((R, self.net.n_out), dtype=_F)
        lib().strict_particle_net(ctypes.byref(self.net), _p(states), _p(traj_bias), R, M, _p(out))
        return out


def dynamics_step(net: ParticleNet, states, traj_bias, noise, tril):
    """``(N, M, d)`` particles -> propagated particles (K2 dynamics kernel incl. its epilogue)."""
    N, M, d = states.shape
    flat = _a(states).reshape(N * M, d)
    raw = net.raw(flat, traj_bias, M)
    out = np.empty_like(flat)
    noise = None if noise is None else _a(noise).reshape(N * M, d)
    lib().strict_dynamics_epilogue(_p(raw), _p(net.b_head), _p(flat), _p(noise), _p(None if noise is None else _a(tril)),
                                   N * M, d, _p(out))
    return out.reshape(N, M, d)


def measure_step(net: ParticleNet, states, traj_bias, loglik, *, mod_logw=None, stride=0, combine=False):
    """One unimodal measurement launch: writes / combines into ``loglik (N, M)`` in place."""
    N, M, d = states.shape
    raw = net.raw(_a(states).reshape(N * M, d), traj_bias, M)
    lw = None if mod_logw is None else _a(mod_logw)
    lib().strict_measure_epilogue(_p(raw), float(net.b_head[0]), _p(lw), stride, N * M, M, int(combine), _p(loglik))
    return loglik


def k1_block(M: int) -> int:
    """Threads of the K1 workgroup for M particles (``pf_resample.hip``: a float4 per thread, <= 1024)."""
    return min(1024, ((M + 3) // 4 + 63) // 64 * 64)


def estimate(tot_logw, states):
    """K1's weighted-mean estimate from the un-normalised log-weights ``(N, M)``."""
    tot = _a(tot_logw)
    N, M = tot.shape
    e = _rs.quantise(tot)[1]
    states = _a(states)
    out = np.empty((N, states.shape[2]), dtype=_F)
    lib().strict_estimate(_p(_a(e)), _p(states), N, M, states.shape[2], k1_block(M), _p(out))
    return out


# ------------------------------------------------------------------ the particle filter
class StrictParticleFilter:
    """The oracle's particle filter (``oracle.models.ParticleFilter`` -- dynamics, one or two
    unimodal measurement models, optional crossmodal weight model) stepped on the strict arithmetic.
    Step order: ``oracle/tf/filters.py::ParticleFilter.forward`` (torchfilter);
    belief-independent terms are evaluated per step exactly as the engine's K7 / K4 do for a row."""

    def __init__(self, oracle_filter):
        f = oracle_filter
        self.f = f
        dyn = f.dynamics_model
        self.d = dyn.state_dim
        self.dyn_net = ParticleNet(encoder=dyn.state_layers, join=dyn.shared_layers[0], join_state_off=dyn.units,
                                   res_blocks=[dyn.shared_layers[1], dyn.shared_layers[2], dyn.shared_layers[3]],
                                   head=dyn.shared_layers[4], relu_after_join=False)
        meas = f.measurement_model
        self.subs = list(getattr(meas, "measurement_models", [meas]))
        self.enabled = list(getattr(meas, "enabled_models", [True] * len(self.subs)))
        self.weight_model = getattr(meas, "crossmodal_weight_model", None)
        self.meas_nets = [ParticleNet(encoder=m.state_layers, join=m.shared_layers[0],
                                      join_state_off=m.units * len(m.modalities),
                                      res_blocks=[m.shared_layers[2], m.shared_layers[3]],
                                      head=m.shared_layers[4], relu_after_join=True) for m in self.subs]
        self.states = None
        self.logw = None
        self.last_resample_indices = None
        self.last_total_log_weights = None

    # observation-only terms (K4 + K7) -------------------------------------------------
    @staticmethod
    def _sources(model, obs):
        srcs = []
        if "image" in model.modalities:
            srcs.append(image_encoder(model.observation_image_layers, obs["image"]))
        if "pos" in model.modalities:
            srcs.append(vector_encoder(model.observation_pos_layers, obs["gripper_pos"]))
        if "sensors" in model.modalities:
            srcs.append(vector_encoder(model.observation_sensors_layers, obs["gripper_sensors"]))
        return np.concatenate(srcs, axis=1)

    def control_bias(self, controls):
        dyn = self.f.dynamics_model
        c = vector_encoder(dyn.control_layers, controls)
        return linear(c, dyn.shared_layers[0], cols=(0, dyn.units))

    def measurement_terms(self, obs):
        biases = [linear(self._sources(m, obs), m.shared_layers[0], cols=(0, m.units * len(m.modalities)))
                  if on else None for m, on in zip(self.subs, self.enabled)]
        beta = None
        wm = self.weight_model
        if wm is not None:
            x = linear(self._sources(wm, obs), wm.fusion_layers[0], act=ACT_RELU)
            for blk in list(wm.fusion_layers)[2:-1]:
                x = res_linear(blk, x)
            beta = linear(x, wm.fusion_layers[-1])
            if wm.know_image_blackout:
                img = _a(obs["image"])
                dark = np.abs(img.reshape(img.shape[0], -1)).sum(axis=1) < 1e-8
                beta[dark, 0] = -np.inf
        return biases, beta

    # recursion -------------------------------------------------------------------------
    def set_belief(self, states, logw):
        self.states, self.logw = _a(states).copy(), _a(logw).copy()

    def step(self, *, observations, controls, eps, u):
        """One filter step on explicit noise ``eps (N, M, d)`` and resampling uniforms ``u (N,)``;
        returns the estimate ``(N, d)``.  Resamples (systematic), as ``eval()`` mode does."""
        N, M, d = self.states.shape
        tril = _a(self.f.dynamics_model.scale_tril())
        prop = dynamics_step(self.dyn_net, self.states, self.control_bias(controls), eps, tril)
        biases, beta = self.measurement_terms(observations)
        K = len(self.subs)
        loglik = np.empty((N, M), dtype=_F)
        first = True
        for i, on in enumerate(self.enabled):
            if not on:
                continue
            measure_step(self.meas_nets[i], prop, biases[i], loglik,
                         mod_logw=None if beta is None else beta.reshape(-1)[i:], stride=K, combine=not first)
            first = False
        tot = (self.logw + loglik).astype(_F)
        est = estimate(tot, prop)
        idx = _rs.resample_indices(tot, _a(u), "systematic")
        self.last_total_log_weights, self.last_log_likelihoods = tot, loglik
        self.last_resample_indices = idx
        self.states = np.take_along_axis(prop, idx[:, :, None].astype(np.int64), axis=1)
        self.logw = np.full((N, M), _F(-math.log(M)), dtype=_F)
        return est
