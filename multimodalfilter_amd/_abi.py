"""ctypes binding of ``libmmf_hip.so`` (the C ABI declared in ``include/mmf.h``).

This is the thin layer the product path goes through: tensors stay ``torch.Tensor`` (device
memory + stream plumbing), the arithmetic is the hand-written HIP behind these entry
points.  There is deliberately no fallback: if the library is missing, or a launch fails,
the call raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int32, c_size_t, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMF_LIB_PATH") or os.path.join(_HERE, "libmmf_hip.so")  # override: experiment builds only

MMF_UNITS = 64
MMF_MAX_RES = 3
MMF_MAX_STATE_DIM = 4
ABI_VERSION = 40
KIND_DYNAMICS, KIND_MEASURE, KIND_JACOBIAN = 0, 1, 2  # particle-network kinds (csrc/particle_net.hip)
PREC_F32, PREC_F16X3, PREC_BF16, PREC_F16X3_DUAL = 0, 1, 2, 3
PRECISIONS = {"f32": PREC_F32, "f16x3": PREC_F16X3}                          # per-particle networks (K2)
IMAGE_PRECISIONS = {"f32": PREC_F32, "f16x3": PREC_F16X3, "bf16": PREC_BF16}  # image encoder (K4)

_FP = c_void_p  # device pointers travel as integers


class MmfParticleNetDesc(Structure):
    _fields_ = [
        ("d_in", c_int32), ("n_res", c_int32), ("relu_after_join", c_int32), ("n_out", c_int32),
        ("join_in", c_int32), ("join_state_off", c_int32),
        ("w_in", _FP), ("b_in", _FP),
        ("w_enc", _FP * 2), ("b_enc", _FP * 2),
        ("w_join", _FP),
        ("w_res", _FP * (2 * MMF_MAX_RES)), ("b_res", _FP * (2 * MMF_MAX_RES)),
        ("w_head", _FP), ("b_head", _FP),
    ]


TRAJ_MAX_IO, TRAJ_SLOTS = 8, 8
TRAJ_LOAD, TRAJ_LINEAR, TRAJ_STORE, TRAJ_STORE_DIAG = 0, 1, 2, 3
TRAJ_MASK, TRAJ_ADD, TRAJ_ZERO, TRAJ_LOAD_ADD = 4, 5, 6, 7
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_SQRT_SQ_PLUS = 0, 1, 2, 3


class MmfTrajInstr(Structure):
    _fields_ = [("op", c_int32), ("dst", c_int32), ("src", c_int32 * 4), ("src_off", c_int32 * 4), ("src_dim", c_int32 * 4),
                ("out_dim", c_int32), ("w_off", c_int32), ("b_off", c_int32), ("res", c_int32),
                ("act", c_int32), ("io", c_int32), ("io_stride", c_int32), ("io_off", c_int32),
                ("fparam", c_float), ("dst_off", c_int32)]


class MmfPfTrainFinalizeArgs(Structure):
    _fields_ = [(n, c_int32) for n in ("T", "N", "SL", "S", "n_res", "d", "n_out", "join_in", "join_state_off", "fused",
                                       "beta_stride", "beta_col")] + \
               [(n, c_void_p) for n in ("pw", "pb", "p_first", "p_head", "p_dout", "p_traj", "grads", "bias_grad", "d_beta", "scratch")]


class MmfTrajPackDesc(Structure):
    _fields_ = [("src", c_uint64), ("kind", c_int32), ("rows", c_int32), ("ld", c_int32), ("col0", c_int32), ("dim", c_int32),
                ("out_pad", c_int32), ("dst_off", c_int32), ("reserved", c_int32)]


TRAJ_PACK_LAYER, TRAJ_PACK_TRANSPOSED, TRAJ_PACK_BIAS = 0, 1, 2


class MmfTrajGradDesc(Structure):
    _fields_ = [("x_col", c_int32), ("x_dim", c_int32), ("dz_col", c_int32), ("out_dim", c_int32),
                ("grad_off", c_int32), ("grad_ld", c_int32), ("bias_off", c_int32), ("reserved", c_int32)]


LOOP_MAX_MEAS = 4


class MmfPfLoopArgs(Structure):
    _fields_ = [("T", c_int32), ("N", c_int32), ("M", c_int32), ("d", c_int32), ("n_meas", c_int32),
                ("resample_mode", c_int32), ("precision", c_int32), ("n_res_dyn", c_int32),
                ("n_res_meas", c_int32), ("logw_stride", c_int32),
                ("dyn_packed", _FP), ("dyn_bias", _FP),
                ("meas_packed", _FP * LOOP_MAX_MEAS), ("meas_bias", _FP * LOOP_MAX_MEAS),
                ("meas_logw", _FP * LOOP_MAX_MEAS),
                ("noise", _FP), ("scale_tril", _FP), ("uniforms", _FP),
                ("states_a", _FP), ("states_b", _FP), ("logw_a", _FP), ("logw_b", _FP),
                ("loglik", _FP), ("estimates", _FP), ("range_flag", _FP),
                ("final_location", POINTER(c_int32)), ("events", POINTER(c_void_p)),
                ("event_stride", c_int32), ("loglik_steps", _FP), ("indices_steps", _FP),
                ("noise_seed", ctypes.c_uint64), ("noise_step0", ctypes.c_uint32), ("noise_traj0", ctypes.c_uint32),
                ("noise_mode", c_int32),
                ("soft_alpha", ctypes.c_float), ("estimate_argmax", c_int32), ("estimate_scratch", _FP),
                ("persistent", c_int32), ("n_sync_words", c_int32), ("sync_words", _FP)]


class MmfTrainNet(Structure):
    _fields_ = [("packed", _FP), ("packed_f32", _FP), ("packed_t", _FP), ("head_w", _FP), ("pw", _FP), ("pb", _FP),
                ("p_first", _FP), ("p_head", _FP), ("p_dout", _FP), ("p_traj", _FP), ("packed_dual", _FP)]


class MmfPfTrainArgs(Structure):
    _fields_ = [("T", c_int32), ("N", c_int32), ("M", c_int32), ("d", c_int32), ("n_meas", c_int32),
                ("n_res_dyn", c_int32), ("n_res_meas", c_int32), ("logw_stride", c_int32), ("precision", c_int32),
                ("chunk_traj", c_int32), ("n_splits", c_int32), ("n_slices", c_int32),
                ("dyn", MmfTrainNet), ("meas", MmfTrainNet * LOOP_MAX_MEAS),
                ("dyn_bias", _FP), ("meas_bias", _FP * LOOP_MAX_MEAS), ("meas_logw", _FP * LOOP_MAX_MEAS),
                ("noise", _FP), ("scale_tril", _FP), ("g_estimates", _FP),
                ("states", _FP), ("logw", _FP), ("estimates", _FP), ("d_states0", _FP), ("d_logw0", _FP),
                ("stash", _FP), ("mask", _FP), ("dz", _FP), ("raw", _FP), ("d_raw", _FP), ("loglik", _FP), ("ll_steps", _FP),
                ("g_states_a", _FP), ("g_states_b", _FP), ("g_logw_a", _FP), ("g_logw_b", _FP), ("d_tmp", _FP),
                ("range_flag", _FP), ("dz_scale", _FP),
                ("fused", c_int32), ("fused_act", _FP), ("fused_g_act", _FP), ("fused_sets", c_int32)]


class MmfTrainFusedArgs(Structure):
    _fields_ = [("packed_dual", _FP), ("n_res", c_int32), ("kind", c_int32), ("d", c_int32), ("N", c_int32), ("M", c_int32),
                ("n_slots", c_int32), ("states", _FP), ("traj_bias", _FP), ("d_out", _FP), ("g_next", _FP), ("d_raw", _FP),
                ("act", _FP), ("g_act", _FP), ("d_states", _FP), ("dz_first_h", _FP), ("sc_first", _FP), ("dz_join_h", _FP),
                ("sc_join", _FP), ("h_last_h", _FP), ("pw", _FP), ("pb", _FP), ("d_states_base", _FP)]


class MmfEkfLoopArgs(Structure):
    _fields_ = [("T", c_int32), ("N", c_int32), ("d", c_int32), ("K", c_int32),
                ("fusion", c_int32), ("feedback", c_int32), ("n_res_dyn", c_int32), ("precision", c_int32),
                ("range_flag", _FP),
                ("dyn_packed", _FP * LOOP_MAX_MEAS), ("dyn_bias", _FP * LOOP_MAX_MEAS),
                ("q_tril", _FP), ("z", _FP), ("r_tril", _FP), ("fuse_w", _FP),
                ("mu", _FP), ("Sigma", _FP), ("mu_pred", _FP), ("A", _FP), ("Sigma_f", _FP),
                ("estimates", _FP), ("feedback_gate", _FP),
                ("persistent", c_int32), ("n_sync_words", c_int32), ("sync_words", _FP)]


class MmfImageEncoderDesc(Structure):
    _fields_ = [("conv_w", _FP * 5), ("conv_b", _FP * 5), ("fc_w", _FP), ("fc_b", _FP),
                ("res_w", _FP * 2), ("res_b", _FP * 2), ("variant", c_int32)]


SIGNATURES = {
    "mmf_version": (c_int, []),
    "mmf_pf_reweight_resample": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP,
                                         c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mmf_pf_reweight_resample_soft": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP,
                                              c_int, c_int, c_int, c_int, c_int, ctypes.c_float, c_void_p]),
    "mmf_pf_reweight_resample_lds_bytes": (c_size_t, [c_int, c_int]),
    "mmf_pf_set_resample_cluster": (None, [c_int]),
    "mmf_pf_get_resample_cluster": (c_int, []),
    "mmf_particle_net_floats": (c_size_t, [c_int]),
    "mmf_pack_particle_net": (c_int, [POINTER(MmfParticleNetDesc), _FP, c_int, c_void_p]),
    "mmf_pf_dynamics": (c_int, [_FP, c_int, c_int, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_pf_measure": (c_int, [_FP, c_int, c_int, _FP, _FP, _FP, c_int, _FP, c_int, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_pf_measure_multi": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, _FP, POINTER(c_void_p), POINTER(c_void_p), c_int,
                                     POINTER(c_void_p), _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_dynamics_jacobian": (c_int, [_FP, c_int, c_int, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_void_p]),
    "mmf_ekf_step": (c_int, [_FP] * 10 + [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mmf_ekf_step_gated": (c_int, [_FP] * 10 + [c_int, c_int, c_int, c_int, c_int, _FP, c_void_p]),
    "mmf_ekf_step_backward": (c_int, [_FP] * 13 + [c_int, c_int, c_int, c_void_p]),
    "mmf_ukf_sigma_points": (c_int, [_FP, _FP, ctypes.c_float, _FP, _FP, c_int, c_int, c_void_p]),
    "mmf_ukf_moments": (c_int, [_FP, ctypes.c_float, ctypes.c_float, ctypes.c_float, _FP, _FP, _FP, c_int, c_int, c_void_p]),
    "mmf_pf_reweight_backward": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_pf_init_particles": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_pf_forward_loop": (c_int, [POINTER(MmfPfLoopArgs), c_void_p]),
    "mmf_pf_argmax_estimate": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_pf_persistent_plan": (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "mmf_pf_persistent_sync_words": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mmf_pf_dynamics_philox": (c_int, [_FP, c_int, c_int, _FP, _FP, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                       _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_philox_normals": (c_int, [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_philox_uniforms": (c_int, [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _FP, c_int, c_int, c_void_p]),
    "mmf_dynamics_forward_loop": (c_int, [_FP, c_int, c_int, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_traj_program": (c_int, [_FP, c_int, _FP, POINTER(c_void_p), c_int, c_int, c_int, c_void_p]),
    "mmf_pf_train_finalize": (c_int, [POINTER(MmfPfTrainFinalizeArgs), c_void_p]),
    "mmf_traj_pack": (c_int, [c_void_p, c_int, _FP, c_void_p]),
    "mmf_traj_weight_grads": (c_int, [c_void_p, c_int, _FP, c_int, _FP, c_int, _FP, c_int, _FP, c_int, c_int, c_void_p]),
    "mmf_fc64_train_forward": (c_int, [_FP, _FP, _FP, _FP, _FP, c_int, c_int, c_void_p]),
    "mmf_fc64_train_backward": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_void_p]),
    "mmf_fuse_virtual_sensors": (c_int, [_FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_void_p]),
    "mmf_ekf_forward_loop": (c_int, [POINTER(MmfEkfLoopArgs), c_void_p]),
    "mmf_ekf_persistent_plan": (c_int, [c_int, c_int]),
    "mmf_ekf_persistent_sync_words": (c_size_t, [c_int, c_int, c_int]),
    "mmf_dynamics_jacobian_multi": (c_int, [POINTER(c_void_p), c_int, c_int, _FP, POINTER(c_void_p), _FP, _FP, _FP,
                                            c_int, c_int, c_int, c_void_p]),
    "mmf_particle_net_train_forward": (c_int, [_FP, c_int, c_int, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_particle_net_weight_grads": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
    "mmf_particle_net_small_grads": (c_int, [_FP] * 9 + [c_int] * 5 + [c_void_p]),
    "mmf_particle_net_weight_grads_acc": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, c_void_p]),
    "mmf_pf_train_forward": (c_int, [POINTER(MmfPfTrainArgs), c_void_p]),
    "mmf_pf_train_backward": (c_int, [POINTER(MmfPfTrainArgs), c_void_p]),
    "mmf_particle_net_train_fused": (c_int, [POINTER(MmfTrainFusedArgs), c_void_p]),
    "mmf_particle_net_train_fused_multi": (c_int, [POINTER(MmfTrainFusedArgs), c_int, c_void_p]),
    "mmf_particle_net_train_backward": (c_int, [_FP, _FP, c_int, c_int, _FP, _FP, _FP, _FP, c_int, c_int, c_void_p]),
    "mmf_image_encoder_floats": (c_size_t, []),
    "mmf_image_encoder_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mmf_pack_image_encoder": (c_int, [POINTER(MmfImageEncoderDesc), _FP, c_void_p]),
    "mmf_image_convs_backward_floats": (c_size_t, []),
    "mmf_pack_image_convs_backward": (c_int, [POINTER(MmfImageEncoderDesc), _FP, c_void_p]),
    "mmf_image_convs_train_forward": (c_int, [_FP] * 8 + [c_int, c_int, c_void_p]),
    "mmf_image_convs_train_backward": (c_int, [_FP] * 10 + [c_int, c_void_p]),
    "mmf_conv_weight_grads": (c_int, [_FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, _FP, _FP, c_void_p]),
    "mmf_image_convs_train_backward_h": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_void_p]),
    "mmf_conv_weight_grads_h": (c_int, [_FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_int, _FP, _FP, c_void_p]),
    "mmf_image_encoder": (c_int, [POINTER(c_void_p), c_int, _FP, _FP, _FP, _FP, c_int, c_int, c_int, c_void_p]),
}

_lib = None


class MmfError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MmfError(
                f"{LIB_PATH} is missing: build it with `python -m multimodalfilter_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        got = lib.mmf_version()
        if got != ABI_VERSION:
            raise MmfError(f"ABI mismatch: library {got}, binding {ABI_VERSION}")
        _lib = lib
    return _lib


def _check(code: int, what: str):
    if code != 0:
        kind = "argument error" if code < 0 else "hipError_t"
        raise MmfError(f"{what} failed: {kind} {code}")


def ptr(t: torch.Tensor, *, dtype=torch.float32):
    """Device pointer of a contiguous CUDA/HIP tensor (``None`` -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MmfError("libmmf_hip works on device memory only (got a CPU tensor); no CPU fallback")
    if t.dtype != dtype:
        raise MmfError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise MmfError("tensor must be contiguous")
    return t.data_ptr()


def _on(t: torch.Tensor):
    """Device context of ``t``; CPU tensors are refused here, before any launch."""
    ptr(t, dtype=t.dtype)
    return torch.cuda.device(t.device)


def stream_of(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


# ------------------------------------------------------------------ typed wrappers
def pf_reweight_resample(loglik, logw_in, states_in, u, estimate, states_out, logw_out,
                         indices_out, mode: int, soft_alpha: float = 1.0):
    """K1; ``soft_alpha < 1`` (modes 1 / 2): torchfilter's soft resampling through
    ``mmf_pf_reweight_resample_soft``."""
    N, M, d = states_in.shape
    M_out = logw_out.shape[1]
    assert loglik.shape == (N, M) and logw_in.shape == (N, M) and estimate.shape == (N, d)
    with _on(states_in):
        if soft_alpha < 1.0 and mode != 0:
            _check(load().mmf_pf_reweight_resample_soft(
                ptr(loglik), ptr(logw_in), ptr(states_in), ptr(u), ptr(estimate), ptr(states_out),
                ptr(logw_out), ptr(indices_out, dtype=torch.int32), N, M, M_out, d, mode, float(soft_alpha),
                stream_of(states_in)), "mmf_pf_reweight_resample_soft")
            return
        _check(load().mmf_pf_reweight_resample(
            ptr(loglik), ptr(logw_in), ptr(states_in), ptr(u), ptr(estimate), ptr(states_out),
            ptr(logw_out), ptr(indices_out, dtype=torch.int32), N, M, M_out, d, mode,
            stream_of(states_in)), "mmf_pf_reweight_resample")


def pf_set_resample_cluster(enabled: bool) -> bool:
    """K1 for few trajectories as a cluster of workgroups per trajectory (same bits; off by default: no faster). Returns the previous setting."""
    was = bool(load().mmf_pf_get_resample_cluster())
    load().mmf_pf_set_resample_cluster(int(bool(enabled)))
    return was


def particle_net_floats(n_res: int) -> int:
    return int(load().mmf_particle_net_floats(n_res))


def pack_particle_net(desc: MmfParticleNetDesc, packed: torch.Tensor, precision: int):
    with _on(packed):
        _check(load().mmf_pack_particle_net(ctypes.byref(desc), ptr(packed), precision,
                                            stream_of(packed)), "mmf_pack_particle_net")


def pf_dynamics(packed, n_res, precision, states_in, traj_bias, noise, scale_tril, states_out,
                range_flag, N, M, d):
    with _on(states_in):
        _check(load().mmf_pf_dynamics(ptr(packed), n_res, precision, ptr(states_in), ptr(traj_bias), ptr(noise),
                                      ptr(scale_tril), ptr(states_out),
                                      ptr(range_flag, dtype=torch.int32), N, M, d,
                                      stream_of(states_in)), "mmf_pf_dynamics")


def dynamics_forward_loop(packed, n_res, precision, x0, traj_bias, out, range_flag, T, N, d):
    with _on(x0):
        _check(load().mmf_dynamics_forward_loop(ptr(packed), n_res, precision, ptr(x0), ptr(traj_bias), ptr(out),
                                                ptr(range_flag, dtype=torch.int32), T, N, d, stream_of(x0)),
               "mmf_dynamics_forward_loop")


def pf_dynamics_philox(packed, n_res, precision, states_in, traj_bias, seed, step, traj0, scale_tril, states_out,
                       range_flag, N, M, d):
    with _on(states_in):
        _check(load().mmf_pf_dynamics_philox(ptr(packed), n_res, precision, ptr(states_in), ptr(traj_bias), seed, step,
                                             traj0, ptr(scale_tril), ptr(states_out),
                                             ptr(range_flag, dtype=torch.int32), N, M, d, stream_of(states_in)),
               "mmf_pf_dynamics_philox")


def philox_normals(seed: int, step: int, traj0: int, out: torch.Tensor):
    N, M, d = out.shape
    with _on(out):
        _check(load().mmf_philox_normals(seed, step, traj0, ptr(out), N, M, d, stream_of(out)), "mmf_philox_normals")


def philox_uniforms(seed: int, step0: int, traj0: int, out: torch.Tensor):
    T, N = out.shape
    with _on(out):
        _check(load().mmf_philox_uniforms(seed, step0, traj0, ptr(out), T, N, stream_of(out)), "mmf_philox_uniforms")


def pf_measure(packed, n_res, precision, states, traj_bias, modality_logw, logw_stride, loglik, combine,
               range_flag, N, M, d):
    with _on(states):
        _check(load().mmf_pf_measure(ptr(packed), n_res, precision, ptr(states), ptr(traj_bias),
                                     ptr(modality_logw), logw_stride, ptr(loglik), int(combine),
                                     ptr(range_flag, dtype=torch.int32), N, M, d,
                                     stream_of(states)), "mmf_pf_measure")


def dynamics_jacobian(packed, n_res, precision, states_in, traj_bias, states_out, jac, range_flag, N, d):
    with _on(states_in):
        _check(load().mmf_dynamics_jacobian(ptr(packed), n_res, precision, ptr(states_in), ptr(traj_bias),
                                            ptr(states_out), ptr(jac), ptr(range_flag, dtype=torch.int32), N, d,
                                            stream_of(states_in)), "mmf_dynamics_jacobian")


def pf_persistent_plan(N: int, M: int, n_meas: int) -> int:
    """Workgroups the persistent step loop would use for this problem; <= 0: not eligible (include/mmf.h)."""
    return int(load().mmf_pf_persistent_plan(N, M, n_meas, None, None, None))


def pf_persistent_sync_words(N: int, M: int, d: int, n_meas: int) -> int:
    return int(load().mmf_pf_persistent_sync_words(N, M, d, n_meas))


def ekf_persistent_plan(N: int, K: int) -> int:
    """Workgroups the persistent EKF step loop would use for this problem; <= 0: not eligible (include/mmf.h)."""
    return int(load().mmf_ekf_persistent_plan(N, K))


def ekf_persistent_sync_words(N: int, K: int, d: int) -> int:
    return int(load().mmf_ekf_persistent_sync_words(N, K, d))


def pf_argmax_estimate(loglik, logw_in, states, estimate):
    N, M, d = states.shape
    with _on(states):
        _check(load().mmf_pf_argmax_estimate(ptr(loglik), ptr(logw_in), ptr(states), ptr(estimate), N, M, d,
                                             stream_of(states)), "mmf_pf_argmax_estimate")


def ekf_step(A, mu_pred, q_tril, z, r_tril, fuse_w, mu, Sigma, mu_f, Sigma_f, fusion: int,
             feedback: int, feedback_gate=None):
    """``feedback_gate``: int32 device word; the write-back applies only where it is non-zero
    (``mmf_ekf_step_gated``)."""
    K, N, d = mu_pred.shape
    with _on(mu_pred):
        if feedback_gate is not None:
            _check(load().mmf_ekf_step_gated(ptr(A), ptr(mu_pred), ptr(q_tril), ptr(z), ptr(r_tril),
                                             ptr(fuse_w), ptr(mu), ptr(Sigma), ptr(mu_f), ptr(Sigma_f),
                                             N, d, K, fusion, feedback, ptr(feedback_gate, dtype=torch.int32),
                                             stream_of(mu_pred)), "mmf_ekf_step_gated")
            return
        _check(load().mmf_ekf_step(ptr(A), ptr(mu_pred), ptr(q_tril), ptr(z), ptr(r_tril),
                                   ptr(fuse_w), ptr(mu), ptr(Sigma), ptr(mu_f), ptr(Sigma_f),
                                   N, d, K, fusion, feedback, stream_of(mu_pred)), "mmf_ekf_step")


def image_convs_train_forward(packed, images, a1, h, a2, a3, a4, range_flag=None, precision: int = PREC_F32):
    with _on(images):
        _check(load().mmf_image_convs_train_forward(ptr(packed), ptr(images), ptr(a1), ptr(h), ptr(a2), ptr(a3), ptr(a4),
                                                    ptr(range_flag, dtype=torch.int32), precision,
                                                    images.shape[0], stream_of(images)), "mmf_image_convs_train_forward")


def image_convs_train_backward(packed_bwd, a1, h, a2, a3, g_a4, g1, gh, g2, g3):
    with _on(g_a4):
        _check(load().mmf_image_convs_train_backward(ptr(packed_bwd), ptr(a1), ptr(h), ptr(a2), ptr(a3), ptr(g_a4), ptr(g1),
                                                     ptr(gh), ptr(g2), ptr(g3), g_a4.shape[0], stream_of(g_a4)),
               "mmf_image_convs_train_backward")


def image_convs_train_backward_h(packed_bwd, a1, h, a2, a3, g_a4, g1, gh, g2, g3, scratch):
    assert scratch.numel() >= 4 and scratch.dtype == torch.float32
    with _on(g_a4):
        _check(load().mmf_image_convs_train_backward_h(ptr(packed_bwd), ptr(a1), ptr(h), ptr(a2), ptr(a3), ptr(g_a4), ptr(g1),
                                                       ptr(gh), ptr(g2), ptr(g3), ptr(scratch), g_a4.shape[0], stream_of(g_a4)),
               "mmf_image_convs_train_backward_h")


def conv_weight_grads(g, act, partial, partial_b, n_blocks: int, dw=None, db=None):
    """``partial (n_blocks, 9, 32, 32)``, ``partial_b (n_blocks, 32)``: one slot per workgroup; ``dw (co, ci, k, k)`` /
    ``db (co)``: the slots summed into ``nn.Conv2d``'s layout by a second launch."""
    assert partial.numel() >= n_blocks * 9 * 32 * 32 and partial_b.numel() >= n_blocks * 32
    with _on(g):
        _check(load().mmf_conv_weight_grads(ptr(g), ptr(act), ptr(partial), ptr(partial_b), g.shape[0], g.shape[1],
                                            act.shape[1], n_blocks, ptr(dw), ptr(db), stream_of(g)), "mmf_conv_weight_grads")


def conv_weight_grads_h(g, act, g_absmax, partial, partial_b, range_flag, n_blocks: int, dw=None, db=None):
    """``conv_weight_grads`` on the f16 matrix pipe with three products per product; ``g_absmax``: device scalar (largest |g|)."""
    assert partial.numel() >= n_blocks * 9 * 32 * 32 and partial_b.numel() >= n_blocks * 32 and g_absmax.numel() == 1
    with _on(g):
        _check(load().mmf_conv_weight_grads_h(ptr(g), ptr(act), ptr(g_absmax), ptr(partial), ptr(partial_b),
                                              ptr(range_flag, dtype=torch.int32), g.shape[0], g.shape[1], act.shape[1], n_blocks,
                                              ptr(dw), ptr(db), stream_of(g)), "mmf_conv_weight_grads_h")


def image_convs_backward_floats() -> int:
    return int(load().mmf_image_convs_backward_floats())


def pack_image_convs_backward(desc: MmfImageEncoderDesc, packed: torch.Tensor):
    with _on(packed):
        _check(load().mmf_pack_image_convs_backward(ctypes.byref(desc), ptr(packed), stream_of(packed)),
               "mmf_pack_image_convs_backward")


def ekf_step_backward(A, mu_pred, q_tril, z, r_tril, Sigma_in, g_mu, g_Sigma, g_A, g_mu_pred, g_z, g_r_tril, g_Sigma_in):
    K, N, d = mu_pred.shape
    with _on(mu_pred):
        _check(load().mmf_ekf_step_backward(ptr(A), ptr(mu_pred), ptr(q_tril), ptr(z), ptr(r_tril), ptr(Sigma_in),
                                            ptr(g_mu), ptr(g_Sigma), ptr(g_A), ptr(g_mu_pred), ptr(g_z), ptr(g_r_tril),
                                            ptr(g_Sigma_in), N, d, K, stream_of(mu_pred)), "mmf_ekf_step_backward")


def ukf_sigma_points(mu, Sigma, scale: float, points, not_pd):
    N, d = mu.shape
    with _on(mu):
        _check(load().mmf_ukf_sigma_points(ptr(mu), ptr(Sigma), float(scale), ptr(points), ptr(not_pd, dtype=torch.int32),
                                           N, d, stream_of(mu)), "mmf_ukf_sigma_points")


def ukf_moments(points, wm0: float, wc0: float, wi: float, q_tril, mu_pred, Sigma_pred):
    N, d = mu_pred.shape
    with _on(points):
        _check(load().mmf_ukf_moments(ptr(points), float(wm0), float(wc0), float(wi), ptr(q_tril), ptr(mu_pred),
                                      ptr(Sigma_pred), N, d, stream_of(points)), "mmf_ukf_moments")


def particle_net_train_forward(packed, n_res: int, kind: int, states, traj_bias, stash, mask, out, N: int, M: int, d: int):
    with _on(states):
        _check(load().mmf_particle_net_train_forward(ptr(packed), n_res, kind, ptr(states), ptr(traj_bias),
                                                     ptr(stash), ptr(mask, dtype=torch.int32), ptr(out), N, M, d,
                                                     stream_of(states)),
               "mmf_particle_net_train_forward")


def particle_net_train_backward(packed_t, head_w, n_res: int, kind: int, mask, d_out, dz, d_states, R: int, d: int):
    with _on(d_out):
        _check(load().mmf_particle_net_train_backward(ptr(packed_t), ptr(head_w), n_res, kind,
                                                      ptr(mask, dtype=torch.int32), ptr(d_out), ptr(dz), ptr(d_states),
                                                      R, d, stream_of(d_out)),
               "mmf_particle_net_train_backward")


def particle_net_weight_grads(dz, stash, partial_w, partial_b, n_layers: int, R: int, n_splits: int):
    with _on(dz):
        _check(load().mmf_particle_net_weight_grads(ptr(dz), ptr(stash), ptr(partial_w), ptr(partial_b),
                                                    n_layers, R, n_splits, stream_of(dz)),
               "mmf_particle_net_weight_grads")


def particle_net_small_grads(dz_first, dz_join, h_last, states, d_out, p_first, p_head, p_dout, p_traj,
                             N: int, M: int, n_slices: int):
    with _on(dz_first):
        _check(load().mmf_particle_net_small_grads(ptr(dz_first), ptr(dz_join), ptr(h_last), ptr(states), ptr(d_out),
                                                   ptr(p_first), ptr(p_head), ptr(p_dout), ptr(p_traj), N, M,
                                                   states.shape[1], d_out.shape[1], n_slices, stream_of(dz_first)),
               "mmf_particle_net_small_grads")


def pf_train_forward(args: MmfPfTrainArgs, like: torch.Tensor):
    with _on(like):
        _check(load().mmf_pf_train_forward(ctypes.byref(args), stream_of(like)), "mmf_pf_train_forward")


def pf_train_backward(args: MmfPfTrainArgs, like: torch.Tensor):
    with _on(like):
        _check(load().mmf_pf_train_backward(ctypes.byref(args), stream_of(like)), "mmf_pf_train_backward")


def particle_net_train_fused(args: MmfTrainFusedArgs, like: torch.Tensor):
    """One fused network call of the training backward (see include/mmf.h)."""
    with _on(like):
        _check(load().mmf_particle_net_train_fused(ctypes.byref(args), stream_of(like)), "mmf_particle_net_train_fused")


def pf_measure_multi(packed, n_res: int, precision: int, states, traj_bias, modality_logw, logw_stride: int, loglik, range_flag_t):
    """Every modality's own log-likelihood of the same ``(N, M, d)`` particles in one launch; lists of tensors (``modality_logw``
    entries may be ``None``)."""
    n = len(packed)
    arr = lambda ts: (c_void_p * n)(*[ptr(t) for t in ts])
    N, M, d = states.shape
    with _on(states):
        _check(load().mmf_pf_measure_multi(arr(packed), n, n_res, precision, ptr(states), arr(traj_bias), arr(modality_logw), logw_stride,
                                           arr(loglik), ptr(range_flag_t, dtype=torch.int32), N, M, d, stream_of(states)),
               "mmf_pf_measure_multi")


def particle_net_train_fused_multi(args_list, like: torch.Tensor):
    """Several measurement networks' fused calls of one step as one launch (see include/mmf.h)."""
    arr = (MmfTrainFusedArgs * len(args_list))(*args_list)
    with _on(like):
        _check(load().mmf_particle_net_train_fused_multi(arr, len(args_list), stream_of(like)), "mmf_particle_net_train_fused_multi")


def fuse_virtual_sensors(z, tril, w, z_out, tril_out, mode: int):
    K, N, d = z.shape
    with _on(z):
        _check(load().mmf_fuse_virtual_sensors(ptr(z), ptr(tril), ptr(w), ptr(z_out), ptr(tril_out), N, d, K, mode,
                                               stream_of(z)), "mmf_fuse_virtual_sensors")


def ekf_forward_loop(args: MmfEkfLoopArgs, like: torch.Tensor):
    """Enqueue T fused-EKF steps (see include/mmf.h)."""
    with _on(like):
        _check(load().mmf_ekf_forward_loop(ctypes.byref(args), stream_of(like)), "mmf_ekf_forward_loop")


def image_encoder_floats() -> int:
    return int(load().mmf_image_encoder_floats())


def image_encoder_workspace_bytes(n_images: int, n_nets: int) -> int:
    return int(load().mmf_image_encoder_workspace_bytes(n_images, n_nets))


def pack_image_encoder(desc: MmfImageEncoderDesc, packed: torch.Tensor):
    with _on(packed):
        _check(load().mmf_pack_image_encoder(ctypes.byref(desc), ptr(packed), stream_of(packed)),
               "mmf_pack_image_encoder")


ENCODER_DEFAULT, ENCODER_SPANNING_POOL = 0, 1


def image_encoder(blobs, images: torch.Tensor, feat: torch.Tensor, workspace: torch.Tensor,
                  range_flag, precision: int, variant: int = ENCODER_DEFAULT):
    n = len(blobs)
    arr = (c_void_p * n)(*[ptr(b) for b in blobs])
    N = images.shape[0]
    assert tuple(images.shape[1:]) == (32, 32) and tuple(feat.shape) == (n, N, 64)
    with _on(images):
        _check(load().mmf_image_encoder(arr, n, ptr(images), ptr(feat),
                                        ptr(workspace, dtype=torch.uint8),
                                        ptr(range_flag, dtype=torch.int32), precision, variant, N,
                                        stream_of(images)),
               "mmf_image_encoder")


def traj_program(prog: torch.Tensor, n_instr: int, weights: torch.Tensor, io_tensors, R: int,
                 n_slots: int = TRAJ_SLOTS, vec_width: int = 128):
    """``prog``: uint8 device tensor holding ``n_instr`` MmfTrajInstr; ``io_tensors``: up to
    TRAJ_MAX_IO float32 device tensors (``None`` = unused); ``n_slots`` / ``vec_width``: what the
    program uses (sizes the launch's LDS)."""
    arr = (c_void_p * TRAJ_MAX_IO)(*[ptr(t) for t in io_tensors] + [None] * (TRAJ_MAX_IO - len(io_tensors)))
    with _on(weights):
        _check(load().mmf_traj_program(ptr(prog, dtype=torch.uint8), n_instr, ptr(weights), arr, R,
                                       n_slots, vec_width, stream_of(weights)), "mmf_traj_program")


def pf_train_finalize(args: MmfPfTrainFinalizeArgs, like: torch.Tensor):
    with _on(like):
        _check(load().mmf_pf_train_finalize(ctypes.byref(args), stream_of(like)), "mmf_pf_train_finalize")


def traj_pack(desc: torch.Tensor, n_desc: int, blob: torch.Tensor):
    with _on(blob):
        _check(load().mmf_traj_pack(ptr(desc, dtype=torch.uint8), n_desc, ptr(blob), stream_of(blob)), "mmf_traj_pack")


def traj_weight_grads(desc: torch.Tensor, n_desc: int, stash: torch.Tensor, dz: torch.Tensor, grads: torch.Tensor,
                      partials, n_slices: int, R: int):
    """``desc``: uint8 device tensor of ``n_desc`` MmfTrajGradDesc; ``stash`` / ``dz``: ``(R, ld)``; ``grads``: flat."""
    with _on(grads):
        _check(load().mmf_traj_weight_grads(ptr(desc, dtype=torch.uint8), n_desc, ptr(stash), stash.shape[1], ptr(dz),
                                            dz.shape[1], ptr(grads), grads.numel(), ptr(partials), n_slices, R,
                                            stream_of(grads)), "mmf_traj_weight_grads")


def fc64_train_forward(x, w, b, y):
    R, K = x.shape
    partial = torch.empty((max(1, K // 512), R, 64), dtype=torch.float32, device=x.device)
    with _on(x):
        _check(load().mmf_fc64_train_forward(ptr(x), ptr(w), ptr(b), ptr(y), ptr(partial), R, K, stream_of(x)),
               "mmf_fc64_train_forward")


def fc64_train_backward(g, x, w, dx, dw, db):
    R, K = x.shape
    with _on(x):
        _check(load().mmf_fc64_train_backward(ptr(g), ptr(x), ptr(w), ptr(dx), ptr(dw), ptr(db), R, K, stream_of(x)),
               "mmf_fc64_train_backward")


def pf_reweight_backward(logw_out, states, g_estimate, g_logw_out, d_a, d_states):
    N, M, d = states.shape
    with _on(states):
        _check(load().mmf_pf_reweight_backward(ptr(logw_out), ptr(states), ptr(g_estimate), ptr(g_logw_out), ptr(d_a),
                                               ptr(d_states), N, M, d, stream_of(states)), "mmf_pf_reweight_backward")


def pf_init_particles(mean, covariance, eps, states, logw, not_pd):
    N, M, d = eps.shape
    with _on(eps):
        _check(load().mmf_pf_init_particles(ptr(mean), ptr(covariance), ptr(eps), ptr(states), ptr(logw),
                                            ptr(not_pd, dtype=torch.int32), N, M, d, stream_of(eps)),
               "mmf_pf_init_particles")


def pf_forward_loop(args: MmfPfLoopArgs, like: torch.Tensor, events=None, event_stride: int = 1) -> int:
    """Enqueue T filter steps; returns the final-location bits (see include/mmf.h).
    ``events``: optional flat list of created ``torch.cuda.Event`` (timing) recorded in C
    around the launches of every ``event_stride``-th step."""
    loc = c_int32(0)
    args.final_location = ctypes.pointer(loc)
    if events is not None:
        arr = (c_void_p * len(events))(*[e.cuda_event for e in events])
        args.events = arr
        args.event_stride = event_stride
    with _on(like):
        _check(load().mmf_pf_forward_loop(ctypes.byref(args), stream_of(like)), "mmf_pf_forward_loop")
    return int(loc.value)
