"""One fused network call of the training backward (``mmf_particle_net_train_fused``, C ABI) against the exact-fp32 K6
step kernels (``engine.ParticleNetFunction``): shared by tests/test_gpu_training.py and scripts/debug/fused_check.py."""
import ctypes
import time

import torch

import multimodalfilter_amd as mmf
from multimodalfilter_amd import _abi, engine


def run_case(task, kind, N, M, seed=0, verbose=True, timing=False, return_rows=False):
    dev = torch.device("cuda:0")
    ns = mmf.door_models if task == "door" else mmf.push_models
    P = task.capitalize()
    torch.manual_seed(seed + N * 100 + M)
    if kind == "dynamics":
        model = ns.DoorDynamicsModelBrent() if task == "door" else getattr(ns, P + "DynamicsModel")()
    else:
        model = getattr(ns, P + "MeasurementModel")(modalities={"pos", "sensors"})
    model.to(dev)
    net = model._net
    d, R, U = net.d_in, N * M, 64
    NL = 3 + 2 * net.n_res
    g = torch.Generator().manual_seed(5 + seed)
    states = (0.7 * torch.randn((R, d), generator=g)).to(dev).requires_grad_(True)
    tbias = torch.randn((N, U), generator=g).to(dev).requires_grad_(True)
    params = net._sources()
    k = 0 if kind == "dynamics" else 1
    # gradient magnitudes that differ by orders of magnitude between trajectories (particle weights do that)
    row_scale = torch.logspace(0, -4, N).repeat_interleave(M).to(dev)
    out = engine.ParticleNetFunction.apply(net, k, N, M, states, tbias, *params)
    if kind == "dynamics":
        g_next = (torch.randn((R, d), generator=g).to(dev) * row_scale[:, None]).contiguous()
        x_next = out[:, :d] * torch.sigmoid(out[:, d:])
        want = torch.autograd.grad(x_next, [states, tbias, out] + params, g_next)
        d_out_ref = want[2]
        want = want[:2] + want[3:]
    else:
        d_out_ref = (torch.randn((R, 1), generator=g).to(dev) * row_scale[:, None]).contiguous()
        want = torch.autograd.grad(out, [states, tbias] + params, d_out_ref)

    S = min(256, max(1, -(-R // 128)))
    E = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    H = lambda *s: torch.empty(s, dtype=torch.float16, device=dev)
    bufs = dict(d_states=E(R, d), dzf=H(R, U), scf=E(R), dzj=H(R, U), scj=E(R), hl=H(R, U),
                pw=torch.zeros((NL, S, U, U), device=dev), pb=torch.zeros((NL, S, U), device=dev),
                d_raw=E(R, d + 1), act=E(R, U), g_act=E(R, U))
    a = _abi.MmfTrainFusedArgs()
    Pp = lambda t: ctypes.c_void_p(t.data_ptr())
    blob = net.blob(_abi.PREC_F16X3_DUAL)
    a.packed_dual, a.n_res, a.kind, a.d, a.N, a.M, a.n_slots = Pp(blob), net.n_res, k, d, N, M, S
    st = states.detach().contiguous()
    tb = tbias.detach().contiguous()
    a.states, a.traj_bias = Pp(st), Pp(tb)
    if kind == "dynamics":
        a.g_next, a.d_raw, a.act, a.g_act = Pp(g_next), Pp(bufs["d_raw"]), Pp(bufs["act"]), Pp(bufs["g_act"])
    else:
        d_out = d_out_ref.reshape(R).contiguous()
        a.d_out = Pp(d_out)
    a.d_states, a.dz_first_h, a.sc_first, a.dz_join_h, a.sc_join, a.h_last_h = (Pp(bufs[n]) for n in ("d_states", "dzf", "scf", "dzj", "scj", "hl"))
    a.pw, a.pb = Pp(bufs["pw"]), Pp(bufs["pb"])
    _abi.particle_net_train_fused(a, st)
    torch.cuda.synchronize()
    if timing:
        for _ in range(3):
            _abi.particle_net_train_fused(a, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            _abi.particle_net_train_fused(a, st)
        torch.cuda.synchronize()
        print(f"  {task} {kind} {N}x{M}: {(time.perf_counter() - t0) / reps * 1e6:.1f} us per fused call")
        bufs["pw"].zero_(); bufs["pb"].zero_()
        _abi.particle_net_train_fused(a, st)
        torch.cuda.synchronize()

    dW, db = bufs["pw"].sum(1), bufs["pb"].sum(1)
    dz_first = bufs["dzf"].float() * bufs["scf"][:, None]
    dz_join = bufs["dzj"].float() * bufs["scj"][:, None]
    h_last = bufs["hl"].float()
    d_out_used = bufs["d_raw"] if kind == "dynamics" else d_out_ref
    got = {"d_states": bufs["d_states"], "d_traj_bias": dz_join.view(N, M, U).sum(1)}
    grads = [None] * len(params)
    grads[0] = dz_first.t() @ st
    grads[1] = dz_first.sum(0)
    grads[2], grads[3], grads[4], grads[5] = dW[0], db[0], dW[1], db[1]
    gj = torch.zeros_like(params[6])
    gj[:, net.join_state_off:net.join_state_off + U] = dW[2]
    grads[6] = gj
    for i in range(net.n_res):
        for kk in range(2):
            layer = 3 + 2 * i + kk
            grads[7 + 4 * i + 2 * kk] = dW[layer]
            grads[8 + 4 * i + 2 * kk] = db[layer]
    grads[-2] = d_out_used.t() @ h_last
    grads[-1] = d_out_used.sum(0)
    names = ["d_states", "d_traj_bias"] + [f"param{i}{tuple(p.shape)}" for i, p in enumerate(params)]
    gots = [got["d_states"], got["d_traj_bias"]] + grads
    worst = 0.0
    if kind == "dynamics":
        err = float((bufs["d_raw"] - d_out_ref).abs().max()) / max(1e-30, float(d_out_ref.abs().max()))
        worst = max(worst, err)
        if verbose:
            print(f"    d_raw: {err:.2e}")
    for n, a_, b_ in zip(names, gots, want):
        scale = max(1e-30, float(b_.abs().max()))
        err = float((a_.double() - b_.double()).abs().max()) / scale
        bad = not torch.isfinite(a_).all()
        worst = max(worst, err if not bad else float("inf"))
        if verbose:
            print(f"    {n}: {err:.2e}{'  NON-FINITE' if bad else ''}")
    # per-row check of d_states (rows of small trajectories must keep their relative precision)
    rs = want[0].abs().amax(1).clamp_min(1e-30)
    row_err = float(((bufs['d_states'] - want[0]).abs().amax(1) / rs).max())
    print(f"  {task} {kind} {N}x{M}: worst tensor error {worst:.2e}, worst d_states ROW error {row_err:.2e}")
    return (worst, row_err) if return_rows else worst
