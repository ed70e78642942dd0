"""Kernel-level parity on a real MI355X, through the C ABI (``multimodalfilter_amd._abi``):
K1 (reweight + resample) bit-exact on indices against ``oracle/resample.py``; K3 (EKF
algebra + fusion) and K2/K5 (per-particle networks, Jacobian) within 1e-4 relative of the
oracle's fp32 torch restatement (the tolerance ``north_star`` states).
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import models as om
from oracle import resample as rs

import _tol


def _cuda():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X (run with gpurun / -m gpu on the GPU box)")
    return torch.device("cuda:0")


def _abi():
    from multimodalfilter_amd import _abi

    _abi.load()
    return _abi


# ------------------------------------------------------------------------------ K1
def _k1(abi, ll, lw, x, u, mode, M_out=None, want_idx=True, soft_alpha=1.0):
    dev = _cuda()
    N, M, d = x.shape
    M_out = M if M_out is None else M_out
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    est = torch.empty((N, d), device=dev)
    lw_out = torch.empty((N, M_out), device=dev)
    if mode == 0:
        abi.pf_reweight_resample(t(ll), t(lw), t(x), None, est, None, lw_out, None, 0)
        return est.cpu().numpy(), None, lw_out.cpu().numpy(), None
    xo = torch.empty((N, M_out, d), device=dev)
    idx = torch.empty((N, M_out), dtype=torch.int32, device=dev) if want_idx else None
    abi.pf_reweight_resample(t(ll), t(lw), t(x), t(u), est, xo, lw_out, idx, mode, soft_alpha)
    torch.cuda.synchronize()
    return est.cpu().numpy(), xo.cpu().numpy(), lw_out.cpu().numpy(), None if idx is None else idx.cpu().numpy()


@pytest.mark.parametrize("N,M,d", [(1, 1, 3), (3, 7, 2), (4, 64, 3), (5, 300, 3), (2, 1000, 2),
                                   (8, 1024, 3), (3, 4096, 3), (2, 4099, 1), (2, 8192, 2),
                                   (1, 16384, 3), (2, 6000, 4)])
@pytest.mark.parametrize("mode", ["systematic", "multinomial"])
def test_k1_indices_bit_exact(N, M, d, mode):
    abi = _abi()
    rng = np.random.RandomState(N * 1000 + M + d)
    ll = (rng.standard_normal((N, M)) * 3).astype(np.float32)
    lw = np.log(rng.dirichlet(np.ones(M) * 0.5, N) + 1e-30).astype(np.float32)
    x = rng.standard_normal((N, M, d)).astype(np.float32)
    u = rng.uniform(0, 1, (N,) if mode == "systematic" else (N, M)).astype(np.float32)
    code = {"systematic": 1, "multinomial": 2}[mode]
    est, xo, lwo, idx = _k1(abi, ll, lw, x, u, code)
    w_est, w_x, w_lw, w_idx = rs.reweight_resample(ll, lw, x, u, mode)
    np.testing.assert_array_equal(idx, w_idx)            # bit-exact ancestors
    np.testing.assert_array_equal(xo, w_x)               # gathered particles are copies
    np.testing.assert_allclose(lwo, w_lw, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(est, w_est, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("N,M,d", [(8, 1024, 3), (32, 4096, 3), (5, 2048, 2), (3, 4096, 2), (100, 512, 3), (16, 3072, 3)])
def test_k1_cluster_of_workgroups_equals_one_workgroup_per_trajectory(N, M, d):
    """Round 6 ("for N < 256 split M across a WG-cluster"): with few trajectories K1 gives each one a CLUSTER of
    workgroups that meet twice through L2 (row maximum; integer totals + the waves' float partial sums).  Same particle
    ownership, same per-thread chains, same per-wave trees, the partial sums added in wave order: ancestors, gathered
    particles AND estimates carry the bits of the one-workgroup kernel -- which the same trajectories get when they sit in a
    batch of more than half the CU count (or with the cluster switched off, as by default) -- and the ancestors equal the oracle's.  Hard cases inside: all the weight in one
    slice (the other workgroups of the cluster write nothing), ``-inf`` log-likelihoods, ``u`` at 0 and at 1 - ulp."""
    abi = _abi()
    was = abi.pf_set_resample_cluster(True)   # off by default (no faster: profiles/r06/k1_cluster_ab.txt)
    rng = np.random.RandomState(N * 31 + M + d)
    ll = (rng.standard_normal((N, M)) * 3).astype(np.float32)
    lw = np.log(rng.dirichlet(np.ones(M) * 0.5, N) + 1e-30).astype(np.float32)
    x = rng.standard_normal((N, M, d)).astype(np.float32)
    u = rng.uniform(0, 1, (N,)).astype(np.float32)
    ll[0, : M // 2] = -np.inf                      # the first half of the slices carries nothing
    ll[1 % N, :] = -60.0
    ll[1 % N, M - 3] = 0.0                         # (almost) all the weight on one particle of the last slice
    ll[2 % N, rng.randint(0, M, 17)] = -np.inf
    u[0], u[1 % N] = 0.0, np.float32(1.0) - np.float32(2.0 ** -24)
    try:
        est, xo, lwo, idx = _k1(abi, ll, lw, x, u, 1)
    finally:
        abi.pf_set_resample_cluster(was)
    # the same trajectories inside a large batch: one workgroup each (the cluster is for N <= half the CUs)
    reps = -(-200 // N)
    big = lambda a: np.concatenate([a] * reps, axis=0)
    est1, xo1, lwo1, idx1 = _k1(abi, big(ll), big(lw), big(x), big(u), 1)
    np.testing.assert_array_equal(idx, idx1[:N])
    np.testing.assert_array_equal(xo, xo1[:N])
    np.testing.assert_array_equal(est, est1[:N])     # every bit of the estimate
    np.testing.assert_array_equal(lwo, lwo1[:N])
    w_est, w_x, w_lw, w_idx = rs.reweight_resample(ll, lw, x, u, "systematic")
    np.testing.assert_array_equal(idx, w_idx)
    np.testing.assert_array_equal(xo, w_x)


@pytest.mark.parametrize("N,M,Mo,d", [(1, 1, 1, 3), (3, 7, 7, 2), (4, 30, 300, 3), (5, 300, 30, 3), (2, 1000, 1000, 2),
                                      (3, 4096, 4096, 3), (2, 4099, 4099, 1), (2, 8192, 8192, 2), (1, 16384, 16384, 3)])
@pytest.mark.parametrize("mode", ["systematic", "multinomial"])
@pytest.mark.parametrize("alpha", [0.5, 0.9])
def test_k1_soft_resampling_matches_oracle(N, M, Mo, d, mode, alpha):
    """``soft_resample_alpha`` < 1 (``mmf_pf_reweight_resample_soft``): ancestors drawn from the
    fixed-point mixture are bit-exact against the oracle, the survivors' importance weights agree to
    1e-5 and are normalised; ``-inf`` log-likelihoods can only be reached through the uniform part."""
    abi = _abi()
    rng = np.random.RandomState(N * 977 + M + 13 * d + int(100 * alpha))
    ll = (rng.standard_normal((N, M)) * 3).astype(np.float32)
    if M >= 30:
        ll[:, rng.randint(0, M, 3)] = -np.inf
    lw = np.log(rng.dirichlet(np.ones(M) * 0.5, N) + 1e-30).astype(np.float32)
    x = rng.standard_normal((N, M, d)).astype(np.float32)
    u = rng.uniform(0, 1, (N,) if mode == "systematic" else (N, Mo)).astype(np.float32)
    code = {"systematic": 1, "multinomial": 2}[mode]
    est, xo, lwo, idx = _k1(abi, ll, lw, x, u, code, M_out=Mo, soft_alpha=alpha)
    w_est, w_x, w_lw, w_idx = rs.reweight_resample(ll, lw, x, u, mode, Mo, soft_alpha=alpha)
    np.testing.assert_array_equal(idx, w_idx)
    np.testing.assert_array_equal(xo, w_x)
    fin = np.isfinite(w_lw)
    assert np.array_equal(fin, np.isfinite(lwo))
    np.testing.assert_allclose(lwo[fin], w_lw[fin], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(est, w_est, rtol=1e-4, atol=1e-5)
    assert abs(np.exp(lwo.astype(np.float64)).sum(1) - 1).max() < 1e-4
    if M > 1:  # and it is a different draw from the plain resampler's
        assert not np.array_equal(idx, rs.resample_indices(ll + lw, u, mode, Mo)) or M < 30


@pytest.mark.parametrize("N,M,d", [(1, 1, 2), (4, 30, 3), (3, 300, 3), (2, 4096, 2), (2, 5001, 3), (1, 32768, 3)])
def test_k1_no_resample_mode(N, M, d):
    abi = _abi()
    rng = np.random.RandomState(M)
    ll = (rng.standard_normal((N, M)) * 2).astype(np.float32)
    lw = np.log(rng.dirichlet(np.ones(M), N) + 1e-30).astype(np.float32)
    x = rng.standard_normal((N, M, d)).astype(np.float32)
    est, _, lwo, _ = _k1(abi, ll, lw, x, None, 0)
    w_est, _, w_lw, _ = rs.reweight_resample(ll, lw, x, None, "none")
    np.testing.assert_allclose(lwo, w_lw, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(est, w_est, rtol=1e-4, atol=1e-5)
    assert abs(np.exp(lwo.astype(np.float64)).sum(1) - 1).max() < 1e-4


def test_k1_edge_cases_changing_count_neg_inf_and_degenerate():
    abi = _abi()
    rng = np.random.RandomState(0)
    # particle count changes at the resampling step (eval() flips 30 -> 300 in the reference)
    N, M, Mo, d = 3, 30, 300, 3
    ll = rng.standard_normal((N, M)).astype(np.float32)
    lw = np.full((N, M), -math.log(M), np.float32)
    x = rng.standard_normal((N, M, d)).astype(np.float32)
    for mode, u in ((1, rng.uniform(0, 1, N)), (2, rng.uniform(0, 1, (N, Mo)))):
        u = u.astype(np.float32)
        est, xo, lwo, idx = _k1(abi, ll, lw, x, u, mode, M_out=Mo)
        name = {1: "systematic", 2: "multinomial"}[mode]
        np.testing.assert_array_equal(idx, rs.resample_indices(ll + lw, u, name, Mo))
        np.testing.assert_allclose(lwo, -math.log(Mo), rtol=1e-6)
    # -inf log-likelihoods (blacked-out modality) and a single surviving particle
    ll = np.full((2, 64), -np.inf, np.float32)
    ll[0, 5] = 0.0
    ll[1, [3, 60]] = [-1.0, -2.0]
    lw = np.zeros((2, 64), np.float32)
    x = rng.standard_normal((2, 64, 2)).astype(np.float32)
    u = np.array([0.25, 0.75], np.float32)
    est, xo, lwo, idx = _k1(abi, ll, lw, x, u, 1)
    np.testing.assert_array_equal(idx, rs.resample_indices(ll, u, "systematic"))
    assert set(idx[0]) == {5} and set(idx[1]) <= {3, 60}
    np.testing.assert_allclose(est[0], x[0, 5], rtol=1e-6)
    # u at the ends of [0, 1)
    ll = rng.standard_normal((2, 128)).astype(np.float32)
    lw = np.zeros((2, 128), np.float32)
    x = rng.standard_normal((2, 128, 3)).astype(np.float32)
    u = np.array([0.0, np.nextafter(np.float32(1), np.float32(0))], np.float32)
    _, _, _, idx = _k1(abi, ll, lw, x, u, 1)
    np.testing.assert_array_equal(idx, rs.resample_indices(ll, u, "systematic"))


@pytest.mark.parametrize("mode", [1, 2])
def test_k1_uniform_weight_shortcuts(mode):
    """Null ``logw_in`` = uniform ``-log M`` and null ``logw_out`` = nothing written (what the native
    step loop passes from its second step on): same ancestors, particles and estimate, bit for bit, as
    the call with the materialised uniform weights; refused where the weights are not uniform."""
    abi = _abi()
    dev = _cuda()
    N, M, d = 5, 4096, 3
    g = torch.Generator().manual_seed(3)
    ll = (torch.randn((N, M), generator=g) * 2).to(dev)
    x = torch.randn((N, M, d), generator=g).to(dev)
    u = torch.rand((N,) if mode == 1 else (N, M), generator=g).to(dev)
    lw = torch.full((N, M), 0.0, device=dev)
    # the uniform weights exactly as a resampling step leaves them
    abi.pf_reweight_resample(torch.zeros_like(ll), lw.clone(), x, torch.rand((N,), generator=g).to(dev),
                             torch.empty((N, d), device=dev), torch.empty_like(x), lw, None, 1)
    assert float((lw + math.log(M)).abs().max()) < 1e-6
    est0, xo0 = torch.empty((N, d), device=dev), torch.empty_like(x)
    lw0, idx0 = torch.empty_like(lw), torch.empty((N, M), dtype=torch.int32, device=dev)
    abi.pf_reweight_resample(ll, lw, x, u, est0, xo0, lw0, idx0, mode)
    est1, xo1, idx1 = torch.empty_like(est0), torch.empty_like(x), torch.empty_like(idx0)
    # the wrapper takes the output count from logw_out: hand it a (N, M) view that the library is told not to fill
    keep = torch.full((N, M), 7.0, device=dev)
    with abi._on(x):
        abi._check(abi.load().mmf_pf_reweight_resample(abi.ptr(ll), None, abi.ptr(x), abi.ptr(u), abi.ptr(est1), abi.ptr(xo1),
                                                     None, abi.ptr(idx1, dtype=torch.int32), N, M, M, d, mode,
                                                     abi.stream_of(x)), "mmf_pf_reweight_resample")
    torch.cuda.synchronize()
    assert torch.equal(idx0, idx1) and torch.equal(xo0, xo1) and torch.equal(est0, est1)
    assert float(keep.min()) == 7.0
    # mode 0 (weights are the output) and soft resampling (weights are not uniform) need both tensors
    assert abi.load().mmf_pf_reweight_resample(abi.ptr(ll), None, abi.ptr(x), None, abi.ptr(est1), None, abi.ptr(lw0), None,
                                              N, M, M, d, 0, abi.stream_of(x)) != 0
    assert abi.load().mmf_pf_reweight_resample_soft(abi.ptr(ll), abi.ptr(lw), abi.ptr(x), abi.ptr(u), abi.ptr(est1), abi.ptr(xo1),
                                                   None, None, N, M, M, d, mode, 0.5, abi.stream_of(x)) != 0


def test_k1_argument_errors():
    abi = _abi()
    dev = _cuda()
    x = torch.zeros((2, 8, 3), device=dev)
    ll = torch.zeros((2, 8), device=dev)
    est = torch.zeros((2, 3), device=dev)
    with pytest.raises(abi.MmfError):  # in-place gather is refused
        abi.pf_reweight_resample(ll, ll.clone(), x, torch.zeros(2, device=dev), est, x, ll.clone(), None, 1)
    with pytest.raises(abi.MmfError):  # CPU tensors never reach the kernel
        abi.pf_reweight_resample(ll.cpu(), ll.cpu(), x.cpu(), None, est.cpu(), None, ll.cpu(), None, 0)
    big = torch.zeros((1, 24000), device=dev)
    with pytest.raises(abi.MmfError):  # the 8-byte CDF would not fit the 160 KiB LDS
        abi.pf_reweight_resample(big, big, torch.zeros((1, 24000, 3), device=dev), torch.zeros(1, device=dev),
                                 torch.zeros((1, 3), device=dev), torch.zeros((1, 24000, 3), device=dev),
                                 big.clone(), None, 1)


def test_k1_full_size_properties():
    """BASELINE sizes (N=256, M=4096, d=3): idempotence on uniform weights, sorted ancestors,
    counts within one of M*w, determinism across launches."""
    abi = _abi()
    dev = _cuda()
    N, M, d = 256, 4096, 3
    g = torch.Generator(device="cpu").manual_seed(7)
    x = torch.randn((N, M, d), generator=g).to(dev)
    lw = torch.full((N, M), -math.log(M), device=dev)
    u = torch.rand((N,), generator=g).to(dev)
    est = torch.empty((N, d), device=dev)
    xo = torch.empty_like(x)
    lwo = torch.empty_like(lw)
    idx = torch.empty((N, M), dtype=torch.int32, device=dev)
    abi.pf_reweight_resample(torch.zeros_like(lw), lw, x, u, est, xo, lwo, idx, 1)
    assert torch.equal(idx, torch.arange(M, device=dev, dtype=torch.int32).expand(N, M))
    assert torch.equal(xo, x)
    torch.testing.assert_close(est, x.mean(1), rtol=1e-4, atol=1e-5)
    ll = (torch.randn((N, M), generator=g) * 2).to(dev)
    abi.pf_reweight_resample(ll, lw, x, u, est, xo, lwo, idx, 1)
    idx2 = torch.empty_like(idx)
    abi.pf_reweight_resample(ll, lw, x, u, est, torch.empty_like(x), lwo, idx2, 1)
    assert torch.equal(idx, idx2)
    assert bool((idx[:, 1:] >= idx[:, :-1]).all())
    w = torch.softmax(ll.double(), dim=1)
    counts = torch.zeros((N, M), dtype=torch.float64, device=dev).scatter_add_(
        1, idx.long(), torch.ones((N, M), dtype=torch.float64, device=dev))
    assert float((counts - M * w).abs().max()) < 1.0 + 1e-2
    torch.testing.assert_close(est.double(), (w[:, :, None] * x.double()).sum(1), rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------ K3
def _spd(rng, K, N, d):
    """Covariances with eigenvalues in [0.05, 0.5] (condition number <= 10, like a belief
    that started at 0.1 I)."""
    q, _ = np.linalg.qr(rng.standard_normal((K, N, d, d)))
    lam = rng.uniform(0.05, 0.5, (K, N, d))
    return np.einsum("knij,knj,knlj->knil", q, lam, q).astype(np.float32)


@pytest.mark.parametrize("d", [1, 2, 3, 4])
@pytest.mark.parametrize("K,fusion", [(1, 0), (2, 0), (2, 1), (2, 2), (3, 1), (3, 2)])
def test_k3_matches_oracle_algebra(d, K, fusion):
    abi = _abi()
    dev = _cuda()
    rng = np.random.RandomState(d * 10 + K + fusion)
    N = 37
    A = (np.eye(d, dtype=np.float32) + 0.2 * rng.standard_normal((K, N, d, d))).astype(np.float32)
    mu_pred = rng.standard_normal((K, N, d)).astype(np.float32)
    L = np.tril(0.2 * rng.standard_normal((K, d, d))).astype(np.float32)
    z = rng.standard_normal((K, N, d)).astype(np.float32)
    r = (np.tril(0.05 * rng.standard_normal((K, N, d, d)), -1)
         + rng.uniform(0.3, 0.6, (K, N, d, 1)) * np.eye(d)).astype(np.float32)
    S0 = _spd(rng, K, N, d)
    w = rng.uniform(0.05, 1.0, (K, N, d)).astype(np.float32)
    T = lambda a: torch.from_numpy(a).to(dev)
    mu = torch.empty((K, N, d), device=dev)
    Sigma = T(S0.copy())
    mu_f = torch.empty((N, d), device=dev)
    Sig_f = torch.empty((N, d, d), device=dev)
    abi.ekf_step(T(A), T(mu_pred), T(L), T(z), T(r), T(w) if fusion == 1 else None, mu, Sigma,
                 mu_f if fusion else None, Sig_f if fusion else None, fusion=fusion, feedback=0)
    # the oracle's algebra (oracle/tf/filters.py, oracle/models.py): once in fp32 (what the
    # oracle computes) and once in fp64 (ground truth).  The kernel must be within 1e-4 of
    # the truth, or -- where double inversion of fp32 data is itself less accurate than that
    # -- at least as accurate as the oracle's own fp32 evaluation (x3 slack).
    def algebra(dt):
        tA, tS, tL, tz, tr, tmp, tw = (torch.from_numpy(a).to(dt) for a in (A, S0, L, z, r, mu_pred, w))
        Sp = tA @ tS @ tA.transpose(-1, -2) + (tL @ tL.transpose(-1, -2))[:, None]
        R = tr @ tr.transpose(-1, -2)
        Kg = Sp @ torch.inverse(Sp + R)
        mu_w = tmp + (Kg @ (tz - tmp)[..., None]).squeeze(-1)
        S_w = (torch.eye(d, dtype=dt) - Kg) @ Sp
        f_mu = f_S = None
        if fusion == 1:
            f_mu, f_S = om._fuse_crossmodal(tw, mu_w, S_w)
        elif fusion == 2:
            prec = torch.inverse(S_w + 1e-9)
            f_S = torch.inverse(prec.sum(0) + 1e-9)
            f_mu = (f_S @ (prec @ mu_w[..., None]).sum(0)).squeeze(-1)
        return mu_w, S_w, f_mu, f_S

    truth, fp32 = algebra(torch.float64), algebra(torch.float32)

    def check(got, i):
        ref = truth[i]
        scale = max(1e-3, float(ref.abs().max()))
        err = float((got.cpu().double() - ref).abs().max()) / scale
        oracle_err = float((fp32[i].double() - ref).abs().max()) / scale
        assert err < max(1e-4, 3 * oracle_err), (i, err, oracle_err)

    check(mu, 0)
    check(Sigma, 1)
    if fusion:
        check(mu_f, 2)
        check(Sig_f, 3)
        # feedback=1 writes the fused belief into every sub-filter
        mu2 = torch.empty((K, N, d), device=dev)
        Sigma2 = T(S0.copy())
        abi.ekf_step(T(A), T(mu_pred), T(L), T(z), T(r), T(w) if fusion == 1 else None, mu2, Sigma2,
                     mu_f, Sig_f, fusion=fusion, feedback=1)
        for k in range(K):
            assert torch.equal(mu2[k], mu_f) and torch.equal(Sigma2[k], Sig_f)


# ------------------------------------------------------------------------------ K2 / K5
def _seeded(module, seed=0):
    module.load_state_dict(om.seeded_state_dict(module, seed=seed, gain=1.4))
    return module


def _rel_err(got, want):
    scale = max(1e-6, float(want.abs().max()))
    return float((got - want).abs().max()) / scale


@pytest.fixture()
def precision(request):
    from multimodalfilter_amd import engine

    old = engine.DEFAULT_PRECISION
    engine.set_default_precision(request.param)
    yield request.param
    engine.set_default_precision(old)


@pytest.mark.parametrize("precision", ["f32", "f16x3"], indirect=True)
@pytest.mark.parametrize("task", ["door", "push"])
@pytest.mark.parametrize("N,M", [(1, 1), (3, 5), (2, 64), (4, 300), (2, 4096), (40, 4096), (33, 4001)])
def test_k2_dynamics_and_measurement_match_oracle(task, N, M, precision):
    import multimodalfilter_amd as mmf

    dev = _cuda()
    spec = om.TASKS[task]
    d = spec.state_dim
    g = torch.Generator().manual_seed(N * 31 + M)
    x = torch.randn((N, M, d), generator=g)
    u = torch.randn((N, 7), generator=g)
    eps = torch.randn((N, M, d), generator=g)
    obs = {"image": torch.randn((N, 32, 32), generator=g).clamp(-1, 1),
           "gripper_pos": torch.randn((N, 3), generator=g),
           "gripper_sensors": torch.randn((N, 7), generator=g)}
    models = mmf.door_models if task == "door" else mmf.push_models
    P = "Door" if task == "door" else "Push"

    # dynamics (PF variant) with reparameterised noise
    o_dyn = _seeded(om.DynamicsModel(spec, brent_noise=spec.pf_noise_brent))
    p_dyn = getattr(models, f"{P}DynamicsModelBrent" if task == "door" else f"{P}DynamicsModel")()
    p_dyn.load_state_dict(o_dyn.state_dict())
    p_dyn.to(dev)
    with torch.no_grad():
        mean, tril = o_dyn(initial_states=x.reshape(N * M, d), controls=u.repeat_interleave(M, 0))
        want = (mean + torch.einsum("rij,rj->ri", tril, eps.reshape(N * M, d))).reshape(N, M, d)
    got = p_dyn.propagate_encoded(x.to(dev), p_dyn.encode_controls(u.to(dev)), eps.to(dev)).cpu()
    assert _rel_err(got, want) < 1e-4
    got_mean, got_tril = p_dyn(initial_states=x.reshape(N * M, d).to(dev),
                               controls=u.repeat_interleave(M, 0).to(dev))
    assert _rel_err(got_mean.cpu(), mean) < 1e-4
    torch.testing.assert_close(got_tril.cpu(), tril)

    # crossmodal measurement model = two unimodal nets + modality logsumexp
    o_pf = _seeded(om.ParticleFilter(spec, "crossmodal"))
    p_pf = getattr(models, f"{P}CrossmodalParticleFilter")()
    p_pf.load_state_dict(o_pf.state_dict())
    p_pf.to(dev)
    dobs = {k: v.to(dev) for k, v in obs.items()}
    with torch.no_grad():
        want = o_pf.measurement_model(states=x, observations=obs)
    got = p_pf.measurement_model(states=x.to(dev), observations=dobs).cpu()
    assert _tol.rel_err(got, want, dims=1) < 1e-4, _tol.rel_err(got, want, dims=1)   # every trajectory's (M,) log-likelihood row
    for mask in ([True, False], [False, True]):
        o_pf.measurement_model.enabled_models = mask
        p_pf.measurement_model.enabled_models = mask
        with torch.no_grad():
            want = o_pf.measurement_model(states=x, observations=obs)
        got = p_pf.measurement_model(states=x.to(dev), observations=dobs).cpu()
        assert _tol.rel_err(got, want, dims=1) < 1e-4, (mask, _tol.rel_err(got, want, dims=1))


@pytest.mark.parametrize("task", ["door", "push"])
@pytest.mark.parametrize("N", [1, 5, 64, 1000])
def test_k5_jacobian_matches_autograd(task, N):
    import multimodalfilter_amd as mmf

    dev = _cuda()
    spec = om.TASKS[task]
    d = spec.state_dim
    g = torch.Generator().manual_seed(N)
    x = torch.randn((N, d), generator=g)
    u = torch.randn((N, 7), generator=g)
    o_dyn = _seeded(om.DynamicsModel(spec))
    models = mmf.door_models if task == "door" else mmf.push_models
    p_dyn = getattr(models, f"{spec.name.capitalize()}DynamicsModel")()
    p_dyn.load_state_dict(o_dyn.state_dict())
    p_dyn.to(dev)
    want = o_dyn.jacobian(initial_states=x, controls=u).detach()
    with torch.no_grad():
        want_x = o_dyn(initial_states=x, controls=u)[0]
    got = p_dyn.jacobian(initial_states=x.to(dev), controls=u.to(dev)).cpu()
    mu, A, L = p_dyn.predict_with_jacobian(x.to(dev), p_dyn.encode_controls(u.to(dev)))
    assert _rel_err(got, want) < 1e-4
    assert _rel_err(A.cpu(), want) < 1e-4
    assert _rel_err(mu.cpu(), want_x) < 1e-4


# ------------------------------------------------------------------------------ K4
@pytest.mark.parametrize("N", [1, 3, 16, 50, 256])
@pytest.mark.parametrize("nets", [1, 2, 3])
def test_k4_image_encoders_match_oracle(N, nets):
    """Batched image encoders (shared images, different weights) vs the oracle's torch stack."""
    from multimodalfilter_amd import engine, layers

    dev = _cuda()
    g = torch.Generator().manual_seed(N + 100 * nets)
    img = (torch.randn((N, 32, 32), generator=g) * 0.5).clamp(-1, 1)
    img[N // 2] = 0.0  # a blacked-out frame
    oracles = [_seeded(om.image_encoder(64), seed=k) for k in range(nets)]
    encs = []
    for o in oracles:
        e = layers.image_encoder(64)
        e.load_state_dict(o.state_dict())
        encs.append(e.to(dev))
    got = engine.encode_images(encs, img.to(dev))
    for o, gk in zip(oracles, got):
        with torch.no_grad():
            want = o(img[:, None])
        assert _rel_err(gk.cpu(), want) < 1e-4
    # weights changed in place (optimiser step / load_state_dict) -> blob is re-packed
    with torch.no_grad():
        encs[0][7].weight.mul_(0.5)
        oracles[0][7].weight.mul_(0.5)
        want = oracles[0](img[:, None])
    assert _rel_err(engine.encode_images(encs[:1], img.to(dev))[0].cpu(), want) < 1e-4


@pytest.mark.parametrize("N,nets", [(1, 1), (37, 2), (300, 3)])
def test_k4_bf16_mode_is_a_reduced_precision_twin(N, nets):
    """MMF_PREC_BF16 (BASELINE config 5: "bf16 measurement CNN on MFMA"): the two fused
    convolution kernels with ONE bf16 product per MAC and fp32 accumulation.  Stated tolerance:
    3e-2 relative on the 64 features (bf16 keeps 8 significant bits; five layers deep), and the
    mode must differ from the default arithmetic (it is really running)."""
    from multimodalfilter_amd import engine, layers

    dev = _cuda()
    g = torch.Generator().manual_seed(900 + N)
    img = (torch.randn((N, 32, 32), generator=g) * 0.5).clamp(-1, 1)
    oracles = [_seeded(om.image_encoder(64), seed=40 + k) for k in range(nets)]
    encs = []
    for o in oracles:
        e = layers.image_encoder(64)
        e.load_state_dict(o.state_dict())
        encs.append(e.to(dev))
    exact = engine.encode_images(encs, img.to(dev))
    engine.set_image_encoder_precision("bf16")
    try:
        got = engine.encode_images(encs, img.to(dev))
    finally:
        engine.set_image_encoder_precision(None)
    for o, gk, ek in zip(oracles, got, exact):
        with torch.no_grad():
            want = o(img[:, None])
        err = _rel_err(gk.cpu(), want)
        assert err < 3e-2, err
        assert _rel_err(ek.cpu(), want) < 1e-4
        assert not torch.equal(gk, ek)


@pytest.mark.parametrize("N", [1, 5, 64])
def test_k4_spanning_pool_variant_and_mixed_batches(N):
    """The push virtual sensor's stack (16->2 convolution, full-height / full-width average
    pools, Linear 64->64; push_models/layers.py:43-65,77-90) runs in K4 too; a call mixing
    both architectures returns each encoder's own features, and no torch module is called."""
    from multimodalfilter_amd import engine, layers

    dev = _cuda()
    g = torch.Generator().manual_seed(7 + N)
    img = (torch.randn((N, 32, 32), generator=g) * 0.5).clamp(-1, 1)
    spans = [True, False, True]
    oracles = [_seeded(om.image_encoder(64, sp), seed=20 + k) for k, sp in enumerate(spans)]
    encs = []
    for o, sp in zip(oracles, spans):
        e = layers.image_encoder(64, sp)
        e.load_state_dict(o.state_dict())
        e.forward = None  # the torch module must not be used
        encs.append(e.to(dev))
    got = engine.encode_images(encs, img.to(dev))
    for o, gk in zip(oracles, got):
        with torch.no_grad():
            want = o(img[:, None])
        assert gk.shape == want.shape
        assert _rel_err(gk.cpu(), want) < 1e-4


@pytest.mark.parametrize("task", ["door", "push"])
def test_k2_f16x3_error_against_fp64(task):
    """The split-f16 path vs the f32-MFMA path, both measured against an fp64 evaluation of
    the same network: f16x3 must stay within 1e-5 relative (an order below the 1e-4 bar)."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = _cuda()
    spec = om.TASKS[task]
    d, N, M = spec.state_dim, 8, 2048
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, M, d), generator=g) * 2.0
    u = torch.randn((N, 7), generator=g)
    o_dyn = _seeded(om.DynamicsModel(spec, brent_noise=spec.pf_noise_brent))
    ref64 = om.DynamicsModel(spec, brent_noise=spec.pf_noise_brent).double()
    ref64.load_state_dict({k: v.double() for k, v in o_dyn.state_dict().items()})
    with torch.no_grad():
        truth = ref64(initial_states=x.reshape(N * M, d).double(), controls=u.repeat_interleave(M, 0).double())[0]
    models = mmf.door_models if task == "door" else mmf.push_models
    p_dyn = getattr(models, "DoorDynamicsModelBrent" if task == "door" else "PushDynamicsModel")()
    p_dyn.load_state_dict(o_dyn.state_dict())
    p_dyn.to(dev)
    errs = {}
    old = engine.DEFAULT_PRECISION
    try:
        for prec in ("f32", "f16x3"):
            engine.set_default_precision(prec)
            got = p_dyn.propagate_encoded(x.to(dev), p_dyn.encode_controls(u.to(dev)), None)
            errs[prec] = float((got.cpu().double().reshape(N * M, d) - truth).abs().max() / truth.abs().max())
    finally:
        engine.set_default_precision(old)
    assert errs["f32"] < 1e-5 and errs["f16x3"] < 1e-5, errs


def test_k2_f16x3_range_flag_reports_saturated_split():
    """Activations beyond the f16 range make the two-half split inexact: the kernel must say so."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi, engine

    dev = _cuda()
    dyn = mmf.door_models.DoorDynamicsModelBrent()
    _seeded(dyn)
    with torch.no_grad():
        dyn.state_layers[0].weight.mul_(1e5)  # first-layer outputs ~1e5 > 65504
    dyn.to(dev)
    x = torch.randn((2, 64, 3), device=dev)
    ctx = dyn.encode_controls(torch.randn((2, 7), device=dev))
    old = engine.DEFAULT_PRECISION
    try:
        engine.set_default_precision("f16x3")
        engine.check_range(dev)  # clear
        dyn.propagate_encoded(x, ctx, None)
        with pytest.raises(_abi.MmfError, match="f16x3 operand range"):
            engine.check_range(dev)
        engine.check_range(dev)  # flag was reset
        engine.set_default_precision("f32")
        dyn.propagate_encoded(x, ctx, None)
        engine.check_range(dev)  # f32 path never raises it
    finally:
        engine.set_default_precision(old)


# ------------------------------------------------------------------------------ K7
@pytest.mark.parametrize("R", [1, 7, 8, 100, 4096])
def test_k7_traj_program_matches_torch_modules(R):
    """A hand-built program (encoders -> concat linear -> wide residual blocks -> heads with
    sigmoid / sqrt(x^2+eps) / diagonal stores) against the same nn.Modules in torch (CPU)."""
    from multimodalfilter_amd import _abi, layers
    from multimodalfilter_amd.trajprog import TrajProgram

    dev = _cuda()
    torch.manual_seed(R)
    enc_a, enc_b = layers.vector_encoder(3, 64), layers.vector_encoder(7, 64)
    fuse = torch.nn.Linear(64 * 3, 128)
    wide = layers.ResLinear(128)
    head_z = torch.nn.Sequential(torch.nn.Linear(64, 3), torch.nn.ReLU(), layers.ResLinear(3), torch.nn.Linear(3, 3))
    head_w = torch.nn.Linear(128, 6)
    a, b, f = torch.randn(R, 3), torch.randn(R, 7), torch.randn(R, 64)
    with torch.no_grad():
        x = torch.relu(fuse(torch.cat([f, enc_a(a), enc_b(b)], 1)))
        x = wide(x)
        want_z = head_z(x[:, 64:])
        want_w = torch.sigmoid(head_w(x))
        want_d = torch.diag_embed(torch.sqrt(want_z ** 2 + 1e-6))

    p = TrajProgram()
    sf = p.load("f", 64)
    ra = p.load("a", 3); ea = p.vector_encoder(enc_a, ra, 3); p.free(ra)
    rb = p.load("b", 7); eb = p.vector_encoder(enc_b, rb, 7); p.free(rb)
    x = p.linear([(sf, 0, 64), (ea, 0, 64), (eb, 0, 64)], fuse, _abi.ACT_RELU)
    p.res_linear(wide, x, 128)
    z = p.linear([(x, 64, 64)], head_z[0], _abi.ACT_RELU)
    p.res_linear(head_z[2], z, 3)
    zo = p.linear([(z, 0, 3)], head_z[3])
    p.store("z", zo, 3)
    p.store("d", zo, 3, diag=True, act=_abi.ACT_SQRT_SQ_PLUS, fparam=1e-6)
    p.store("w", p.linear([(x, 0, 128)], head_w, _abi.ACT_SIGMOID), 6)
    t = {"f": f.to(dev), "a": a.to(dev), "b": b.to(dev),
         "z": torch.empty((R, 3), device=dev), "d": torch.empty((R, 3, 3), device=dev),
         "w": torch.empty((R, 6), device=dev)}
    p.run(t, R)
    assert _rel_err(t["z"].cpu(), want_z) < 1e-5
    assert _rel_err(t["w"].cpu(), want_w) < 1e-5
    assert _rel_err(t["d"].cpu(), want_d) < 1e-5
    # parameters updated in place -> the weight blob is rebuilt
    with torch.no_grad():
        head_w.weight.mul_(-1.0)
        want_w = torch.sigmoid(head_w(x_ref := wide(torch.relu(fuse(torch.cat([f, enc_a(a), enc_b(b)], 1))))))
    p.run(t, R)
    assert _rel_err(t["w"].cpu(), want_w) < 1e-5


def test_k4_k7_rows_do_not_depend_on_batch_position():
    """A trajectory's encoder outputs are the same bits whether it is evaluated alone, in a
    batch of N or in the T*N batch ``forward_loop`` builds (the step-by-step and the batched
    paths of the filters must agree exactly)."""
    from multimodalfilter_amd import _abi, engine, layers
    from multimodalfilter_amd.trajprog import TrajProgram

    dev = _cuda()
    torch.manual_seed(0)
    enc = layers.image_encoder(64).to(dev)
    imgs = (torch.randn(37, 32, 32, device=dev) * 0.5).clamp(-1, 1)
    big = engine.encode_images([enc], imgs)[0]
    for lo, hi in [(0, 1), (5, 10), (17, 37), (36, 37)]:
        assert torch.equal(engine.encode_images([enc], imgs[lo:hi].contiguous())[0], big[lo:hi])

    ve, wide, head = layers.vector_encoder(7, 64), layers.ResLinear(128), torch.nn.Linear(128, 5)
    up = torch.nn.Linear(64, 128)
    p = TrajProgram()
    s = p.load("x", 7)
    h = p.vector_encoder(ve, s, 7)
    x = p.linear([(h, 0, 64)], up, _abi.ACT_RELU)
    p.res_linear(wide, x, 128)
    p.store("y", p.linear([(x, 0, 128)], head, _abi.ACT_SIGMOID), 5)
    for m in (ve, wide, head, up):
        m.to(dev)
    xs = torch.randn(37, 7, device=dev)

    def run(rows):
        y = torch.empty((rows.shape[0], 5), device=dev)
        p.run({"x": rows.contiguous(), "y": y}, rows.shape[0])
        return y

    big = run(xs)
    for r in range(37):
        assert torch.equal(run(xs[r:r + 1]), big[r:r + 1]), r


@pytest.mark.parametrize("d", [2, 3])
@pytest.mark.parametrize("N,M", [(1, 1), (5, 300), (3, 4096)])
def test_pf_init_particles_matches_cholesky_sampling(N, M, d):
    """``mmf_pf_init_particles`` against ``mean + cholesky(cov) eps`` (torch, CPU fp32) on full
    (non-diagonal) covariances; uniform log-weights; a non-PD covariance raises."""
    import math

    import multimodalfilter_amd as mmf

    dev = _cuda()
    g = torch.Generator().manual_seed(N * 7 + M + d)
    A = torch.randn((N, d, d), generator=g)
    cov = A @ A.transpose(-1, -2) + 0.1 * torch.eye(d)
    mean = torch.randn((N, d), generator=g)
    eps = torch.randn((N, M, d), generator=g)
    want = mean[:, None, :] + torch.einsum("nij,nmj->nmi", torch.linalg.cholesky(cov), eps)
    f = (mmf.door_models.DoorParticleFilter() if d == 3 else mmf.push_models.PushParticleFilter()).to(dev).eval()
    f.num_particles = M
    f.noise = mmf.ReplayNoise([eps], [])
    f.initialize_beliefs(mean=mean.to(dev), covariance=cov.to(dev))
    assert _rel_err(f.particle_states.cpu(), want) < 1e-5
    assert torch.equal(f.particle_log_weights.cpu(), torch.full((N, M), -math.log(M), dtype=torch.float32))
    bad = cov.clone()
    bad[0] = -torch.eye(d)
    f.noise = mmf.ReplayNoise([eps], [])
    with pytest.raises(ValueError):
        f.initialize_beliefs(mean=mean.to(dev), covariance=bad.to(dev))


@pytest.mark.parametrize("task", ["door", "push"])
def test_k2_full_size_cross_checks(task):
    """BASELINE.json's full size (256 trajectories x 4096 particles = 1M rows), where the oracle
    would take minutes: the f16x3 kernels (64-particle tiles, pipelined halves) against the exact
    f32-MFMA kernels on the same inputs (two independent code paths, 1e-4 relative), bit-for-bit
    repeatability, and no range flag."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = _cuda()
    N, M = 256, 4096
    d = om.TASKS[task].state_dim
    g = torch.Generator().manual_seed(99)
    x = torch.randn((N, M, d), generator=g).to(dev)
    u = torch.randn((N, 7), generator=g).to(dev)
    eps = torch.randn((N, M, d), generator=g).to(dev)
    obs = {"image": torch.randn((N, 32, 32), generator=g).clamp(-1, 1).to(dev),
           "gripper_pos": torch.randn((N, 3), generator=g).to(dev),
           "gripper_sensors": torch.randn((N, 7), generator=g).to(dev)}
    torch.manual_seed(5)
    pf = mmf.model_types(task)[f"{task.capitalize()}CrossmodalParticleFilter"]().to(dev).eval()
    dyn, meas = pf.dynamics_model, pf.measurement_model
    old = engine.DEFAULT_PRECISION
    out = {}
    try:
        for prec in ("f16x3", "f32"):
            engine.set_default_precision(prec)
            ctx = dyn.encode_controls(u)
            nxt = dyn.propagate_encoded(x, ctx, eps)
            ll = meas(states=x, observations=obs)
            out[prec] = (nxt.clone(), ll.clone())
            if prec == "f16x3":
                again = dyn.propagate_encoded(x, ctx, eps)
                assert torch.equal(again, nxt)
                assert torch.equal(meas(states=x, observations=obs), ll)
                engine.check_range(dev)
    finally:
        engine.set_default_precision(old)
    for a, b in zip(out["f16x3"], out["f32"]):
        assert bool(torch.isfinite(a).all())
        assert _rel_err(a.cpu(), b.cpu()) < 1e-4


@pytest.mark.parametrize("layer,what", [(0, "stem output (A / B planes)"), (2, "ResConv output (C rows, X waves)"),
                                        (3, "conv 32->16 output (D rows, Y waves: the ring of the fused conv 16->8)"),
                                        (5, "conv 16->8 output (E, split by the linear layer)")])
def test_k4_f16x3_range_flag_reports_saturated_split(layer, what):
    """Every place the fused image encoder splits an activation into two f16 halves tracks its range: an activation
    beyond 65504 must raise the engine's range flag (``engine.check_range``), whichever layer produces it; the bf16
    mode (fp32 exponent range) and the exact-f32 per-layer path never do."""
    from multimodalfilter_amd import _abi, engine, layers

    dev = _cuda()
    torch.manual_seed(layer)
    enc = layers.image_encoder(64)
    with torch.no_grad():  # a bias of 1e5 on one output channel of the layer
        target = enc[layer].block2 if layer == 2 else enc[layer]
        target.bias[0] = 1.0e5
    enc.to(dev)
    img = (torch.randn((5, 32, 32), device=dev) * 0.5).clamp(-1, 1)
    old = engine.DEFAULT_PRECISION
    try:
        engine.set_default_precision("f16x3")
        engine.check_range(dev)  # clear
        engine.encode_images([enc], img)
        with pytest.raises(_abi.MmfError, match="f16x3 operand range"):
            engine.check_range(dev)
        # bf16 mode: the 32-channel planes are bf16 (fp32 exponent range), but conv 16->8 and the linear layer stay
        # f16x3, so D and E are still split into f16 halves there
        engine.set_image_encoder_precision("bf16")
        engine.encode_images([enc], img)
        if layer in (0, 2):
            engine.check_range(dev)
        else:
            with pytest.raises(_abi.MmfError, match="f16x3 operand range"):
                engine.check_range(dev)
        engine.set_image_encoder_precision(None)
        engine.set_default_precision("f32")
        out = engine.encode_images([enc], img)[0]
        engine.check_range(dev)
        assert bool(torch.isfinite(out).all())
    finally:
        engine.set_image_encoder_precision(None)
        engine.set_default_precision(old)
