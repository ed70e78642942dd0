"""Normative fixed-point resampler (numpy).  TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Why fixed point: ``north_star`` asks for bit-exact resample indices under a fixed seed,
but a floating-point prefix sum rounds differently for every scan tree, and libm ``exp``
differs from the GPU's by an ulp.  So the *definition* of the resampler is integer:

1. ``x_i = logw_i - max_j logw_j``                         (one IEEE fp32 subtract)
2. ``e_i = detexp(x_i)``                                   (fp32 mul/add only, fixed order)
3. ``q_i = floor(e_i * 2**24)`` as u64, ``Q = sum q_i``    (integer => order independent)
4. ``cdf_i = q_0 + ... + q_i``                             (inclusive integer scan)
5. systematic (one uniform ``u`` per trajectory), ``U = floor(u * 2**24)``::

       p_k = (k*Q + ((U*Q) >> 24)) // M_out        k = 0 .. M_out-1

   multinomial (one uniform per output particle)::

       p_k = (U_k * Q) >> 24

6. ``index_k = #{i : cdf_i <= p_k}``  (first ``i`` whose inclusive CDF exceeds ``p_k``)

Soft resampling (upstream's ``soft_resample_alpha`` < 1, an option the reference leaves at 1.0): the
ancestors are drawn from the mixture ``alpha w_i + (1 - alpha) / M``; in fixed point, with
``A = floor(alpha * 2**24)``::

       q'_i = ((A * q_i << 8) + (2**24 - A) * ((Q << 8) // M)) >> 32     (<= 2**24: all bounds below hold;
                                                                           the mean weight Q / M keeps 8 extra bits)

replaces ``q_i`` in steps 4-6, and the survivors carry the importance weights
``w_idx / (alpha w_idx + (1 - alpha) / M)``, normalised, instead of ``1 / M_out``.

Every quantity fits u64 for ``M <= 65536`` (``Q <= 2**40``, ``U*Q < 2**64``,
``k*Q < 2**56``).  Upstream torchfilter resamples with
``Categorical(logits=logw).sample((M,))`` (multinomial, torch global RNG; SURVEY.md
A.2) which cannot be reproduced bit-for-bit by anything; mode ``"multinomial"`` here is
the same distribution driven by explicit uniforms, ``"systematic"`` is the low-variance
scheme ``north_star`` asks for.
"""
import numpy as np

_F = np.float32
LOG2E = _F(1.4426950408889634)
# Taylor coefficients of 2**f = exp(f ln 2), degree 6, highest first.
_POLY = [
    _F(0.00015403530393381608),
    _F(0.0013333558146428443),
    _F(0.009618129107628477),
    _F(0.05550410866482158),
    _F(0.2402265069591007),
    _F(0.6931471805599453),
    _F(1.0),
]
FIX_BITS = 24
MAX_PARTICLES = 1 << 16


def detexp(x: np.ndarray) -> np.ndarray:
    """Deterministic fp32 ``exp`` for ``x <= 0``: separate (un-fused) multiplies and adds
    in a fixed order, so numpy and the HIP kernel agree bit for bit."""
    x = np.asarray(x, dtype=_F)
    with np.errstate(invalid="ignore", over="ignore"):
        t = (x * LOG2E).astype(_F)
        t = np.maximum(t, _F(-126.0)).astype(_F)  # also maps -inf -> -126 (=> q == 0)
        n = np.rint(t).astype(_F)  # round-half-even, like v_rndne_f32
        f = (t - n).astype(_F)
        p = np.full_like(f, _POLY[0])
        for c in _POLY[1:]:
            p = (p * f).astype(_F)
            p = (p + c).astype(_F)
        scale = ((n.astype(np.int32) + 127) << 23).astype(np.int32).view(_F)
        return (p * scale).astype(_F)


def quantise(logw: np.ndarray):
    """``(N, M)`` fp32 log-weights -> (``q`` u64 ``(N, M)``, fp32 ``e`` ``(N, M)``, max ``(N,)``)."""
    logw = np.asarray(logw, dtype=_F)
    m = logw.max(axis=1, keepdims=True)
    e = detexp((logw - m).astype(_F))
    q = np.floor(e.astype(np.float64) * float(1 << FIX_BITS)).astype(np.uint64)
    return q, e, m[:, 0]


def _fix_uniform(u) -> np.ndarray:
    u = np.asarray(u, dtype=_F)
    assert np.all((u >= 0) & (u < 1)), "uniforms must lie in [0, 1)"
    return np.floor(u.astype(np.float64) * float(1 << FIX_BITS)).astype(np.uint64)


def soft_mixture(q: np.ndarray, alpha: float) -> np.ndarray:
    """Fixed-point weights of the soft-resampling mixture ``alpha w + (1 - alpha) / M``."""
    A = np.uint64(np.floor(np.float64(_F(alpha)) * float(1 << FIX_BITS)))
    one = np.uint64(1 << FIX_BITS)
    M = np.uint64(q.shape[1])
    Q = q.sum(axis=1, dtype=np.uint64)
    e8 = np.uint64(8)
    return (((A * q) << e8) + ((one - A) * ((Q << e8) // M))[:, None]) >> np.uint64(FIX_BITS + 8)


def resample_indices(logw: np.ndarray, u: np.ndarray, mode: str, num_out: int = None, soft_alpha: float = 1.0) -> np.ndarray:
    """Resampling ancestor indices, ``(N, num_out)`` int32.

    ``mode="systematic"``: ``u`` has shape ``(N,)``; ``mode="multinomial"``: ``(N, num_out)``.
    """
    logw = np.asarray(logw, dtype=_F)
    N, M = logw.shape
    num_out = M if num_out is None else int(num_out)
    assert 1 <= M <= MAX_PARTICLES and 1 <= num_out <= MAX_PARTICLES
    assert 0.0 < soft_alpha <= 1.0
    q, _, _ = quantise(logw)
    if soft_alpha < 1.0:
        q = soft_mixture(q, soft_alpha)
    cdf = np.cumsum(q, axis=1, dtype=np.uint64)
    Q = cdf[:, -1]
    U = _fix_uniform(u)
    sh = np.uint64(FIX_BITS)
    if mode == "systematic":
        assert U.shape == (N,)
        R = (U * Q) >> sh
        k = np.arange(num_out, dtype=np.uint64)[None, :]
        p = (k * Q[:, None] + R[:, None]) // np.uint64(num_out)
    elif mode == "multinomial":
        assert U.shape == (N, num_out)
        p = (U * Q[:, None]) >> sh
    else:
        raise ValueError(mode)
    idx = np.empty((N, num_out), dtype=np.int32)
    for n in range(N):
        idx[n] = np.searchsorted(cdf[n], p[n], side="right")
    assert idx.max(initial=0) < M
    return idx


def reweight_resample(loglik, logw, states, u, mode: str, num_out: int = None, soft_alpha: float = 1.0):
    """What kernel K1 (``mmf_pf_reweight_resample``) computes, restated on the CPU.

    Follows the post-measurement half of upstream ``ParticleFilter.forward`` (SURVEY.md
    A.2 / section 3.2): ``logw += loglik``; normalise; weighted-mean estimate; then, when
    ``mode != "none"``, resample ancestors, gather states and reset log-weights to
    ``-log(num_out)``.

    Returns ``(estimate (N,d), states_out, logw_out, indices or None)``.
    """
    loglik = np.asarray(loglik, dtype=_F)
    logw = np.asarray(logw, dtype=_F)
    states = np.asarray(states, dtype=_F)
    N, M, d = states.shape
    tot = (logw + loglik).astype(_F)
    _, e, m = quantise(tot)
    S = e.astype(np.float64).sum(axis=1)
    w = e.astype(np.float64) / S[:, None]
    estimate = (w[:, :, None] * states.astype(np.float64)).sum(axis=1).astype(_F)
    if mode == "none":
        logw_out = ((tot - m[:, None]).astype(np.float64) - np.log(S)[:, None]).astype(_F)
        return estimate, states.copy(), logw_out, None
    num_out = M if num_out is None else int(num_out)
    idx = resample_indices(tot, u, mode, num_out, soft_alpha)
    states_out = np.take_along_axis(states, idx[:, :, None].astype(np.int64), axis=1)
    if soft_alpha < 1.0:
        a = np.float64(_F(soft_alpha))
        ratio = w / (a * w + (1.0 - a) / M)                 # importance weight of every candidate
        r = np.take_along_axis(ratio, idx.astype(np.int64), axis=1)
        with np.errstate(divide="ignore"):
            logw_out = (np.log(r) - np.log(r.sum(axis=1, keepdims=True))).astype(_F)
    else:
        logw_out = np.full((N, num_out), -np.log(_F(num_out)), dtype=_F)
    return estimate, states_out, logw_out, idx


def certify_mismatches(logw_a, logw_b, u, idx_a, idx_b, num_out: int = None):
    """Certificate for systematic-resampling ancestors drawn from two sets of log-weights that
    agree only to rounding (``a``: the checker's, ``b``: the implementation under test; both
    ``(N, M)`` fp32, same uniforms ``u``; ``idx_a`` / ``idx_b`` the ancestors each side drew).

    With ``delta_i = q_i(b) - q_i(a)`` and ``D = sum_i |delta_i|`` (per trajectory), every inclusive
    CDF entry moves by at most ``D`` and every position ``p_k = (k Q + R) // M`` by at most
    ``D + 2``, in opposite-signed shares: ``(cdf_i - p_k)`` changes by
    ``(1 - k/M) sum_{j<=i} delta_j - (k/M) sum_{j>i} delta_j`` up to the two floors, so by at most
    ``D + 2`` in magnitude.  Hence side ``b`` may legitimately draw ancestor ``e != o`` at output
    ``k`` only if side ``a``'s position lies within ``D + 2`` of the CDF boundary it would have to
    cross: ``cdf_a[e-1] - p_k <= D + 2`` when ``e > o``, ``p_k - cdf_a[e] + 1 <= D + 2`` when
    ``e < o``.  A mismatch outside that band is NOT explained by the weight differences.

    Returns a dict: ``mismatches``, ``unexplained`` (must be 0), ``max_slack_used`` (largest
    distance / (D + 2) over the mismatches, <= 1 when all are explained), ``max_hop`` (largest
    number of positive-weight particles of side ``a`` between the two ancestors, 1 = neighbours),
    ``max_D_over_Q`` (L1 weight difference relative to the total: how tight the band is).
    """
    logw_a = np.asarray(logw_a, dtype=_F)
    logw_b = np.asarray(logw_b, dtype=_F)
    idx_a = np.asarray(idx_a).astype(np.int64)
    idx_b = np.asarray(idx_b).astype(np.int64)
    N, M = logw_a.shape
    num_out = M if num_out is None else int(num_out)
    qa = quantise(logw_a)[0].astype(np.int64)
    qb = quantise(logw_b)[0].astype(np.int64)
    cdf = np.cumsum(qa, axis=1)
    Q = cdf[:, -1]
    D = np.abs(qb - qa).sum(axis=1)
    U = _fix_uniform(u).astype(np.int64)
    R = (U * Q) >> FIX_BITS
    out = {"mismatches": 0, "unexplained": 0, "max_slack_used": 0.0, "max_hop": 0,
           "max_D_over_Q": float((D / np.maximum(Q, 1)).max())}
    pos_rank = np.cumsum(qa > 0, axis=1)  # number of positive-weight particles up to and including i
    for n in range(N):
        ks = np.nonzero(idx_a[n] != idx_b[n])[0]
        if ks.size == 0:
            continue
        p = (ks.astype(np.int64) * Q[n] + R[n]) // num_out
        o, e = idx_a[n, ks], idx_b[n, ks]
        up = e > o
        dist = np.where(up, cdf[n, np.maximum(e - 1, 0)] - p, p - cdf[n, e] + 1)
        band = int(D[n]) + 2
        # positive-weight particles of side a in (lo, hi]: 1 = the two ancestors are neighbours
        lo, hi = np.minimum(o, e), np.maximum(o, e)
        out["mismatches"] += int(ks.size)
        out["unexplained"] += int((dist > band).sum())
        out["max_slack_used"] = max(out["max_slack_used"], float((dist / band).max()))
        out["max_hop"] = max(out["max_hop"], int((pos_rank[n, hi] - pos_rank[n, lo]).max()))
    return out
