"""The parity bar as ``north_star`` states it: "posterior means/covariances within 1e-4 rel fp32".

Round 4's tests divided the largest absolute difference by ``max(1, |want|.max())``, which for EKF covariances (<= 0.1) is
an ABSOLUTE 1e-4 = 1e-3 of the largest entry.  Round 5 holds every covariance MATRIX and every mean VECTOR to 1e-4 of its
own Frobenius norm (``rel_err``: the worst matrix / vector of the batch decides).

The stricter element-wise form (``rel_err_elementwise``: every entry against max(|its own value|, 1e-3 x the tensor's
largest)) was tried on the whole suite first and is NOT met everywhere -- measured on MI355X, worst entry per test:
UKF covariance off-diagonals of 1e-5 beside diagonals of 1.5e-2: 2.1e-4 (absolute 2e-8 = one fp32 ulp of the
diagonal); EKF posterior means with components of 0.02 beside components of 0.8: 1.1e-4 .. 2.4e-4 (absolute 3e-6).
Entries a thousand times smaller than their neighbours are differences of O(1) fp32 quantities: no fp32 evaluation of
the same formula in another summation order reproduces them to 1e-4 of themselves -- the CPU oracle in fp64 against
itself in fp32 does not either (``bench.py``: ``parity_vs_oracle.oracle_self``).  The element-wise number is still what
``bench.py`` reports for the EKF legs (``max_rel_err_posterior_covariance_vs_oracle``), next to the norm-wise one."""
import torch

REL_TOL = 1e-4


def rel_err(got, want, dims: int = None) -> float:
    """Worst ``||got - want||_F / ||want||_F`` over the leading (batch) axes; the norm runs over the last ``dims`` axes
    (default: 2 for square trailing matrices, else 1)."""
    got = torch.as_tensor(got).detach().cpu().double()
    want = torch.as_tensor(want).detach().cpu().double()
    if dims is None:
        dims = 2 if want.dim() >= 2 and want.shape[-1] == want.shape[-2] else 1
    axes = tuple(range(want.dim() - dims, want.dim()))
    num = (got - want).pow(2).sum(axes).sqrt()
    den = want.pow(2).sum(axes).sqrt()
    top = float(den.max()) if den.numel() else 0.0
    if top == 0.0:
        return float(num.max()) if num.numel() else 0.0
    return float((num / den.clamp_min(1e-3 * top)).max())


def rel_err_elementwise(got, want, floor: float = 1e-3) -> float:
    got = torch.as_tensor(got).detach().cpu().double()
    want = torch.as_tensor(want).detach().cpu().double()
    top = float(want.abs().max())
    if top == 0.0:
        return float(got.abs().max())
    return float(((got - want).abs() / want.abs().clamp_min(floor * top)).max())


def rel_err_finite(got, want) -> float:
    """``rel_err`` over the entries where ``want`` is finite (log-likelihoods carry ``-inf`` for blacked-out modalities; the
    caller asserts that the infinities sit at the same places): masked entries count as equal."""
    import numpy as np
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if want.ndim == 0:
        return scalar_rel(got, want)
    fin = np.isfinite(want)
    return rel_err(torch.from_numpy(np.where(fin, got, 0.0)), torch.from_numpy(np.where(fin, want, 0.0)))


def scalar_rel(got, want) -> float:
    """|got - want| / |want| for two scalars (losses)."""
    got, want = float(got), float(want)
    return abs(got - want) / abs(want) if want != 0.0 else abs(got)


# Where the REFERENCE's own fp32 arithmetic is not accurate to 1e-4: the same formulas (oracle/, pinned to the golden vectors
# bit for bit) evaluated in fp64 differ from the reference's fp32 output by the number given -- the posterior covariance of the
# measurement-fused EKF after 3 steps is a difference of nearly equal matrices (``P - K H P``).  Two fp32 evaluations in
# different operation orders each sit about that far from the exact value, in unrelated directions: the engine is held to
# FOUR times that distance from the reference there (measured on MI355X: 1.3e-3 in round 5's K4 arithmetic, 3.1e-3 in round
# 6's -- the image features moved by 1e-7, the covariance by 2e-3 of itself); ``tests/test_oracle_golden.py::test_conditioning_exceptions_are_the_references_own_fp32_error``
# re-measures the numbers on the CPU.  ``None``: the covariance has collapsed to rounding noise (|entries| <= 6e-10 under a
# 0.1 I prior, fp64 says ~1e-17): there is no relative error to speak of, the engine is held to 1e-4 of the PRIOR's scale.
REFERENCE_FP32_GAP = {
    "filter_kf_meas_crossmodal/door/n1m1/belief_covariance": 1.38e-3,
    "filter_kf_meas_unimodal/door/n4m8/belief_covariance": None,
}
PRIOR_COVARIANCE_SCALE = 0.1
