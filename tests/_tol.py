"""The parity bar as ``north_star`` states it: "posterior means/covariances within 1e-4 rel fp32".  ``rel_err`` is
element-wise relative -- every entry against ITS OWN magnitude -- with a floor of ``floor`` times the tensor's largest
entry (an entry that is 1000x smaller than the largest is compared on the largest's 1e-3 scale: below that an fp32
result of the same formula carries no more relative information).  Round 4's tests divided by ``max(1, |want|.max())``,
which for EKF covariances (<= 0.1) is an ABSOLUTE 1e-4 = 1e-3 of the largest entry."""
import torch

REL_TOL = 1e-4


def rel_err(got, want, floor: float = 1e-3) -> float:
    got = torch.as_tensor(got).detach().cpu().double()
    want = torch.as_tensor(want).detach().cpu().double()
    top = float(want.abs().max())
    if top == 0.0:
        return float(got.abs().max())
    denom = want.abs().clamp_min(floor * top)
    return float(((got - want).abs() / denom).max())
