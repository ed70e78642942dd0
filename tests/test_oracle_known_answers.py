"""Known answers for the restated recursion (T1/T2/T3) -- the reference holds no tests for
it (``torchfilter`` is an absent third-party dependency), so it is pinned analytically:
linear-Gaussian closed forms, Jacobian vs. finite differences, resampling invariants.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from oracle import resample as rs
from oracle import tf
from oracle.tf.base import NoiseSource, ReplayNoise


class LinearDynamics(tf.base.DynamicsModel):
    def __init__(self, A, B, Ltril):
        super().__init__(state_dim=A.shape[0])
        self.A, self.B, self.L = A, B, Ltril

    def forward(self, *, initial_states, controls):
        R, d = initial_states.shape
        return initial_states @ self.A.T + controls @ self.B.T, self.L[None].expand(R, d, d)


class DirectSensor(tf.base.VirtualSensorModel):
    def __init__(self, d, Rtril):
        super().__init__(state_dim=d)
        self.Rtril = Rtril

    def forward(self, *, observations):
        N = observations["z"].shape[0]
        return observations["z"], self.Rtril[None].expand(N, *self.Rtril.shape)


class GaussianLik(tf.base.ParticleFilterMeasurementModel):
    def __init__(self, d, R):
        super().__init__(state_dim=d)
        self.Rinv = torch.inverse(R)

    def forward(self, *, states, observations):
        e = observations["z"][:, None, :] - states
        return -0.5 * torch.einsum("nmi,ij,nmj->nm", e, self.Rinv, e)


def _system(d=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = torch.eye(d) * 0.9 + 0.05 * torch.randn(d, d, generator=g)
    B = 0.1 * torch.randn(d, 7, generator=g)
    L = torch.diag(torch.tensor([0.2, 0.1, 0.15][:d]))
    Rt = torch.diag(torch.tensor([0.3, 0.25, 0.2][:d]))
    return A, B, L, Rt


def _kalman_closed_form(A, B, L, Rt, mu, S, us, zs):
    Q, R = L @ L.T, Rt @ Rt.T
    out = []
    for u, z in zip(us, zs):
        mu = mu @ A.T + u @ B.T
        S = A @ S @ A.T + Q
        K = S @ torch.inverse(S + R)
        mu = mu + (z - mu) @ K.T
        S = (torch.eye(len(Q)) - K) @ S
        out.append((mu.clone(), S.clone()))
    return out


def test_ekf_equals_kalman_filter_on_linear_system():
    A, B, L, Rt = _system()
    N, T, d = 5, 6, 3
    g = torch.Generator().manual_seed(1)
    us = torch.randn(T, N, 7, generator=g)
    zs = torch.randn(T, N, d, generator=g)
    mu0 = torch.randn(N, d, generator=g)
    f = tf.filters.VirtualSensorExtendedKalmanFilter(
        dynamics_model=LinearDynamics(A, B, L), virtual_sensor_model=DirectSensor(d, Rt))
    f.initialize_beliefs(mean=mu0, covariance=(0.1 * torch.eye(d))[None].expand(N, d, d))
    est = f.forward_loop(observations={"z": zs}, controls=us)
    for n in range(N):
        want = _kalman_closed_form(A, B, L, Rt, mu0[n], 0.1 * torch.eye(d), us[:, n], zs[:, n])
        for t in range(T):
            torch.testing.assert_close(est[t, n], want[t][0], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(f._belief_covariance[n], want[-1][1], rtol=1e-5, atol=1e-6)


def test_default_jacobian_matches_finite_differences():
    torch.manual_seed(0)

    class Net(tf.base.DynamicsModel):
        def __init__(self):
            super().__init__(state_dim=3)
            self.f = nn.Sequential(nn.Linear(10, 16), nn.Tanh(), nn.Linear(16, 3))

        def forward(self, *, initial_states, controls):
            y = initial_states + self.f(torch.cat([initial_states, controls], -1))
            return y, torch.eye(3)[None].expand(len(y), 3, 3)

    m = Net().double()
    x = torch.randn(4, 3, dtype=torch.float64)
    u = torch.randn(4, 7, dtype=torch.float64)
    J = m.jacobian(initial_states=x, controls=u)
    h = 1e-6
    for j in range(3):
        dx = torch.zeros(3, dtype=torch.float64)
        dx[j] = h
        col = (m(initial_states=x + dx, controls=u)[0] - m(initial_states=x - dx, controls=u)[0]) / (2 * h)
        torch.testing.assert_close(J[:, :, j], col, rtol=1e-6, atol=1e-8)


def test_particle_filter_converges_to_kalman_filter():
    A, B, L, Rt = _system()
    N, T, d, M = 2, 4, 3, 16384
    g = torch.Generator().manual_seed(2)
    us = torch.randn(T, N, 7, generator=g)
    zs = 0.3 * torch.randn(T, N, d, generator=g)
    mu0 = 0.2 * torch.randn(N, d, generator=g)
    for mode in ("systematic", "multinomial"):
        pf = tf.filters.ParticleFilter(
            dynamics_model=LinearDynamics(A, B, L),
            measurement_model=GaussianLik(d, Rt @ Rt.T), num_particles=M, resample_mode=mode)
        pf.eval()
        pf.noise = NoiseSource(3)
        pf.initialize_beliefs(mean=mu0, covariance=(0.1 * torch.eye(d))[None].expand(N, d, d))
        est = pf.forward_loop(observations={"z": zs}, controls=us)
        for n in range(N):
            want = _kalman_closed_form(A, B, L, Rt, mu0[n], 0.1 * torch.eye(d), us[:, n], zs[:, n])
            for t in range(T):
                assert torch.max(torch.abs(est[t, n] - want[t][0])) < 0.02, (mode, n, t)


# ------------------------------------------------------------------ resampler invariants
def test_detexp_is_accurate_and_monotone():
    x = -np.abs(np.random.RandomState(0).standard_normal(20000) * 8).astype(np.float32)
    x[:3] = [0.0, -1e-7, -87.0]
    e = rs.detexp(x)
    ref = np.exp(x.astype(np.float64))
    big = ref > 1e-30
    rel = np.abs(e / np.where(big, ref, 1) - 1)
    # fp32 range reduction: error grows with |x| (one ulp of x*log2e); tight near the max
    assert np.max(rel[big]) < 4e-6 and np.max(rel[x > -1.0]) < 4e-7
    assert e[0] == 1.0 and np.all(e <= 1.0) and np.all(e >= 0)
    xs = np.sort(x)
    assert np.all(np.diff(rs.detexp(xs).astype(np.float64)) >= -1e-7 * rs.detexp(xs)[1:])


def test_systematic_uniform_weights_is_identity():
    N, M = 3, 257
    idx = rs.resample_indices(np.zeros((N, M), np.float32), np.array([0.0, 0.5, 0.999], np.float32), "systematic")
    np.testing.assert_array_equal(idx, np.tile(np.arange(M), (N, 1)))


def test_systematic_counts_within_one_of_expectation():
    rng = np.random.RandomState(4)
    N, M = 6, 1000
    logw = (rng.standard_normal((N, M)) * 2).astype(np.float32)
    u = rng.uniform(0, 1, N).astype(np.float32)
    idx = rs.resample_indices(logw, u, "systematic")
    q, _, _ = rs.quantise(logw)
    for n in range(N):
        counts = np.bincount(idx[n], minlength=M)
        expect = M * q[n].astype(np.float64) / q[n].sum()
        assert np.all(np.abs(counts - expect) < 1.0 + 1e-6)
        assert np.all(np.diff(idx[n]) >= 0)  # sorted ancestors


def test_resample_handles_neg_inf_and_changes_particle_count():
    logw = np.array([[0.0, -np.inf, -1.0, -np.inf]], np.float32)
    for mode, u in (("systematic", np.array([0.3], np.float32)),
                    ("multinomial", np.random.RandomState(0).uniform(0, 1, (1, 9)).astype(np.float32))):
        idx = rs.resample_indices(logw, u, mode, num_out=9)
        assert idx.shape == (1, 9) and set(idx.ravel()) <= {0, 2}


def test_multinomial_frequencies():
    rng = np.random.RandomState(5)
    logw = np.log(np.array([[0.5, 0.25, 0.125, 0.125]], np.float32))
    idx = rs.resample_indices(logw, rng.uniform(0, 1, (1, 40000)).astype(np.float32), "multinomial", 40000)
    freq = np.bincount(idx[0], minlength=4) / 40000
    np.testing.assert_allclose(freq, [0.5, 0.25, 0.125, 0.125], atol=0.01)


def test_reweight_resample_none_mode_matches_logsumexp():
    rng = np.random.RandomState(6)
    N, M, d = 4, 33, 3
    ll = rng.standard_normal((N, M)).astype(np.float32)
    lw = np.log(rng.dirichlet(np.ones(M), N)).astype(np.float32)
    x = rng.standard_normal((N, M, d)).astype(np.float32)
    est, xo, lwo, idx = rs.reweight_resample(ll, lw, x, None, "none")
    t = torch.from_numpy(ll + lw)
    want = t - torch.logsumexp(t, 1, keepdim=True)
    np.testing.assert_allclose(lwo, want.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(est, (want.exp()[:, :, None] * torch.from_numpy(x)).sum(1).numpy(), rtol=1e-5, atol=1e-6)
    assert idx is None


def test_particle_count_adaptation_without_resampling():
    """eval (300 particles) -> train (30) -> eval-sized again with ``resample=False``: upstream's
    copy / sample-without-replacement adaptation keeps the weights normalised, draws every new
    particle from the old set, and expanding by whole copies leaves the estimate unchanged."""
    A, B, L, Rt = _system()
    d, N = 3, 4
    pf = tf.filters.ParticleFilter(dynamics_model=LinearDynamics(A, B, L), measurement_model=GaussianLik(d, Rt @ Rt.T),
                                   num_particles=300, resample=False)
    pf.noise = NoiseSource(7)
    g = torch.Generator().manual_seed(8)
    pf.initialize_beliefs(mean=torch.zeros(N, d), covariance=(0.1 * torch.eye(d))[None].expand(N, d, d))
    step = lambda: pf(observations={"z": 0.3 * torch.randn(N, d, generator=g)}, controls=torch.randn(N, 7, generator=g))
    step()
    old = pf.particle_states.clone()
    pf.num_particles = 30  # shrink: a sample without replacement, shared by the batch
    step()
    assert pf.particle_states.shape == (N, 30, d) and pf.particle_log_weights.shape == (N, 30)
    torch.testing.assert_close(torch.logsumexp(pf.particle_log_weights, 1), torch.zeros(N), atol=1e-5, rtol=0)
    before = pf.particle_states.clone(), pf.particle_log_weights.clone()
    pf.num_particles = 75  # expand: two whole copies + 15 without replacement
    pf._adapt_particle_count()
    S, W = pf.particle_states, pf.particle_log_weights
    assert S.shape == (N, 75, d)
    assert torch.equal(S[:, :30], before[0]) and torch.equal(S[:, 30:60], before[0])
    for n in range(N):
        rows = {tuple(r.tolist()) for r in before[0][n]}
        tail = [tuple(r.tolist()) for r in S[n, 60:]]
        assert set(tail) <= rows and len(set(tail)) == 15
    torch.testing.assert_close(torch.logsumexp(W, 1), torch.zeros(N), atol=1e-5, rtol=0)
    pf.num_particles = 60  # whole copies only: the weighted mean is unchanged
    pf.particle_states, pf.particle_log_weights = before
    mean = (before[1].exp()[:, :, None] * before[0]).sum(1)
    pf._adapt_particle_count()
    torch.testing.assert_close((pf.particle_log_weights.exp()[:, :, None] * pf.particle_states).sum(1), mean,
                               atol=1e-6, rtol=1e-5)
    del old


def test_unscented_filter_equals_kalman_filter_on_linear_system():
    """The unscented transform is exact for linear maps: the (general-form) oracle UKF reproduces
    the Kalman closed form with either sigma-point strategy."""
    A, B, L, Rt = _system()
    N, T, d = 4, 5, 3
    g = torch.Generator().manual_seed(11)
    us = torch.randn(T, N, 7, generator=g)
    zs = torch.randn(T, N, d, generator=g)
    mu0 = torch.randn(N, d, generator=g)
    for strategy in (tf.filters.JulierSigmaPointStrategy(), tf.filters.MerweSigmaPointStrategy(alpha=0.5)):
        f = tf.filters.VirtualSensorUnscentedKalmanFilter(dynamics_model=LinearDynamics(A, B, L),
                                                          virtual_sensor_model=DirectSensor(d, Rt),
                                                          sigma_point_strategy=strategy)
        f.initialize_beliefs(mean=mu0, covariance=(0.1 * torch.eye(d))[None].expand(N, d, d))
        est = f.forward_loop(observations={"z": zs}, controls=us)
        for n in range(N):
            want = _kalman_closed_form(A, B, L, Rt, mu0[n], 0.1 * torch.eye(d), us[:, n], zs[:, n])
            for t in range(T):
                torch.testing.assert_close(est[t, n], want[t][0], rtol=1e-4, atol=1e-5)
            torch.testing.assert_close(f._belief_covariance[n], want[-1][1], rtol=1e-4, atol=1e-5)
        wc, wm = strategy.compute_sigma_weights(d)
        assert abs(float(wm.sum()) - 1.0) < 1e-5


# ------------------------------------------------------------------------------ soft resampling
def test_soft_resampling_mixture_is_alpha_w_plus_uniform():
    """Fixed-point mixture weights: proportional to ``alpha w + (1 - alpha) / M`` to the scheme's
    resolution (every weight is floored to a multiple of 2^-24 of the LARGEST one, so a probability
    is off by at most M 2^-24), bounded by 2^24 each, and ``alpha = 1`` is the plain resampler."""
    from oracle import resample as rs

    rng = np.random.RandomState(3)
    N, M = 5, 777
    logw = (rng.standard_normal((N, M)) * 3).astype(np.float32)
    logw[:, 10] = -np.inf
    q, e, _ = rs.quantise(logw)
    for alpha in (0.25, 0.5, 0.9):
        qm = rs.soft_mixture(q, alpha)
        assert qm.max() <= (1 << 24)
        w = e.astype(np.float64) / e.astype(np.float64).sum(1, keepdims=True)
        want = alpha * w + (1 - alpha) / M
        got = qm.astype(np.float64) / qm.astype(np.float64).sum(1, keepdims=True)
        assert np.abs(got - want).max() < M * 2.0 ** -24
        assert (qm[:, 10] > 0).all()  # a zero-weight particle is reachable through the uniform part
    u = rng.uniform(0, 1, N).astype(np.float32)
    np.testing.assert_array_equal(rs.resample_indices(logw, u, "systematic", soft_alpha=1.0),
                                  rs.resample_indices(logw, u, "systematic"))


def test_soft_resampling_is_unbiased():
    """Importance weights ``w / mixture`` undo the mixture: over many trajectories the weighted mean
    of the survivors equals the posterior mean (same inputs, independent uniforms)."""
    from oracle import resample as rs

    rng = np.random.RandomState(4)
    N, M, d = 4000, 64, 2
    ll = np.tile((rng.standard_normal((1, M)) * 2).astype(np.float32), (N, 1))
    lw = np.full((N, M), -np.log(M), np.float32)
    x = np.tile(rng.standard_normal((1, M, d)).astype(np.float32), (N, 1, 1))
    u = rng.uniform(0, 1, (N, M)).astype(np.float32)
    est, xo, lwo, _ = rs.reweight_resample(ll, lw, x, u, "multinomial", soft_alpha=0.5)
    assert abs(np.exp(lwo.astype(np.float64)).sum(1) - 1).max() < 1e-5
    # self-normalised importance sampling: ratio of means (its bias is O(1 / M) of the per-draw ratio)
    w = np.exp(lwo.astype(np.float64))
    per_traj = (w[:, :, None] * xo).sum(1)
    spread = x[0].std(0)
    assert np.abs(per_traj.mean(0) - est[0]).max() < 0.05 * spread.max()
    # the plain resampler's survivors, equally weighted, agree too
    _, xo1, _, _ = rs.reweight_resample(ll, lw, x, u, "multinomial")
    assert np.abs(xo1.mean(1).mean(0) - est[0]).max() < 0.05 * spread.max()


def test_oracle_particle_filter_soft_resampling_weights():
    """``oracle.tf.filters.ParticleFilter(soft_resample_alpha=0.5)``: after a resampling step the
    log-weights are the normalised ``logw - log(mixture)`` of the ancestors (upstream ``_resample``),
    and they equal ``oracle.resample.reweight_resample``'s."""
    A, B, L, Rt = _system(2)
    dyn, lik = LinearDynamics(A, B, L), GaussianLik(2, Rt @ Rt.T)
    M, N = 200, 3
    f = tf.filters.ParticleFilter(dynamics_model=dyn, measurement_model=lik, num_particles=M,
                                  resample_mode="systematic", soft_resample_alpha=0.5)
    f.eval()
    g = torch.Generator().manual_seed(0)
    eps0, eps1 = torch.randn((N, M, 2), generator=g), torch.randn((N, M, 2), generator=g)
    u = torch.rand((N,), generator=g)
    f.noise = ReplayNoise([eps0, eps1], [u])
    f.initialize_beliefs(mean=torch.zeros(N, 2), covariance=(0.1 * torch.eye(2))[None].expand(N, 2, 2))
    states0, lw0 = f.particle_states.clone(), f.particle_log_weights.clone()
    z, ctrl = 0.3 * torch.randn((N, 2), generator=g), torch.randn((N, 7), generator=g)
    f(observations={"z": z}, controls=ctrl)
    lw = f.particle_log_weights
    assert float((torch.logsumexp(lw, dim=1)).abs().max()) < 1e-5
    assert float(lw.std()) > 1e-3  # not the uniform reset of the plain resampler
    # the same step through the normative numpy restatement
    pred, tril = dyn(initial_states=states0.reshape(N * M, 2), controls=ctrl.repeat_interleave(M, 0))
    states = (pred + torch.einsum("rij,rj->ri", tril, eps1.reshape(N * M, 2))).reshape(N, M, 2)
    ll = lik(states=states, observations={"z": z})
    _, xo, lwo, idx = rs.reweight_resample(ll.numpy(), lw0.numpy(), states.numpy(), u.numpy(), "systematic", soft_alpha=0.5)
    np.testing.assert_array_equal(f.last_resample_indices.numpy(), idx)
    np.testing.assert_allclose(lw.numpy(), lwo, rtol=1e-5, atol=1e-5)
