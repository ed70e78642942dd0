"""The bit-exact CPU twin of the engine's strict (exact-fp32) mode, ``oracle/strict``, against
the torch oracle it restates (which ``tests/golden`` pins to the reference): same networks, same
inputs, results within a few fp32 roundings -- so that bit-equality of the HIP engine with the
twin (``tests/test_gpu_strict.py``) is also 1e-6-equality with the reference's arithmetic.
Tolerances are written at each comparison."""
import math

import numpy as np
import pytest
import torch

from oracle import models as om
from oracle import resample as ors
from oracle import strict
from oracle.tf.base import ReplayNoise


def test_deterministic_transcendentals_against_fp64():
    g = np.random.default_rng(0)
    x = -np.abs(g.standard_normal(20000) * 20).astype(np.float32)
    e = strict.det_exp_nonpos(x)
    assert np.array_equal(e, ors.detexp(x))                       # the resampler's numpy definition, bit for bit
    ref = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - ref) / ref / (1.0 + np.abs(x))) < 1.5e-7   # the argument x log2(e) is rounded once
    s = (1.0 + g.random(20000) * 7).astype(np.float32)              # sums of up to 8 exponentials
    assert np.max(np.abs(strict.det_log(s) - np.log(s.astype(np.float64)))) < 2.5e-7
    wide = np.exp(g.uniform(-80, 80, 20000)).astype(np.float32)
    lw = np.log(wide.astype(np.float64))
    assert np.max(np.abs(strict.det_log(wide) - lw) / np.maximum(1.0, np.abs(lw))) < 2.5e-7
    gte = (g.standard_normal(20000) * 8).astype(np.float32)
    assert np.max(np.abs(strict.det_sigmoid(gte) - 1.0 / (1.0 + np.exp(-gte.astype(np.float64))))) < 3e-7
    a = (g.standard_normal(20000) * 5).astype(np.float32)
    b = (g.standard_normal(20000) * 5).astype(np.float32)
    b[:10] = -np.inf
    a[5:15] = -np.inf
    lae = strict.det_logaddexp(a, b)
    with np.errstate(divide="ignore"):
        want = np.logaddexp(a.astype(np.float64), b.astype(np.float64))
    assert np.all(np.isneginf(lae[5:10])) and np.all(np.isfinite(lae[15:]))
    fin = np.isfinite(want)
    assert np.max(np.abs(lae[fin] - want[fin])) < 1.5e-6


def test_strict_layers_match_the_torch_oracle():
    torch.manual_seed(0)
    task = om.TASKS["door"]
    g = torch.Generator().manual_seed(1)
    N = 5
    img = (torch.randn((N, 32, 32), generator=g) * 0.5).clamp(-1, 1)
    enc = om.image_encoder(64)
    with torch.no_grad():
        want = enc(img[:, None]).numpy()
    got = strict.image_encoder(enc, img)
    assert np.max(np.abs(got - want)) / max(1.0, np.abs(want).max()) < 2e-6
    ve = om.vector_encoder(7, 64)
    x = torch.randn((N, 7), generator=g)
    with torch.no_grad():
        want = ve(x).numpy()
    assert np.max(np.abs(strict.vector_encoder(ve, x) - want)) < 2e-6


def test_strict_weighted_mean_estimate():
    g = np.random.default_rng(3)
    for M in (30, 300, 4096, 5000):
        lw = (g.standard_normal((3, M)) * 2).astype(np.float32)
        x = g.standard_normal((3, M, 3)).astype(np.float32)
        e = ors.quantise(lw)[1].astype(np.float64)
        want = (e[:, :, None] * x).sum(1) / e.sum(1)[:, None]
        assert np.max(np.abs(strict.estimate(lw, x) - want)) < 2e-6


@pytest.mark.parametrize("cls", ["DoorCrossmodalParticleFilter", "PushUnimodalParticleFilter", "DoorParticleFilter"])
def test_strict_particle_filter_tracks_the_torch_oracle(cls):
    """Teacher-forced over 3 steps (the strict twin restarts every step from the torch oracle's belief):
    log-likelihood-driven ancestors certified, posterior means within 2e-6 relative."""
    o = om.build(cls)
    o.load_state_dict(om.seeded_state_dict(o, seed=5, gain=1.0))
    o.eval()
    d = o.state_dim
    N, M, T = 3, 64, 3
    g = torch.Generator().manual_seed(7)
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1),
           "gripper_pos": torch.randn((T, N, 3), generator=g), "gripper_sensors": torch.randn((T, N, 7), generator=g)}
    ctrl = torch.randn((T, N, 7), generator=g)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    us = [torch.rand((N,), generator=g) for _ in range(T)]
    o.num_particles = M
    o.noise = ReplayNoise([eps0] + eps, us)
    s = strict.StrictParticleFilter(o)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    with torch.no_grad():
        o.initialize_beliefs(mean=torch.randn((N, d), generator=g), covariance=cov)
        for t in range(T):
            s.set_belief(o.particle_states, o.particle_log_weights)
            before = o.particle_states.clone()
            want = o(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t]).numpy()
            got = s.step(observations={k: v[t] for k, v in obs.items()}, controls=ctrl[t], eps=eps[t], u=us[t])
            assert np.max(np.abs(got - want)) / max(1.0, np.abs(want).max()) < 2e-6, t
            idx_o = o.last_resample_indices.numpy()
            bad = int((idx_o != s.last_resample_indices).sum())
            assert bad <= 2, (t, bad)


def _philox_python(key, ctr, rounds=10):
    """Philox4x32 from its definition (Salmon et al., SC'11), in Python integers."""
    M0, M1, W0, W1, mask = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    k0, k1 = key
    c = list(ctr)
    for _ in range(rounds):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & mask, p1 & mask, ((p0 >> 32) ^ c[3] ^ k1) & mask, p0 & mask]
        k0, k1 = (k0 + W0) & mask, (k1 + W1) & mask
    return c


def test_counter_based_noise_generator():
    """``include/mmf_philox.h`` as compiled into the checker: Philox4x32-10 against the Random123
    known-answer vectors and a from-the-definition Python restatement; the Box-Muller normals and the
    resampling uniforms against their distributions; blocks are functions of (seed, step, trajectory,
    particle) only, so a shard draws what the whole batch would."""
    kat = [((0, 0), (0, 0, 0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff, 0xffffffff), (0xffffffff,) * 4, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0xa4093822, 0x299f31d0), (0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for key, ctr, want in kat:
        assert tuple(_philox_python(key, ctr)) == want
        assert tuple(strict.philox_raw(key, ctr)) == want
    g = np.random.default_rng(5)
    for _ in range(50):
        key = [int(x) for x in g.integers(0, 2 ** 32, 2)]
        ctr = [int(x) for x in g.integers(0, 2 ** 32, 4)]
        assert strict.philox_raw(key, ctr) == _philox_python(key, ctr)
    from scipy import stats

    z = strict.philox_normals(1234, 7, 64, 4096, 4)
    assert np.all(np.isfinite(z)) and abs(float(z.mean())) < 5e-3 and abs(float(z.var()) - 1.0) < 5e-3
    for i in range(4):
        assert stats.kstest(z[..., i].ravel()[:200000], "norm").pvalue > 1e-3
    assert abs(float(np.corrcoef(z[..., 0].ravel(), z[..., 1].ravel())[0, 1])) < 5e-3
    assert float(np.abs(z).max()) > 4.0          # tails are there (Box-Muller on 23-bit uniforms: up to 5.7 sigma)
    u = strict.philox_uniforms(1234, 0, 500, 100)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and stats.kstest(u.ravel(), "uniform").pvalue > 1e-3
    # another step / seed / stream: different draws; a shard reproduces its slice of the whole batch
    assert not np.array_equal(z, strict.philox_normals(1234, 8, 64, 4096, 4))
    assert not np.array_equal(z, strict.philox_normals(1235, 7, 64, 4096, 4))
    assert np.array_equal(strict.philox_normals(1234, 7, 16, 4096, 4, traj0=32), z[32:48])
    assert np.array_equal(strict.philox_normals(1234, 7, 64, 4096, 3), z[..., :3])
