"""Training backend "autograd" (opt-in): in train() mode the engine's modules evaluate through
differentiable torch ops on the device, so end-to-end losses and their gradients must equal
the CPU oracle's (same weights, same pre-drawn noise).  eval() always means the HIP path."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import models as om
from oracle.tf.base import ReplayNoise

from _tol import rel_err, scalar_rel


@pytest.fixture()
def autograd_backend():
    from multimodalfilter_amd import engine

    engine.set_training_backend("autograd")
    yield
    engine.set_training_backend(None)


@pytest.fixture(params=["autograd", "hip"])
def training_backend(request):
    """"autograd": torch ops throughout; "hip": per-particle networks through K6."""
    from multimodalfilter_amd import engine

    engine.set_training_backend(request.param)
    yield request.param
    engine.set_training_backend(None)


def _data(task, T, N, seed):
    g = torch.Generator().manual_seed(seed)
    d = task.state_dim
    obs = {"image": (torch.randn((T, N, 32, 32), generator=g) * 0.5).clamp(-1, 1),
           "gripper_pos": torch.randn((T, N, 3), generator=g),
           "gripper_sensors": torch.randn((T, N, 7), generator=g)}
    return obs, torch.randn((T, N, 7), generator=g), torch.randn((N, d), generator=g), \
        torch.randn((T, N, d), generator=g), g


# fp32 engine vs fp32 CPU oracle: different summation orders, and a pre-activation within rounding of
# zero may take the other ReLU branch (kernel-level tests against fp64 with shared masks hold 1e-4)
GRAD_TOL = 5e-3  # observed up to 2.4e-3 (the 5x5 stem, 25 weights each summing a million products), same under both backends


def _compare_grads(oracle, engine_model, loss_o, loss_e, min_checked=20):
    assert scalar_rel(loss_e.detach(), loss_o.detach()) < 1e-4
    loss_o.backward()
    loss_e.backward()
    eng = dict(engine_model.named_parameters())
    checked = 0
    for name, p in oracle.named_parameters():
        if p.grad is None:
            assert eng[name].grad is None or float(eng[name].grad.abs().max()) == 0.0, name
            continue
        g = eng[name].grad.cpu()
        scale = max(1e-6, float(p.grad.abs().max()))
        assert float((g - p.grad).abs().max()) / scale < GRAD_TOL, name
        checked += 1
    assert checked > min_checked


def test_trainable_process_noise_gets_its_gradient(training_backend):
    """the native training recursion (``PfTrainLoopFunction``) returns no gradient for the process-noise
    factor, which is right for the reference's frozen Q only.  With ``Q_scale_tril_diag.requires_grad`` flipped the
    filter must fall back to the step-by-step loop, and the gradient of Q equals the oracle's."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    task = om.TASKS["door"]
    d, T, N, M = task.state_dim, 3, 4, 30
    obs, ctrl, x0, target, g = _data(task, T, N, 23)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)

    oracle = om.ParticleFilter(task, "crossmodal")
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=9, gain=1.0))
    oracle.train()
    oracle.num_particles = M
    oracle.dynamics_model.Q_scale_tril_diag.requires_grad_(True)
    oracle.noise = ReplayNoise([eps0] + eps, [])
    oracle.initialize_beliefs(mean=x0, covariance=cov)
    loss_o = torch.mean((oracle.forward_loop(observations=obs, controls=ctrl) - target) ** 2)
    loss_o.backward()

    eng = mmf.model_types("door")["DoorCrossmodalParticleFilter"]()
    eng.load_state_dict(oracle.state_dict())
    eng.to(dev).train()
    eng.num_particles = M
    eng.dynamics_model.Q_scale_tril_diag.requires_grad_(True)
    eng.noise = mmf.ReplayNoise([eps0] + eps, [])
    eng.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    loops = []
    real_loop = engine.PfTrainLoopFunction.apply
    engine.PfTrainLoopFunction.apply = lambda *a: (loops.append(1), real_loop(*a))[1]
    try:
        pred = eng.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
    finally:
        engine.PfTrainLoopFunction.apply = real_loop
    assert not loops, "a trainable Q must not take the native recursion (it has no gradient for Q)"
    loss_e = torch.mean((pred - target.to(dev)) ** 2)
    loss_e.backward()
    gq_o = oracle.dynamics_model.Q_scale_tril_diag.grad
    gq_e = eng.dynamics_model.Q_scale_tril_diag.grad
    assert gq_e is not None and float(gq_o.abs().max()) > 0
    assert float((gq_e.cpu() - gq_o).abs().max()) / float(gq_o.abs().max()) < GRAD_TOL
    assert scalar_rel(loss_e, loss_o) < 1e-4


@pytest.mark.parametrize("tname,cls,kind,N,M", [("door", "DoorCrossmodalParticleFilter", "crossmodal", 4, 30),
                                                ("push", "PushUnimodalParticleFilter", "unimodal", 4, 30),
                                                # config 5's particle count against the CPU oracle itself
                                                ("push", "PushUnimodalParticleFilter", "unimodal", 2, 8192)])
def test_particle_filter_training_step_matches_oracle(training_backend, tname, cls, kind, N, M):
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    calls = []
    real = engine.ParticleNetFunction.apply

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d, T = task.state_dim, 3  # M = 30: the reference's training particle count
    obs, ctrl, x0, target, g = _data(task, T, N, 21)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)

    oracle = om.ParticleFilter(task, kind)
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=8, gain=1.0))
    oracle.train()
    assert oracle.num_particles == 30
    oracle.num_particles = M
    oracle.noise = ReplayNoise([eps0] + eps, [])
    oracle.initialize_beliefs(mean=x0, covariance=cov)
    loss_o = torch.mean((oracle.forward_loop(observations=obs, controls=ctrl) - target) ** 2)

    eng = mmf.model_types(tname)[cls]()
    eng.load_state_dict(oracle.state_dict())
    eng.to(dev).train()
    assert eng.num_particles == 30
    eng.num_particles = M
    eng.noise = mmf.ReplayNoise([eps0] + eps, [])
    eng.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    loops = []
    real_loop = engine.PfTrainLoopFunction.apply
    engine.ParticleNetFunction.apply = lambda *a: (calls.append(1), real(*a))[1]
    engine.PfTrainLoopFunction.apply = lambda *a: (loops.append(1), real_loop(*a))[1]
    try:
        pred = eng.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
    finally:
        engine.ParticleNetFunction.apply = real
        engine.PfTrainLoopFunction.apply = real_loop
    assert pred.requires_grad
    # "hip": the whole recursion went through the native K6 loop (one Function for all T steps)
    assert (len(loops) == 1) == (training_backend == "hip") and not (loops and calls)
    loss_e = torch.mean((pred - target.to(dev)) ** 2)
    _compare_grads(oracle, eng, loss_o, loss_e)


@pytest.mark.parametrize("tname,cls,kw", [("door", "DoorCrossmodalKalmanFilter", {}),
                                          ("push", "PushUnimodalKalmanFilter", {}),
                                          ("door", "DoorKalmanFilter", {})])
def test_kalman_filter_training_step_matches_oracle(training_backend, tname, cls, kw):
    """"autograd": torch ops throughout; "hip": every sub-filter's Kalman algebra through K3
    forward + ``mmf_ekf_step_backward`` (``engine.EkfStepFunction``)."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    calls = []
    real = engine.EkfStepFunction.apply
    engine.EkfStepFunction.apply = lambda *a: (calls.append(1), real(*a))[1]

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d, T, N = task.state_dim, 3, 6
    obs, ctrl, x0, target, _ = _data(task, T, N, 22)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    oracle = om.build(cls, **kw)
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=9, gain=1.0))
    oracle.train()
    oracle.initialize_beliefs(mean=x0, covariance=cov)
    loss_o = torch.mean((oracle.forward_loop(observations=obs, controls=ctrl) - target) ** 2)
    eng = mmf.model_types(tname)[cls](**kw)
    eng.load_state_dict(oracle.state_dict())
    eng.to(dev).train()
    eng.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
    try:
        pred = eng.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
    finally:
        engine.EkfStepFunction.apply = real
    assert (len(calls) >= T) == (training_backend == "hip")
    loss_e = torch.mean((pred - target.to(dev)) ** 2)
    _compare_grads(oracle, eng, loss_o, loss_e)


@pytest.mark.parametrize("d", [1, 2, 3, 4])
def test_k6_ekf_step_function_matches_fp64_autograd(d):
    """K3 backward: every gradient of ``engine.EkfStepFunction`` (A, mu_pred, z, r_tril, Sigma)
    against fp64 torch autograd through the same algebra, 1e-4 relative, ragged N."""
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    N = 301
    g = torch.Generator().manual_seed(60 + d)
    A = (torch.eye(d) + 0.2 * torch.randn((N, d, d), generator=g))
    mp, z = torch.randn((N, d), generator=g), torch.randn((N, d), generator=g)
    q = torch.diag(0.1 + 0.2 * torch.rand((d,), generator=g))
    T = torch.diag_embed(0.2 + torch.rand((N, d), generator=g)) + 0.05 * torch.randn((N, d, d), generator=g)
    B = torch.randn((N, d, d), generator=g)
    S = 0.1 * torch.eye(d) + 0.05 * B @ B.transpose(-1, -2)
    gm, gS = torch.randn((N, d), generator=g), torch.randn((N, d, d), generator=g)
    leaves = [A, mp, z, T, S]
    dv = [t.to(dev).requires_grad_(True) for t in leaves]
    mu, Sn = engine.EkfStepFunction.apply(dv[0], dv[1], q.to(dev), dv[2], dv[3], dv[4])
    got = torch.autograd.grad([mu, Sn], dv, [gm.to(dev), gS.to(dev)])
    r = [t.double().requires_grad_(True) for t in leaves]
    A6, mp6, z6, T6, S6 = r
    Sp = A6 @ S6 @ A6.transpose(-1, -2) + (q @ q.T).double()
    K = Sp @ torch.inverse(Sp + T6 @ T6.transpose(-1, -2))
    mu6 = mp6 + (K @ (z6 - mp6)[:, :, None]).squeeze(-1)
    S6n = (torch.eye(d, dtype=torch.float64) - K) @ Sp
    want = torch.autograd.grad([mu6, S6n], r, [gm.double(), gS.double()])
    assert float((mu.detach().cpu().double() - mu6.detach()).abs().max()) < 1e-4
    for name, a, b in zip("A mu_pred z r_tril Sigma".split(), got, want):
        scale = max(1e-6, float(b.abs().max()))
        assert float((a.cpu().double() - b).abs().max()) / scale < 1e-4, name


def test_eval_mode_stays_on_the_hip_path(autograd_backend):
    """The training backend never leaks into evaluation: eval() outputs carry no graph."""
    import multimodalfilter_amd as mmf

    dev = torch.device("cuda:0")
    f = mmf.door_models.DoorParticleFilter().to(dev).eval()
    f.initialize_beliefs(mean=torch.zeros((2, 3), device=dev), covariance=torch.eye(3, device=dev)[None].expand(2, 3, 3) * 0.1)
    out = f(observations={"image": torch.zeros((2, 32, 32), device=dev), "gripper_pos": torch.zeros((2, 3), device=dev),
                          "gripper_sensors": torch.zeros((2, 7), device=dev)}, controls=torch.zeros((2, 7), device=dev))
    assert not out.requires_grad and f.particle_states.shape == (2, 300, 3)


@pytest.mark.parametrize("task,kind", [("door", "dynamics"), ("door", "measure"), ("push", "dynamics"), ("push", "measure")])
@pytest.mark.parametrize("N,M", [(3, 40), (2, 64), (5, 7), (9, 1000), (16, 4096)])
def test_k6_particle_net_function_matches_autograd(task, kind, N, M):
    """K6: head outputs and every gradient (states, per-trajectory bias, all weights and biases)
    of ``engine.ParticleNetFunction`` against torch autograd through the same layers (fp64
    reference), 1e-4 relative."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    assert torch.cuda.is_available()
    ns = mmf.door_models if task == "door" else mmf.push_models
    P = task.capitalize()
    torch.manual_seed(N * 100 + M)
    if kind == "dynamics":
        model = ns.DoorDynamicsModelBrent() if task == "door" else getattr(ns, P + "DynamicsModel")()
    else:
        model = getattr(ns, P + "MeasurementModel")(modalities={"pos", "sensors"})
    model.to(dev)
    net = model._net
    d = net.d_in
    R = N * M
    g = torch.Generator().manual_seed(5)
    states = torch.randn((R, d), generator=g).to(dev).requires_grad_(True)
    tbias = torch.randn((N, 64), generator=g).to(dev).requires_grad_(True)
    gout = torch.randn((R, net.n_out), generator=g).to(dev)
    params = net._sources()
    out = engine.ParticleNetFunction.apply(net, 0 if kind == "dynamics" else 1, N, M, states, tbias, *params)
    got = torch.autograd.grad(out, [states, tbias] + params, gout)

    # reference: the same network in fp64 torch ops
    p64 = [p.detach().double().requires_grad_(True) for p in params]
    s64 = states.detach().double().requires_grad_(True)
    t64 = tbias.detach().double().requires_grad_(True)
    # Among millions of pre-activations a few sit within fp32 rounding of zero, where an fp32
    # forward and an fp64 one take different ReLU branches (one such particle moves a
    # per-trajectory gradient by 1e-3).  For the large cases the reference therefore applies the
    # masks the kernel's own forward produced (its stash: layer inputs > 0), so that only the
    # arithmetic is compared; the small cases use plain ReLUs.
    NL = 3 + 2 * net.n_res
    masks = None
    if R > 1000:
        from multimodalfilter_amd import _abi
        stash = torch.empty((NL + 1, R, 64), dtype=torch.float32, device=dev)
        scratch = torch.empty((R, net.n_out), dtype=torch.float32, device=dev)
        bits = torch.empty((NL + 1, R, 2), dtype=torch.int32, device=dev)
        _abi.particle_net_train_forward(net.blob(_abi.PREC_F32), net.n_res, 0 if kind == "dynamics" else 1,
                                        states.detach().contiguous(), tbias.detach().contiguous(), stash, bits, scratch, N, M, d)
        masks = (stash > 0).double()
        # the sign bits the backward reads are those of the stash: bit 16 t + r of word h <-> feature
        # 32 t + (r & 3) + 8 (r >> 2) + 4 h
        feat = torch.tensor([[32 * (b // 16) + ((b % 16) & 3) + 8 * ((b % 16) >> 2) + 4 * h for b in range(32)] for h in range(2)],
                            device=dev)
        bits_got = ((bits.long()[..., None] >> torch.arange(32, device=dev)) & 1)     # (NL + 1, R, 2, 32)
        bits_want = (stash > 0).long()[:, :, feat]                                     # (NL + 1, R, 2, 32)
        assert torch.equal(bits_got, bits_want)
    layer = [0]

    def relu(z):  # the output of every ReLU is the input of the next 64x64 layer (or of the head)
        k = layer[0]
        layer[0] += 1
        return torch.relu(z) if masks is None else z * masks[k]

    a = relu(s64 @ p64[0].t() + p64[1])                                   # stash[0]
    h = relu(a @ p64[2].t() + p64[3])                                     # stash[1]
    a = relu(a + h @ p64[4].t() + p64[5])                                 # stash[2]
    off = net.join_state_off
    jn = a @ p64[6][:, off:off + 64].t() + t64.repeat_interleave(M, dim=0)
    if net.relu_after_join:
        a = relu(jn)                                                      # stash[3]
    else:
        a = jn
        layer[0] += 1
    for i in range(net.n_res):
        w1, b1, w2, b2 = p64[7 + 4 * i: 11 + 4 * i]
        h = relu(a @ w1.t() + b1)
        a = relu(a + h @ w2.t() + b2)
    assert layer[0] == NL + 1
    want_out = a @ p64[-2].t() + p64[-1]
    want = torch.autograd.grad(want_out, [s64, t64] + p64, gout.double())

    def rel(x, y):
        return float((x.detach().double() - y.detach()).abs().max()) / max(1e-6, float(y.detach().abs().max()))

    assert rel(out, want_out) < 1e-4
    names = ["states", "traj_bias"] + [f"param{i}" for i in range(len(params))]
    for n, x, y in zip(names, got, want):
        assert x.shape == y.shape, n
        assert rel(x, y) < 1e-4, f"{n}: {rel(x, y):.2e}"


def test_train_filter_step_descends_and_matches_manual_sgd():
    """``train.train_filter_step`` (backend "hip"): the step it takes equals a manual SGD step on
    the same loss computed with the "autograd" backend, and repeated steps reduce the loss."""
    import copy

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic, train

    dev = torch.device("cuda:0")
    d, L, N = 3, 5, 6
    batch = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=3).items()}
    cov = torch.eye(d, device=dev) * 0.1
    torch.manual_seed(1)
    f = mmf.door_models.DoorParticleFilter().to(dev).train()
    g = copy.deepcopy(f)
    eps_init = torch.randn((N, d))
    eps = [torch.randn((N, 30, d)) for _ in range(L)]

    def noise():
        return mmf.ReplayNoise([eps_init] + [e.clone() for e in eps], [])

    lr = 1e-3
    try:
        engine.set_training_backend("hip")
        opt = torch.optim.SGD(f.parameters(), lr=lr)
        loss0 = train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=noise())
        engine.set_training_backend("autograd")
        loss_ref = train.filter_loss(g, batch, initial_covariance=cov, noise=noise())
        assert scalar_rel(loss0, loss_ref) < 1e-4
        loss_ref.backward()
        for (n, p), q in zip(f.named_parameters(), g.parameters()):
            if q.grad is None:
                continue
            want = q.detach() - lr * q.grad
            scale = max(1e-6, float((lr * q.grad).abs().max()))
            # + one rounding of the parameter itself: an update of an analytically-zero gradient (rounding noise of
            # 1e-9 in either backend) may or may not move a parameter of 0.4 by its last bit (3e-8)
            ulp = 1.2e-7 * float(q.detach().abs().max())
            assert float((p.detach() - want).abs().max()) < 1e-2 * scale + ulp, n
        engine.set_training_backend("hip")
        losses = [loss0] + [train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=noise())
                            for _ in range(5)]
        assert losses[-1] < losses[0]
    finally:
        engine.set_training_backend(None)


def test_training_step_at_config_c5_particle_count():
    """SURVEY.md 8d config C5 at its particle count (push, unimodal, 8,192 particles, train mode,
    forward + backward): the "hip" backend's loss and gradients against the "autograd" backend's on
    the same weights and noise; batch and length cut to what the autograd backend's saved activations fit
    in a test (``scripts/bench_train.py`` times the 32 x 16 shape)."""
    import copy

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic, train

    dev = torch.device("cuda:0")
    d, L, N, M = 2, 4, 4, 8192
    batch = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=5).items()}
    cov = torch.eye(d, device=dev) * 0.1
    torch.manual_seed(2)
    f = mmf.push_models.PushUnimodalParticleFilter().to(dev).train()
    f.num_particles = M
    g = copy.deepcopy(f)
    eps_init = torch.randn((N, d))

    def noise():  # the initial-belief perturbation; the particle noise is the filters' own (equal seeds)
        return mmf.ReplayNoise([eps_init], [])

    try:
        engine.set_training_backend("hip")
        loss_hip = train.filter_loss(f, batch, initial_covariance=cov, noise=noise())
        loss_hip.backward()
        engine.set_training_backend("autograd")
        loss_ref = train.filter_loss(g, batch, initial_covariance=cov, noise=noise())
        assert scalar_rel(loss_hip, loss_ref) < 1e-4
        loss_ref.backward()
        checked = 0
        for (n, p), q in zip(f.named_parameters(), g.parameters()):
            if q.grad is None or float(q.grad.abs().max()) == 0.0:
                continue
            scale = float(q.grad.abs().max())
            assert float((p.grad - q.grad).abs().max()) / scale < GRAD_TOL, n
            checked += 1
        assert checked > 20
        engine.set_training_backend("hip")
        opt = torch.optim.SGD(f.parameters(), lr=1e-3)
        # the filter's own particle noise has advanced: a different draw of the same loss
        assert train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=noise()) == pytest.approx(float(loss_hip), rel=0.2)
    finally:
        engine.set_training_backend(None)


@pytest.mark.parametrize("precision,loss_tol,grad_tol", [("bf16", 5e-3, 1e-1), ("f16x3", 1e-5, 1e-3)])
def test_config_c5_training_step_with_reduced_precision_cnn(precision, loss_tol, grad_tol):
    """BASELINE config 5 as specified, on one GPU: push unimodal PF, 8,192 particles, train mode,
    **bf16 measurement CNN on MFMA**, forward + backward + optimiser step.  The image encoder's
    training forward runs its two 32->32 convolutions with bf16 products
    (``set_image_encoder_precision("bf16")`` -> ``mmf_image_convs_train_forward(MMF_PREC_BF16)``), the
    backward differentiates through the saved activations in fp32.  Against the exact-fp32 forward on
    the same weights and noise: loss within 5e-3, every gradient within 1e-1 of its tensor's largest
    entry (a reduced-precision mode: stated tolerance.  Round 6: the bf16 forward is the resident K4 kernel, as at
    inference -- stem and conv 32->16 in bf16 too, where rounds 2-5 kept them exact / f16x3: observed 6.6e-2, before
    3e-2); the f16x3 forward is held to 1e-5 / 1e-3."""
    import copy

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic, train

    dev = torch.device("cuda:0")
    d, L, N, M = 2, 4, 4, 8192
    batch = {k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=5).items()}
    cov = torch.eye(d, device=dev) * 0.1
    torch.manual_seed(2)
    f = mmf.push_models.PushUnimodalParticleFilter().to(dev).train()
    f.num_particles = M
    g = copy.deepcopy(f)
    eps_init = torch.randn((N, d))
    noise = lambda: mmf.ReplayNoise([eps_init], [])
    calls = []
    real = mmf._abi.image_convs_train_forward
    try:
        engine.set_training_backend("hip")
        engine.set_image_encoder_precision("f32")   # the reference run: exact fp32 products in the convolutions
        loss_ref = train.filter_loss(g, batch, initial_covariance=cov, noise=noise())
        loss_ref.backward()
        engine.set_image_encoder_precision(precision)
        mmf._abi.image_convs_train_forward = lambda *a: (calls.append(a[-1]), real(*a))[1]
        loss = train.filter_loss(f, batch, initial_covariance=cov, noise=noise())
        loss.backward()
        assert calls and all(c == mmf._abi.IMAGE_PRECISIONS[precision] for c in calls)
        assert scalar_rel(loss, loss_ref) < loss_tol
        assert precision == "f16x3" or float(loss) != float(loss_ref)  # bf16 really ran
        checked = 0
        for (n, p), q in zip(f.named_parameters(), g.parameters()):
            if q.grad is None or float(q.grad.abs().max()) == 0.0:
                continue
            assert float((p.grad - q.grad).abs().max()) / float(q.grad.abs().max()) < grad_tol, n
            checked += 1
        assert checked > 20
        opt = torch.optim.SGD(f.parameters(), lr=1e-3)
        assert train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=noise()) == pytest.approx(float(loss), rel=0.2)
    finally:
        mmf._abi.image_convs_train_forward = real
        engine.set_image_encoder_precision(None)
        engine.set_training_backend(None)


@pytest.mark.parametrize("N,M,d", [(1, 1, 3), (4, 30, 3), (3, 1000, 2), (2, 8192, 3)])
def test_k6_reweight_estimate_function_matches_autograd(N, M, d):
    """K6 (K1 no-resample path): estimate, normalised log-weights and the gradients w.r.t.
    log-likelihoods, previous log-weights and particles against torch autograd in fp64."""
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(N + M)
    ll = (torch.randn((N, M), generator=g) * 2).to(dev).requires_grad_(True)
    lw = torch.log_softmax(torch.randn((N, M), generator=g), dim=1).to(dev).requires_grad_(True)
    x = torch.randn((N, M, d), generator=g).to(dev).requires_grad_(True)
    g_est = torch.randn((N, d), generator=g).to(dev)
    g_lw = torch.randn((N, M), generator=g).to(dev)
    est, out = engine.ReweightEstimateFunction.apply(ll, lw, x)
    got = torch.autograd.grad([est, out], [ll, lw, x], [g_est, g_lw])

    l64, w64, x64 = (t.detach().double().requires_grad_(True) for t in (ll, lw, x))
    a = w64 + l64
    o64 = a - torch.logsumexp(a, dim=1, keepdim=True)
    e64 = torch.sum(torch.exp(o64)[:, :, None] * x64, dim=1)
    want = torch.autograd.grad([e64, o64], [l64, w64, x64], [g_est.double(), g_lw.double()])

    def rel(p, q):
        return float((p.detach().double() - q).abs().max()) / max(1e-6, float(q.abs().max()))

    assert rel(est, e64.detach()) < 1e-4 and rel(out, o64.detach()) < 1e-4
    for name, p, q in zip(("loglik", "logw_in", "states"), got, want):
        assert rel(p, q) < 1e-4, f"{name}: {rel(p, q):.2e}"


def test_recordings_to_training_to_evaluation_pipeline():
    """Synthetic door recordings -> ``data.trajectory_from_raw`` -> device-resident subsequence
    batches -> ``train.train_filter_step`` (K6 backend) -> ``evaluation.run_filter`` on the
    stacked trajectories: the pieces of SURVEY.md 8f rows 1 and 3 fit together on the GPU."""
    import numpy as np

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import data, engine, evaluation, train

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)

    def recording(T):
        theta = np.cumsum(rng.normal(size=T) * 0.02) + 0.6
        return {"object-state": np.stack([np.zeros(T), theta, np.zeros(T), np.full(T, 0.001), np.full(T, -0.001)], 1).astype(np.float32),
                "eef_pos": (np.array([0.37, -0.1, 1.57]) + rng.normal(size=(T, 3)) * 0.05).astype(np.float32),
                "ee-force-obs": rng.normal(size=(T, 3)).astype(np.float32) * 10,
                "ee-torque-obs": rng.normal(size=(T, 3)).astype(np.float32),
                "contact-obs": (rng.uniform(size=T) > 0.4).astype(np.float32),
                "image": rng.uniform(size=(T, 64, 64)).astype(np.float32)}

    trajs = [data.trajectory_from_raw(recording(T), data.DOOR, image_blackout_ratio=0.3, rng=rng) for T in (20, 17, 25, 18)]
    loader = data.SubsequenceBatcher(trajs, subsequence_length=4, batch_size=4, device=dev, seed=1)
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).train()
    opt = torch.optim.Adam(f.parameters(), lr=1e-3)
    cov = torch.eye(3, device=dev) * 0.1
    engine.set_training_backend("hip")
    try:
        losses = [train.train_filter_step(f, batch, opt, initial_covariance=cov, noise=mmf.NoiseSource(seed=3))
                  for _ in range(2) for batch in loader]
    finally:
        engine.set_training_backend(None)
    assert len(losses) == 2 * len(loader) and all(np.isfinite(losses))
    f.eval()
    f.noise = mmf.NoiseSource(seed=4)
    batch = data.stack_trajectories(trajs, dev)
    pred = evaluation.run_filter(f, batch)
    assert pred.shape == (16, 4, 3) and bool(torch.isfinite(pred).all()) and not pred.requires_grad
    assert evaluation.raw_rmse(evaluation.per_trajectory_mse(pred, batch["states"][1:], start=5)).shape == (3,)


@pytest.mark.parametrize("tname,cls", [("door", "DoorCrossmodalParticleFilter"), ("push", "PushParticleFilter")])
def test_pretraining_measurement_loss_matches_oracle(training_backend, tname, cls):
    """``train_particle_filter_measurement`` (``train_helpers.py:76-96``):
    one perturbed state per sample, regression on its Gaussian log-pdf.  Loss (1e-4) and every
    parameter gradient against the oracle's restatement, both training backends."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import train
    from oracle.tf import train as otrain

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d, N = task.state_dim, 24
    obs, _ctrl, x0, _t, g = _data(task, 1, N, 31)
    obs = {k: v[0] for k, v in obs.items()}
    noisy = x0 + 0.3 * torch.randn((N, d), generator=g)
    target = -0.5 * ((noisy - x0) ** 2).sum(1) / 0.1 - 1.0
    oracle = om.build(cls)
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=12, gain=1.0))
    oracle.train()
    loss_o = otrain.particle_filter_measurement_loss(oracle.measurement_model, noisy_states=noisy, observations=obs,
                                                     log_likelihoods=target)
    eng = mmf.model_types(tname)[cls]()
    eng.load_state_dict(oracle.state_dict())
    eng.to(dev).train()
    batch = {"noisy_states": noisy.to(dev), "log_likelihoods": target.to(dev), **{k: v.to(dev) for k, v in obs.items()}}
    loss_e = train.particle_filter_measurement_loss(eng.measurement_model, batch)
    _compare_grads(oracle.measurement_model, eng.measurement_model, loss_o, loss_e)


@pytest.mark.parametrize("tname", ["door", "push"])
def test_pretraining_virtual_sensor_and_dynamics_losses_match_oracle(training_backend, tname):
    """``train_virtual_sensor`` (``train_helpers.py:98-121``), ``train_dynamics_single_step`` (mse
    and nll) and ``train_dynamics_recurrent`` (``:31-74``): losses and gradients against the
    oracle's restatement."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import train
    from oracle.tf import train as otrain

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d, N, T = task.state_dim, 16, 4
    obs, ctrl, x0, states, g = _data(task, T, N, 41)
    P = tname.capitalize()
    oracle = om.build(f"{P}KalmanFilter")
    oracle.load_state_dict(om.seeded_state_dict(oracle, seed=13, gain=1.0))
    oracle.train()
    eng = mmf.model_types(tname)[f"{P}KalmanFilter"]()
    eng.load_state_dict(oracle.state_dict())
    eng.to(dev).train()

    def fresh():
        oracle.zero_grad(set_to_none=True)
        eng.zero_grad(set_to_none=True)

    # virtual sensor: z(o_{t+1}) against x_{t+1}
    o1 = {k: v[1] for k, v in obs.items()}
    loss_o = otrain.virtual_sensor_loss(oracle.virtual_sensor_model, observations=o1, states=states[1])
    batch = {"next_states": states[1].to(dev), **{k: v.to(dev) for k, v in o1.items()}}
    loss_e = train.virtual_sensor_loss(eng.virtual_sensor_model, batch)
    _compare_grads(oracle.virtual_sensor_model, eng.virtual_sensor_model, loss_o, loss_e, min_checked=10)

    for lf in ("mse", "nll"):
        fresh()
        loss_o = otrain.dynamics_single_step_loss(oracle.dynamics_model, initial_states=states[0], next_states=states[1],
                                                  controls=ctrl[1], loss_function=lf)
        batch = {"initial_states": states[0].to(dev), "next_states": states[1].to(dev), "controls": ctrl[1].to(dev)}
        loss_e = train.dynamics_single_step_loss(eng.dynamics_model, batch, loss_function=lf)
        _compare_grads(oracle.dynamics_model, eng.dynamics_model, loss_o, loss_e, min_checked=10)

    fresh()
    loss_o = otrain.dynamics_recurrent_loss(oracle.dynamics_model, states=states, controls=ctrl)
    loss_e = train.dynamics_recurrent_loss(eng.dynamics_model, {"states": states.to(dev), "controls": ctrl.to(dev)})
    _compare_grads(oracle.dynamics_model, eng.dynamics_model, loss_o, loss_e, min_checked=10)


def test_pretrain_step_reduces_the_measurement_loss():
    """A few optimiser steps through ``train.pretrain_step`` + ``data.ParticleFilterMeasurementBatcher``
    on the K6 backend: the loss goes down and the default noise source advances between draws."""
    import numpy as np

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import data, engine, synthetic, train
    from multimodalfilter_amd.types import TrajectoryNumpy

    dev = torch.device("cuda:0")
    engine.set_training_backend("hip")
    try:
        traj = synthetic.make_trajectories(state_dim=2, T=12, N=4, seed=3)
        trajs = [TrajectoryNumpy(traj["states"][:, n].numpy(),
                                 {k: traj[k][:, n].numpy() for k in ("image", "gripper_pos", "gripper_sensors")},
                                 traj["controls"][:, n].numpy()) for n in range(4)]
        torch.manual_seed(0)
        pf = mmf.push_models.PushParticleFilter().to(dev).train()
        batcher = data.ParticleFilterMeasurementBatcher(trajs, covariance=np.eye(2) * 0.1, samples_per_pair=10,
                                                        batch_size=64, device=dev, seed=5)
        opt = torch.optim.Adam(pf.measurement_model.parameters(), lr=3e-3)
        epochs = []
        for epoch in range(15):
            epochs.append(np.mean([train.pretrain_step(train.particle_filter_measurement_loss, pf.measurement_model, batch, opt)
                                   for batch in batcher]))
        assert epochs[-1] < 0.6 * epochs[0], epochs
        # default noise sources are persistent (consecutive draws differ)
        src = train.default_noise(pf)
        a = src.gaussian((3, 2), like=torch.zeros(1, device=dev))
        b = src.gaussian((3, 2), like=torch.zeros(1, device=dev))
        assert not torch.equal(a, b)
    finally:
        engine.set_training_backend(None)


@pytest.mark.parametrize("N", [1, 5, 37, 130])
@pytest.mark.parametrize("precision", [None, "f16x3"], ids=["exact_f32", "f16x3"])
def test_k6_image_convs_function_matches_fp64_autograd(N, precision):
    """K6 for the image encoder: outputs and every gradient (five conv weights, five biases) of
    ``engine.ImageConvsFunction`` -- forward with kept activations, dgrad on transposed + flipped
    weights with fused ReLU masks, split-K MFMA weight gradients -- against fp64 torch autograd
    through the same layers, 1e-4 relative; and no MIOpen-backed op is involved.  ``f16x3`` (round 6): the forward is
    the resident K4 kernel keeping its activations, data and weight gradients of the 3x3 layers run on the f16 matrix pipe
    with three products per product and a power-of-two scale per gradient tensor -- the same 1e-4, also with output
    gradients of 1e-7 (scaled ``gout``: a plain f16 split would flush them)."""
    import torch.nn.functional as F

    from multimodalfilter_amd import engine, layers

    engine.set_image_encoder_precision(precision)
    try:
        _image_convs_against_fp64(N, 1.0)
        if precision is not None:
            _image_convs_against_fp64(N, 1e-7)
    finally:
        engine.set_image_encoder_precision(None)


def _image_convs_against_fp64(N, gscale):
    import torch.nn.functional as F

    from multimodalfilter_amd import engine, layers

    dev = torch.device("cuda:0")
    torch.manual_seed(70 + N)
    seq = layers.image_encoder(64).to(dev)
    g = torch.Generator().manual_seed(N)
    img = (torch.randn((N, 32, 32), generator=g) * 0.5).clamp(-1, 1)
    img[N // 2] = 0.0
    gout = torch.randn((N, 8, 32, 32), generator=g) * gscale
    params = engine.PackedImageEncoder(seq)._sources()[:10]
    a4 = engine.ImageConvsFunction.apply(seq, img.to(dev), *params)
    got = torch.autograd.grad(a4, params, gout.to(dev))

    # Among ~1e6 pre-activations a few sit within fp32 rounding of zero, where an fp32 forward and an
    # fp64 one take different ReLU branches (one such pixel moves a weight gradient by 1e-3): the
    # reference applies the masks of the kernel's own forward, so that only arithmetic is compared
    from multimodalfilter_amd import _abi
    mk = lambda c: torch.empty((N, c, 32, 32), dtype=torch.float32, device=dev)
    k1, kh, k2, k3, k4 = mk(32), mk(32), mk(32), mk(16), mk(8)
    _abi.image_convs_train_forward(seq._mmf_packed.blob(), img.to(dev).contiguous(), k1, kh, k2, k3, k4, engine.range_flag(dev),
                                   engine.training_image_precision_code())   # the arithmetic ImageConvsFunction ran in
    m1, mh, m2, m3 = [(t > 0).double().cpu() for t in (k1, kh, k2, k3)]
    p64 = [p.detach().double().cpu().requires_grad_(True) for p in params]
    w1, w2a, w2b, w3, w4, b1, b2a, b2b, b3, b4 = p64
    x = img.double()[:, None]
    a1 = F.conv2d(x, w1, b1, padding=2) * m1
    h = F.conv2d(a1, w2a, b2a, padding=1) * mh
    a2 = (a1 + F.conv2d(h, w2b, b2b, padding=1)) * m2
    a3 = F.conv2d(a2, w3, b3, padding=1) * m3
    ref = F.conv2d(a3, w4, b4, padding=1)
    want = torch.autograd.grad(ref, p64, gout.double())
    assert rel_err(a4.detach(), ref.detach(), dims=3) < 1e-4   # every image's (8, 32, 32) feature map
    for name, a, b in zip("w1 w2a w2b w3 w4 b1 b2a b2b b3 b4".split(), got, want):
        scale = max(1e-30, float(b.abs().max()))
        assert float((a.cpu().double() - b).abs().max()) / scale < 1e-4, (name, gscale)


@pytest.mark.parametrize("tname", ["door", "push"])
@pytest.mark.parametrize("N", [3, 50])
def test_k6_dynamics_with_jacobian_matches_fp64_autograd(tname, N):
    """K6 for K5: ``predict_with_jacobian_autograd`` (primal + tangent rows through the K6 kernels in
    groups of four, tangents following the primal's ReLU masks) against the oracle's dynamics model in
    fp64 with its autograd Jacobian (``create_graph``): ``x'``, ``A`` and the gradients of a random
    linear functional of both with respect to every weight, ``x`` and the controls, 1e-4 relative."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d = task.state_dim
    g = torch.Generator().manual_seed(80 + N)
    x, u = torch.randn((N, d), generator=g), torch.randn((N, 7), generator=g)
    gm, gA = torch.randn((N, d), generator=g), torch.randn((N, d, d), generator=g)
    o = om.DynamicsModel(task)
    o.load_state_dict(om.seeded_state_dict(o, seed=15, gain=1.0))
    o = o.double()
    x6, u6 = x.double().requires_grad_(True), u.double().requires_grad_(True)
    mu6, _ = o(initial_states=x6, controls=u6)
    A6 = o.jacobian(initial_states=x6, controls=u6)
    loss6 = (mu6 * gm.double()).sum() + (A6 * gA.double()).sum()
    names = [n for n, p in o.named_parameters() if p.requires_grad]
    want = torch.autograd.grad(loss6, [x6, u6] + [p for p in o.parameters() if p.requires_grad])

    ns = mmf.door_models if tname == "door" else mmf.push_models
    e = getattr(ns, tname.capitalize() + "DynamicsModel")()
    e.load_state_dict({k: v.float() for k, v in o.state_dict().items()})
    e.to(dev).train()
    engine.set_training_backend("hip")
    try:
        xe, ue = x.to(dev).requires_grad_(True), u.to(dev).requires_grad_(True)
        mu, A = e.predict_with_jacobian_autograd(xe, ue)
        loss = (mu * gm.to(dev)).sum() + (A * gA.to(dev)).sum()
        ep = dict(e.named_parameters())
        got = torch.autograd.grad(loss, [xe, ue] + [ep[n] for n in names])
    finally:
        engine.set_training_backend(None)
    assert rel_err(mu.detach(), mu6.detach(), dims=1) < 1e-4
    assert rel_err(A.detach(), A6.detach(), dims=2) < 1e-4
    for name, a, b in zip(["x", "controls"] + names, got, want):
        scale = max(1e-6, float(b.abs().max()))
        assert float((a.cpu().double() - b).abs().max()) / scale < 1e-4, name


@pytest.mark.parametrize("tname,cls,N,M,T,chunk_rows", [
    ("door", "DoorCrossmodalParticleFilter", 6, 30, 5, 32768),    # the reference's training size class: one chunk
    ("door", "DoorCrossmodalParticleFilter", 5, 300, 3, 600),     # several ragged chunks of trajectories (2, 2, 1)
    ("push", "PushUnimodalParticleFilter", 4, 64, 4, 128),
    ("door", "DoorParticleFilter", 3, 100, 3, 100),               # single measurement network, no modality weights
])
def test_native_training_recursion_matches_stepwise(tname, cls, N, M, T, chunk_rows):
    """``engine.PfTrainLoopFunction`` (``mmf_pf_train_forward`` / ``mmf_pf_train_backward``: the whole
    recursion in two C calls, activations recomputed per chunk of trajectories, weight gradients
    accumulated on the device) against the step-by-step K6 path (one autograd Function per network call):
    same loss to 1e-5 relative, every parameter gradient within GRAD_TOL of its scale (observed 0 .. 1.3e-3) --
    whatever the chunking.  A dropped chunk or step would show as an O(1) difference."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d = task.state_dim
    obs, ctrl, x0, target, g = _data(task, T, N, 31)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    torch.manual_seed(3)
    f = mmf.model_types(tname)[cls]().to(dev).train()
    f.num_particles = M
    engine.set_training_backend("hip")
    old_chunk, old_prec = engine.TRAIN_CHUNK_ROWS, engine.DEFAULT_PRECISION
    engine.TRAIN_CHUNK_ROWS = chunk_rows
    # both paths on exact-fp32 products: with the default f16x3 forward the two particle sets differ by ~1e-6,
    # and on these tiny problems (a few hundred rows) ONE flipped ReLU moves a weight gradient by ~1e-2 of its
    # largest entry -- a property of the comparison, not of the kernels
    engine.set_default_precision("f32")
    results = {}
    try:
        for native in (False, True):
            f.use_native_loop = native
            f.zero_grad(set_to_none=True)
            f.noise = mmf.ReplayNoise([eps0] + eps, [])
            f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
            pred = f.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
            loss = torch.mean((pred - target.to(dev)) ** 2)
            loss.backward()
            results[native] = (float(loss.detach()), pred.detach().clone(),
                               {n: p.grad.detach().clone() for n, p in f.named_parameters() if p.grad is not None},
                               f.particle_states.detach().clone(), f.particle_log_weights.detach().clone())
    finally:
        engine.set_training_backend(None)
        engine.TRAIN_CHUNK_ROWS = old_chunk
        engine.set_default_precision(old_prec)
        f.use_native_loop = True
    (l0, p0, g0, s0, w0), (l1, p1, g1, s1, w1) = results[False], results[True]
    assert scalar_rel(l1, l0) < 1e-5
    assert rel_err(p1, p0, dims=1) < 1e-5
    assert rel_err(s1, s0, dims=1) < 1e-5
    assert float((w0 - w1).abs().max()) < 1e-4
    assert set(g0) == set(g1) and len(g0) > 20
    # a gradient that is zero analytically (the head bias of a single measurement network: the log-weights are
    # normalised, a constant added to every log-likelihood changes nothing) is pure rounding noise in both
    # paths: differences are measured against the larger of the tensor's own scale and 1e-3 of the largest
    # gradient entry of the model
    top = max(float(v.abs().max()) for v in g0.values())
    worst = max((float((g0[k] - g1[k]).abs().max()) / max(1e-3 * top, float(g0[k].abs().max())), k) for k in g0)
    print("largest relative gradient difference:", worst)
    # the two paths' particle sets differ in the last ulp after the first step (the stepwise path applies the
    # sigmoid gate and the noise with torch ops, the native forward in the kernel's fma epilogue), so over several
    # steps a pre-activation at rounding distance from zero may take the other ReLU branch: GRAD_TOL, the file's
    # fp32-vs-fp32 tolerance (scripts/debug/train_loop_diff.py: 0 .. 1.5e-4 for most seeds and sizes, exactly 0
    # at T = 1, 1.3e-3 for one seed at T = 3 whatever the chunking or the model class)
    assert worst[0] < GRAD_TOL, worst


@pytest.mark.parametrize("tname,cls,N,M,T,chunk_rows", [
    ("door", "DoorCrossmodalParticleFilter", 32, 30, 6, 262144),  # the reference's training shape, one chunk
    ("door", "DoorCrossmodalParticleFilter", 5, 300, 3, 600),     # ragged chunks
    ("push", "PushUnimodalParticleFilter", 8, 512, 4, 2048),
])
@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_gradients_of_any_magnitude_survive_the_f16_recompute_buffers(tname, cls, N, M, T, chunk_rows, precision):
    """The backward's recompute buffers are f16 (activations directly, the pre-activation gradients relative to the
    largest magnitude of their 32-row tile with one fp32 scale per row and layer -- three-pass form -- or relative to a
    running exponent per layer -- fused form): a plain f16 store of ``dz`` would flush gradients of 1e-9.  The same step
    with the loss scaled by 1e-6: identical estimates, and every parameter gradient 1e-6 times the unscaled one to 1e-3 of
    its tensor's scale (the backward is linear in the loss; observed <= 3e-4), in the exact-fp32 three-pass form (f32 mode)
    and in the fused form (f16x3 mode)."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d = task.state_dim
    obs, ctrl, x0, target, g = _data(task, T, N, 51)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    torch.manual_seed(5)
    f = mmf.model_types(tname)[cls]().to(dev).train()
    f.num_particles = M
    engine.set_training_backend("hip")
    old_chunk, old_prec = engine.TRAIN_CHUNK_ROWS, engine.DEFAULT_PRECISION
    engine.TRAIN_CHUNK_ROWS = chunk_rows
    engine.set_default_precision(precision)
    seen = []
    real = mmf._abi.pf_train_backward
    mmf._abi.pf_train_backward = lambda a, *rest: (seen.append(int(a.fused)), real(a, *rest))[1]
    results = []
    try:
        for loss_scale in (1.0, 1e-6):
            f.zero_grad(set_to_none=True)
            f.noise = mmf.ReplayNoise([eps0] + eps, [])
            f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
            pred = f.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
            loss = torch.mean((pred - target.to(dev)) ** 2) * loss_scale
            loss.backward()
            torch.cuda.synchronize()
            results.append((pred.detach().clone(), {n: p.grad.detach().clone() for n, p in f.named_parameters() if p.grad is not None}))
    finally:
        mmf._abi.pf_train_backward = real
        engine.set_training_backend(None)
        engine.TRAIN_CHUNK_ROWS = old_chunk
        engine.set_default_precision(old_prec)
    assert seen == [int(precision == "f16x3")] * 2
    (p0, g0), (p1, g1) = results
    assert torch.equal(p0, p1)
    assert set(g0) == set(g1) and len(g0) > 20
    top = max(float(v.abs().max()) for v in g0.values())
    assert top > 0
    worst = max((float((g0[k] - 1e6 * g1[k]).abs().max()) / max(1e-3 * top, float(g0[k].abs().max())), k) for k in g0)
    print("loss x 1e-6 against the unscaled step, largest relative gradient difference:", worst)
    assert worst[0] < 1e-3, worst
    assert all(bool(torch.isfinite(v).all()) for v in g1.values())


@pytest.mark.parametrize("tname,cls,N,M,T,chunk_rows,tol", [
    ("door", "DoorCrossmodalParticleFilter", 32, 30, 6, 262144, GRAD_TOL),  # the reference's training shape, one chunk, tiles straddle trajectories
    # ragged chunks: partial tiles, fewer workgroups than slots.  4,500 rows: ONE pre-activation that lies between the two
    # arithmetics moves a bias gradient by a row's share of the sum (round 4 measured 7.3e-3 for f16x3 against exact-fp32 recompute here)
    ("door", "DoorCrossmodalParticleFilter", 5, 300, 3, 600, 1e-2),
    ("door", "DoorParticleFilter", 3, 100, 3, 100, GRAD_TOL),               # one measurement network; 900 rows: a bias gradient is a sum of few terms
    ("push", "PushUnimodalParticleFilter", 4, 2048, 3, 262144, GRAD_TOL),   # config 5's filter: whole tiles, 64 workgroups
])
def test_fused_network_calls_track_the_three_pass_backward(tname, cls, N, M, T, chunk_rows, tol):
    """``MmfPfTrainArgs.fused`` (round 5): recompute + backward data path + weight gradients of every network call in
    ONE kernel (``mmf_particle_net_train_fused``, the forward pass's own three-product f16 arithmetic) against the
    cross-check form (round 6: the ONE other form left): three passes with exact fp32 products over the f16 recompute
    buffers.  The forward pass is the same (loss and estimates bit-identical); the gradients differ by the two
    arithmetics' 1e-6 on the activations, by 2^-11 relative per stored ``dz`` element, plus the occasional ReLU whose
    pre-activation lies between the arithmetics: GRAD_TOL, the file's fp32-vs-fp32 tolerance.  Twice in a row the fused
    form gives identical bits (its summation order is fixed: no atomics)."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d = task.state_dim
    obs, ctrl, x0, target, g = _data(task, T, N, 41)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    torch.manual_seed(4)
    f = mmf.model_types(tname)[cls]().to(dev).train()
    f.num_particles = M
    engine.set_training_backend("hip")
    old_chunk, old_fused, old_prec = engine.TRAIN_CHUNK_ROWS, engine.TRAIN_FUSED, engine.DEFAULT_PRECISION
    engine.TRAIN_CHUNK_ROWS = chunk_rows
    engine.set_default_precision("f16x3")
    seen = []
    real = mmf._abi.pf_train_backward
    mmf._abi.pf_train_backward = lambda a, *rest: (seen.append(int(a.fused)), real(a, *rest))[1]
    results = []
    try:
        for fused in (False, True, True):
            engine.TRAIN_FUSED = fused
            f.zero_grad(set_to_none=True)
            f.noise = mmf.ReplayNoise([eps0] + eps, [])
            f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
            pred = f.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
            loss = torch.mean((pred - target.to(dev)) ** 2)
            loss.backward()
            torch.cuda.synchronize()
            results.append((loss.detach().clone(), pred.detach().clone(),
                            {n: p.grad.detach().clone() for n, p in f.named_parameters() if p.grad is not None}))
    finally:
        mmf._abi.pf_train_backward = real
        engine.set_training_backend(None)
        engine.TRAIN_CHUNK_ROWS, engine.TRAIN_FUSED = old_chunk, old_fused
        engine.set_default_precision(old_prec)
    assert seen == [0, 1, 1]
    (l0, p0, g0), (l1, p1, g1), (l2, p2, g2) = results
    assert torch.equal(l0, l1) and torch.equal(p0, p1) and torch.equal(l1, l2)
    assert set(g0) == set(g1) and len(g0) > 20
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k  # run to run: the same bits
    top = max(float(v.abs().max()) for v in g0.values())
    worst = max((float((g0[k] - g1[k]).abs().max()) / max(1e-3 * top, float(g0[k].abs().max())), k) for k in g0)
    print("fused vs three-pass backward, largest relative gradient difference:", worst)
    assert worst[0] < tol, worst
    assert all(bool(torch.isfinite(v).all()) for v in g1.values())


@pytest.mark.parametrize("tname,cls,N,M,T,chunk_rows", [
    ("door", "DoorCrossmodalParticleFilter", 32, 30, 6, 262144),  # the reference's training shape
    ("door", "DoorCrossmodalParticleFilter", 5, 300, 3, 600),     # ragged chunks
    ("push", "PushUnimodalParticleFilter", 4, 2048, 3, 262144),
])
def test_measurement_networks_of_a_step_in_one_launch_equal_one_launch_each(tname, cls, N, M, T, chunk_rows):
    """``MmfPfTrainArgs.fused_sets`` (round 5): the measurement networks of a step differentiate independently, so their
    fused calls share ONE launch (``mmf_particle_net_train_fused_multi``, blockIdx.y = network; at the reference's
    training size a call is one tile per wave, i.e. its latency).  Network 0 adds its ``d states`` to the running gradient
    in its own store, the others are added behind it in network order: the same sums in the same order as one launch
    per network -- every gradient bit for bit."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d = task.state_dim
    obs, ctrl, x0, target, g = _data(task, T, N, 43)
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    torch.manual_seed(8)
    f = mmf.model_types(tname)[cls]().to(dev).train()
    f.num_particles = M
    engine.set_training_backend("hip")
    old_chunk, old_merge, old_prec = engine.TRAIN_CHUNK_ROWS, engine.TRAIN_FUSED_MERGE, engine.DEFAULT_PRECISION
    engine.TRAIN_CHUNK_ROWS = chunk_rows
    engine.set_default_precision("f16x3")
    seen = []
    real = mmf._abi.pf_train_backward
    mmf._abi.pf_train_backward = lambda a, *rest: (seen.append((int(a.fused), int(a.fused_sets), int(a.n_meas))), real(a, *rest))[1]
    results = []
    try:
        for merge in (False, True):
            engine.TRAIN_FUSED_MERGE = merge
            f.zero_grad(set_to_none=True)
            f.noise = mmf.ReplayNoise([eps0] + eps, [])
            f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
            pred = f.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
            torch.mean((pred - target.to(dev)) ** 2).backward()
            torch.cuda.synchronize()
            results.append({n: p.grad.detach().clone() for n, p in f.named_parameters() if p.grad is not None})
    finally:
        mmf._abi.pf_train_backward = real
        engine.set_training_backend(None)
        engine.TRAIN_CHUNK_ROWS, engine.TRAIN_FUSED_MERGE = old_chunk, old_merge
        engine.set_default_precision(old_prec)
    assert seen[0][0] == 1 and seen[0][1] == 1 and seen[1][1] == seen[1][2] == 2
    g0, g1 = results
    assert set(g0) == set(g1) and len(g0) > 20
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("task,kind", [("door", "dynamics"), ("door", "measure"), ("push", "dynamics"), ("push", "measure")])
@pytest.mark.parametrize("N,M", [(3, 40), (2, 64), (5, 7), (32, 30), (7, 300)])
def test_k6_fused_network_call_matches_the_exact_fp32_step_kernels(task, kind, N, M):
    """``mmf_particle_net_train_fused`` through the C ABI against ``engine.ParticleNetFunction`` (the exact-fp32 K6 step
    kernels, themselves held to fp64 autograd above): ``d states`` per ROW to 1e-4 (the data path: f16x3 products with a
    power-of-two scale per row, also for trajectories whose gradients are 10^4 below their neighbours'), the
    per-trajectory bias gradient, and every weight / bias gradient rebuilt from the kernel's outputs (``pw`` / ``pb``
    partials, the compact rows of the narrow reductions) to 1e-2 of the tensor's largest entry (f16-rounded operands:
    2^-11 relative per element, sums of a few hundred mixed-sign products)."""
    from _fused_case import run_case

    worst, row_err = run_case(task, kind, N, M, verbose=False, return_rows=True)
    print("fused network call vs exact-fp32 step kernels:", worst, row_err)
    assert row_err < 1e-4, row_err
    assert worst < 1e-2, worst


def test_compact_training_arguments_are_validated_at_the_boundary():
    """``mmf_pf_train_backward`` with ``compact = 1`` but no ``dz_scale`` buffer must refuse (MMF_EINVAL through
    ``MmfError``) instead of writing through a null pointer; with ``compact = 0`` the same call needs none."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine

    dev = torch.device("cuda:0")
    task = om.TASKS["door"]
    d, N, M, T = task.state_dim, 3, 64, 2
    obs, ctrl, x0, target, g = _data(task, T, N, 71)
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    torch.manual_seed(7)
    f = mmf.door_models.DoorParticleFilter().to(dev).train()
    f.num_particles = M
    engine.set_training_backend("hip")
    real = mmf._abi.pf_train_backward

    def without_scale(a, *rest):
        a.dz_scale = None
        return real(a, *rest)

    try:
        mmf._abi.pf_train_backward = without_scale
        f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
        pred = f.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
        loss = torch.mean((pred - target.to(dev)) ** 2)
        with pytest.raises(mmf._abi.MmfError):
            loss.backward()
    finally:
        mmf._abi.pf_train_backward = real
        engine.set_training_backend(None)


@pytest.mark.parametrize("tname,cls,expected", [
    ("door", "DoorCrossmodalParticleFilter", 4),   # control term, two measurement networks' observation terms, weight model
    ("push", "PushUnimodalParticleFilter", 3),
    ("door", "DoorCrossmodalKalmanFilter", 15),    # per step (T = 3; the EKF's training loop is step by step): two virtual sensors, the EKF weight model, each sub-filter's control term
])
def test_hip_backend_differentiates_the_per_trajectory_networks_through_their_programs(tname, cls, expected):
    """Round 5: under the "hip" training backend every N-row network of a training step -- hoisted control / observation
    terms, PF and EKF weight models, virtual sensors -- runs forward and backward as its K7 program
    (``TrajProgram.run_autograd``), not as torch modules; ``engine.TRAIN_TRAJ_PROGRAMS = False`` restores the torch path
    (the cross-check), with gradients within the file's fp32-vs-fp32 tolerance of each other."""
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, trajprog

    dev = torch.device("cuda:0")
    task = om.TASKS[tname]
    d, T, N, M = task.state_dim, 3, 6, 40
    obs, ctrl, x0, target, g = _data(task, T, N, 71)
    torch.manual_seed(9)
    f = mmf.model_types(tname)[cls]().to(dev).train()
    is_pf = hasattr(f, "num_particles")
    if is_pf:
        f.num_particles = M
    eps0 = torch.randn((N, M, d), generator=g)
    eps = [torch.randn((N, M, d), generator=g) for _ in range(T)]
    cov = (torch.eye(d) * 0.1)[None].expand(N, d, d)
    engine.set_training_backend("hip")
    old = engine.TRAIN_TRAJ_PROGRAMS
    calls = []
    real = trajprog.TrajProgram.run_autograd
    trajprog.TrajProgram.run_autograd = lambda self, *a, **k: (calls.append(1), real(self, *a, **k))[1]
    grads = []
    try:
        for on in (True, False):
            engine.TRAIN_TRAJ_PROGRAMS = on
            calls.clear()
            f.zero_grad(set_to_none=True)
            if is_pf:
                f.noise = mmf.ReplayNoise([eps0] + eps, [])
            f.initialize_beliefs(mean=x0.to(dev), covariance=cov.to(dev))
            pred = f.forward_loop(observations={k: v.to(dev) for k, v in obs.items()}, controls=ctrl.to(dev))
            torch.mean((pred - target.to(dev)) ** 2).backward()
            torch.cuda.synchronize()
            assert len(calls) == (expected if on else 0), (on, len(calls))
            grads.append({n: p.grad.detach().clone() for n, p in f.named_parameters() if p.grad is not None})
    finally:
        trajprog.TrajProgram.run_autograd = real
        engine.TRAIN_TRAJ_PROGRAMS = old
        engine.set_training_backend(None)
    g1, g0 = grads
    assert set(g0) == set(g1) and len(g0) > 20
    top = max(float(v.abs().max()) for v in g0.values())
    worst = max((float((g0[k] - g1[k]).abs().max()) / max(1e-3 * top, float(g0[k].abs().max())), k) for k in g0)
    assert worst[0] < GRAD_TOL, worst


@pytest.mark.parametrize("optim", ["sgd", "adam"])
def test_graphed_training_step_replays_the_eager_step(optim):
    """``train.GraphedFilterStep`` (round 6): the reference-sized end-to-end training step -- forward recursion, backward,
    optimiser -- captured ONCE as a hipGraph and replayed per batch (the eager step is host-bound at this size: ~430 launches
    of a few microseconds).  Same kernels in the same order on the same random numbers: over eight steps on changing batches
    the losses and the final weights equal the eager steps' bit for bit, also with the image encoders' training forward on
    the resident K4 kernel."""
    import copy

    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import engine, synthetic, train

    dev = torch.device("cuda:0")
    N, M, L, d = 8, 30, 6, 3
    batches = [{k: v.to(dev) for k, v in synthetic.make_trajectories(state_dim=d, T=L - 1, N=N, seed=3 + i).items()} for i in range(3)]
    cov = torch.eye(d, device=dev) * 0.1
    torch.manual_seed(0)
    f = mmf.door_models.DoorCrossmodalParticleFilter().to(dev).train()
    f.num_particles = M
    g = copy.deepcopy(f)
    mk = (lambda m: torch.optim.SGD(m.parameters(), lr=1e-3)) if optim == "sgd" else (lambda m: torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True))
    of, og = mk(f), mk(g)
    f.noise, g.noise = mmf.NoiseSource(seed=5), mmf.NoiseSource(seed=5)
    engine.set_training_backend("hip")
    engine.set_image_encoder_precision("f16x3" if optim == "adam" else None)
    try:
        step = train.GraphedFilterStep(g, og, initial_covariance=cov, noise=g.noise, eager_steps=2)
        eager = [train.train_filter_step(f, batches[i % 3], of, initial_covariance=cov, noise=f.noise) for i in range(8)]
        graphed = [step(batches[i % 3]) for i in range(8)]
        assert step.graph is not None
        with pytest.raises(AssertionError):
            step({k: v[:, :4] for k, v in batches[0].items()})   # another batch shape: refused, not silently re-captured
    finally:
        engine.set_image_encoder_precision(None)
        engine.set_training_backend(None)
    assert graphed == eager
    for (n, p), q in zip(f.named_parameters(), g.parameters()):
        assert torch.equal(p.detach(), q.detach()), n
