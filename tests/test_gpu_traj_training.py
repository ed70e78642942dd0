"""Reverse mode of the per-trajectory programs (K7) and the image encoder's 8192 -> 64 linear layer: what a training
step used to hand to torch autograd + the GEMM library (``train_helpers.py:124-162`` over ``door_models/layers.py:11-63``,
``crossmodal_pf.py:52-106``).  Each case is held to torch autograd IN FP64 on the same modules; tolerance 1e-5 of a
tensor's largest entry for values and data gradients (exact fp32 products, another summation order), 1e-4 for parameter
gradients summed over thousands of rows."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _rel(a, b):
    b = b.to(torch.float64)
    return float((a.to(torch.float64).cpu() - b.cpu()).abs().max()) / max(1e-30, float(b.abs().max()))


def _encoder(in_dim, units=64):
    from multimodalfilter_amd import layers

    return layers.vector_encoder(in_dim, units)


class _WeightModelLike(nn.Module):
    """image features (differentiable input) + two raw vectors -> encoders -> join -> residual blocks -> head"""

    def __init__(self, units=64, out=2, blocks=2):
        super().__init__()
        from multimodalfilter_amd import layers

        self.pos, self.sen = _encoder(3, units), _encoder(7, units)
        self.fusion = nn.Sequential(nn.Linear(3 * units, units), nn.ReLU(), *[layers.ResLinear(units) for _ in range(blocks)],
                                    nn.Linear(units, out))
        self.hoist = nn.Linear(5 * units, units)   # only columns [units, 3 units) are used by the program

    def reference(self, feat, pos, sen):
        x = torch.cat((feat, self.pos(pos), self.sen(sen)), dim=1)
        y = self.fusion(x)
        b = x[:, 64:] @ self.hoist.weight[:, 64:192].t() + self.hoist.bias
        return y, b

    def program(self):
        from multimodalfilter_amd import _abi
        from multimodalfilter_amd.trajprog import TrajProgram

        p = TrajProgram()
        f = p.load("feat", 64)
        rp = p.load("pos", 3)
        ep = p.vector_encoder(self.pos, rp, 3)
        p.free(rp)
        rs = p.load("sen", 7)
        es = p.vector_encoder(self.sen, rs, 7)
        p.free(rs)
        x = p.linear([(f, 0, 64), (ep, 0, 64), (es, 0, 64)], self.fusion[0], _abi.ACT_RELU)
        for blk in list(self.fusion)[2:-1]:
            p.res_linear(blk, x, 64)
        p.store("y", p.linear([(x, 0, 64)], self.fusion[-1]), self.fusion[-1].out_features)
        p.store("b", p.linear([(ep, 0, 64), (es, 0, 64)], self.hoist, cols=(64, 192)), 64)
        return p


class _TwoHeads(nn.Module):
    """a 128-wide trunk whose halves feed two heads (the virtual sensor's shape, ``kf.py:81-126``): sub-range sources"""

    def __init__(self):
        super().__init__()
        from multimodalfilter_amd import layers

        self.inp = nn.Linear(64, 128)
        self.trunk = layers.ResLinear(128)
        self.h0 = nn.Sequential(nn.Linear(64, 3), nn.ReLU(), layers.ResLinear(3), nn.Linear(3, 3))
        self.h1 = nn.Sequential(nn.Linear(64, 3), nn.ReLU(), layers.ResLinear(3), nn.Linear(3, 3))

    def reference(self, feat):
        sh = self.trunk(torch.relu(self.inp(torch.relu(feat))))
        return self.h0(sh[:, :64]), self.h1(sh[:, 64:])

    def program(self):
        from multimodalfilter_amd import _abi
        from multimodalfilter_amd.trajprog import TrajProgram

        p = TrajProgram()
        f = p.load("feat", 64, act=_abi.ACT_RELU)
        sh = p.linear([(f, 0, 64)], self.inp, _abi.ACT_RELU)
        p.res_linear(self.trunk, sh, 128)
        for name, head, off in (("z", self.h0, 0), ("r", self.h1, 64)):
            a = p.linear([(sh, off, 64)], head[0], _abi.ACT_RELU)
            p.res_linear(head[2], a, 3)
            o = p.linear([(a, 0, 3)], head[3])
            p.store(name, o, 3)
            p.free(a)
            p.free(o)
        return p


def _check(model, inputs, out_names, R, tol_param):
    dev = torch.device("cuda:0")
    ref = copy.deepcopy(model).double()
    model = model.to(dev)
    prog = model.program()
    dev_in = {k: v.to(dev).requires_grad_(k == "feat") for k, v in inputs.items()}
    outs = prog.run_autograd(dev_in, out_names, R)
    ref_in = {k: v.double().requires_grad_(k == "feat") for k, v in inputs.items()}
    ref_out = dict(zip(sorted(out_names), ref.reference(*[ref_in[k] for k in inputs])))
    g = torch.Generator().manual_seed(5)
    loss_e = loss_r = 0.0
    for name in sorted(out_names):
        assert _rel(outs[name], ref_out[name]) < 1e-5, name
        w = torch.randn(ref_out[name].shape, generator=g)
        loss_e = loss_e + (outs[name] * w.to(dev)).sum()
        loss_r = loss_r + (ref_out[name] * w.double()).sum()
    for n in (1 << 12, 1 << 16, 1 << 20, 1 << 22):   # whatever block the allocator hands the backward has held NaNs
        junk = torch.full((n,), float("nan"), device=dev)
        del junk
    loss_e.backward()
    loss_r.backward()
    assert _rel(dev_in["feat"].grad, ref_in["feat"].grad) < 1e-5, "d feat"
    checked = 0
    for (name, p), (_n, q) in zip(model.named_parameters(), ref.named_parameters()):
        if q.grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, name
        assert _rel(p.grad, q.grad) < tol_param, (name, _rel(p.grad, q.grad))
        checked += 1
    return checked


@pytest.mark.parametrize("R", [1, 16, 37, 512, 3000])
def test_traj_program_reverse_mode_matches_fp64_autograd(R):
    torch.manual_seed(R)
    inputs = {"feat": torch.randn(R, 64), "pos": torch.randn(R, 3), "sen": torch.randn(R, 7)}
    class M(_WeightModelLike):
        def reference(self, feat, pos, sen):
            y, b = super().reference(feat, pos, sen)
            return b, y   # sorted names: "b" (the hoisted columns of a wider layer), "y"

    assert _check(M(), inputs, {"y": 2, "b": 64}, R, 1e-4) >= 20


@pytest.mark.parametrize("R", [5, 130])
def test_traj_program_reverse_mode_with_sub_range_sources_and_activated_load(R):
    torch.manual_seed(R + 1)

    class M(_TwoHeads):
        def reference(self, feat):
            z, r = super().reference(feat)
            return r, z   # sorted names: "r", "z"

    assert _check(M(), {"feat": torch.randn(R, 64)}, {"z": 3, "r": 3}, R, 1e-4) >= 16


def test_traj_program_weights_follow_in_place_parameter_updates():
    """The blob is re-packed (one launch) when an optimiser has stepped: same addresses, new versions."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = _TwoHeads().to(dev)
    prog = m.program()
    x = torch.randn(9, 64, device=dev)
    a = prog.run_autograd({"feat": x}, {"z": 3, "r": 3}, 9)["z"].detach().clone()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.05 * torch.randn_like(p))
    b = prog.run_autograd({"feat": x}, {"z": 3, "r": 3}, 9)["z"]
    ref = m.reference(x)[0]
    assert float((b - ref).abs().max()) < 1e-5 and float((a - b).abs().max()) > 1e-4


@pytest.mark.parametrize("R,K", [(1, 512), (16, 8192), (37, 8192), (512, 8192), (100, 1024)])
def test_fc64_training_kernels_match_fp64(R, K):
    from multimodalfilter_amd import _abi

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(R + K)
    x, w, b = torch.randn(R, K, generator=g), torch.randn(64, K, generator=g) / K ** 0.5, torch.randn(64, generator=g)
    gy = torch.randn(R, 64, generator=g)
    y = torch.empty(R, 64, device=dev)
    xd, wd = x.to(dev), w.to(dev)
    _abi.fc64_train_forward(xd, wd, b.to(dev), y)
    assert _rel(y, x.double() @ w.double().t() + b.double()) < 1e-6
    dx, dw, db = torch.empty(R, K, device=dev), torch.empty(64, K, device=dev), torch.empty(64, device=dev)
    _abi.fc64_train_backward(gy.to(dev), xd, wd, dx, dw, db)
    assert _rel(dx, gy.double() @ w.double()) < 1e-6
    assert _rel(dw, gy.double().t() @ x.double()) < 1e-6
    assert _rel(db, gy.double().sum(0)) < 1e-6
    # twice the same bits: fixed summation order
    dw2 = torch.empty_like(dw)
    _abi.fc64_train_backward(gy.to(dev), xd, wd, None, dw2, None)
    assert torch.equal(dw, dw2)
