"""Host-side logic that needs no GPU: noise sources, struct layouts of the step-loop arguments,
the training entry points' loud failures."""
import ctypes

import pytest
import torch

import multimodalfilter_amd as mmf
from multimodalfilter_amd import _abi, engine, train


def test_stacked_noise_serves_steps_in_replay_order():
    """``StackedNoise`` hands out the same tensors, in the same order, as ``ReplayNoise`` fed
    the same draws one by one -- and ``draw_steps`` returns zero-copy views of its blocks."""
    g = torch.Generator().manual_seed(0)
    N, M, d, T = 3, 5, 2, 4
    eps0 = torch.randn((N, M, d), generator=g)
    eps = torch.randn((T, N, M, d), generator=g)
    u = torch.rand((T, N), generator=g)
    like = torch.zeros(1)
    a = mmf.StackedNoise(eps0, eps, u)
    b = mmf.ReplayNoise([eps0] + list(eps), list(u))
    assert torch.equal(a.gaussian((N, M, d), like=like), b.gaussian((N, M, d), like=like))
    for _ in range(2):
        assert torch.equal(a.gaussian((N, M, d), like=like), b.gaussian((N, M, d), like=like))
        assert torch.equal(a.uniform((N,), like=like), b.uniform((N,), like=like))
    gs, us = a.draw_steps(2, (N, M, d), (N,), like=like)
    assert torch.equal(gs, eps[2:4]) and torch.equal(us, u[2:4])
    assert gs.data_ptr() == eps[2].data_ptr()
    gs2, us2 = b.draw_steps(2, (N, M, d), (N,), like=like)
    assert torch.equal(gs2, gs) and torch.equal(us2, us)


def test_stacked_noise_checks_shapes():
    a = mmf.StackedNoise(torch.zeros(2, 3, 2), torch.zeros(1, 2, 3, 2), torch.zeros(1, 2))
    with pytest.raises(AssertionError):
        a.gaussian((2, 4, 2), like=torch.zeros(1))


def test_loop_argument_structs_match_the_header():
    """Field order / sizes of the host structs handed to the native step loops."""
    P, I = ctypes.sizeof(ctypes.c_void_p), 4
    # + feedback_gate (ABI 33); + persistent, n_sync_words (two int32 = one pointer slot), sync_words (ABI 40)
    assert ctypes.sizeof(_abi.MmfEkfLoopArgs) == 8 * I + (2 * _abi.LOOP_MAX_MEAS + 12) * P + 2 * I + P
    assert _abi.MmfEkfLoopArgs.sync_words.offset == _abi.MmfEkfLoopArgs.persistent.offset + 2 * I
    pf = _abi.MmfPfLoopArgs
    assert pf.T.offset == 0 and pf.dyn_packed.offset == 10 * I
    assert pf.event_stride.offset + I <= ctypes.sizeof(pf)
    assert [n for n, _ in _abi.MmfEkfLoopArgs._fields_][:9] == ["T", "N", "d", "K", "fusion", "feedback", "n_res_dyn",
                                                                "precision", "range_flag"]
    assert _abi.MmfEkfLoopArgs.range_flag.offset == 8 * I  # eight int32 fields, then pointers


def test_training_entry_points_fail_loudly_without_backend_or_gpu():
    f = mmf.door_models.DoorParticleFilter().train()
    batch = {"states": torch.zeros(3, 2, 3), "controls": torch.zeros(3, 2, 7), "image": torch.zeros(3, 2, 32, 32),
             "gripper_pos": torch.zeros(3, 2, 3), "gripper_sensors": torch.zeros(3, 2, 7)}
    engine.set_training_backend(None)
    with pytest.raises(AssertionError):
        train.filter_loss(f, batch, initial_covariance=torch.eye(3) * 0.1)
    engine.set_training_backend("hip")
    try:
        with pytest.raises(_abi.MmfError):  # CPU tensors: the HIP path refuses them, nothing falls back
            train.filter_loss(f, batch, initial_covariance=torch.eye(3) * 0.1)
    finally:
        engine.set_training_backend(None)
    with pytest.raises(AssertionError):
        engine.set_training_backend("cpu")
