"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol that
``include/mmf.h`` declares; the product path refuses to run without device memory."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mmf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from multimodalfilter_amd import _abi, build

    build.build()
    lib = ctypes.CDLL(_abi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 9
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mmf.h but not exported"
    assert set(declared) == set(_abi.SIGNATURES), "ctypes binding and header disagree"
    assert _abi.load().mmf_version() == _abi.ABI_VERSION


def test_size_queries_need_no_gpu():
    from multimodalfilter_amd import _abi

    lib = _abi.load()
    assert lib.mmf_particle_net_floats(3) * 4 <= 160 * 1024  # a whole network fits LDS
    assert lib.mmf_particle_net_floats(2) < lib.mmf_particle_net_floats(3)
    assert lib.mmf_particle_net_floats(4) == 0
    assert lib.mmf_pf_reweight_resample_lds_bytes(4096, 1) >= 4096 * 8
    assert lib.mmf_pf_reweight_resample_lds_bytes(4096, 0) >= 4096 * 4


def test_descriptor_layout_matches_header():
    from multimodalfilter_amd import _abi

    # 6 int32 + (2 + 2 + 2 + 1 + 6 + 6 + 2) pointers
    assert ctypes.sizeof(_abi.MmfParticleNetDesc) == 6 * 4 + 21 * 8


def test_product_path_refuses_cpu_tensors():
    import multimodalfilter_amd as mmf
    from multimodalfilter_amd import _abi

    pf = mmf.door_models.DoorParticleFilter()
    pf.eval()
    with pytest.raises(_abi.MmfError):
        pf.initialize_beliefs(mean=torch.zeros(2, 3), covariance=torch.eye(3)[None].expand(2, 3, 3))
    dyn = mmf.door_models.DoorDynamicsModel()
    with pytest.raises(_abi.MmfError):
        dyn(initial_states=torch.zeros(2, 3), controls=torch.zeros(2, 7))
    with pytest.raises(_abi.MmfError):
        _abi.ptr(torch.zeros(3))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from multimodalfilter_amd import _abi

    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "LIB_PATH", str(tmp_path / "libmmf_hip.so"))
    with pytest.raises(_abi.MmfError, match="no CPU fallback"):
        _abi.load()


def test_model_registry_and_state_dict_keys_match_oracle():
    import multimodalfilter_amd as mmf
    from oracle import models as om

    for task in ("door", "push"):
        for name, cls in mmf.model_types(task).items():
            assert cls.__name__ == name
            assert set(cls().state_dict()) == set(om.build(name).state_dict()), name
