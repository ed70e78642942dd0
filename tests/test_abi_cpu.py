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


def test_persistent_loop_sizing_needs_no_gpu():
    """The persistent loops' host-side sizing: workspace words follow the documented layout; without a device no problem is
    eligible (the plan asks the runtime for the CU count) and nothing crashes; nonsense arguments are refused."""
    from multimodalfilter_amd import _abi

    assert _abi.pf_persistent_sync_words(32, 300, 3, 2) == 4 + 2 * (2 * 32 * 300 * 3 + 2 * 32 * 300)
    assert _abi.ekf_persistent_sync_words(32, 2, 3) == 4 + 2 * (2 * 2 * 32 * (3 + 9))
    assert _abi.ekf_persistent_sync_words(0, 2, 3) == 0
    if not torch.cuda.is_available():
        assert _abi.pf_persistent_plan(32, 300, 2) == 0 and _abi.ekf_persistent_plan(32, 2) == 0
    assert _abi.ekf_persistent_plan(0, 2) < 0 and _abi.ekf_persistent_plan(32, 9) < 0  # MMF_EINVAL


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
    # round 5's training entry points: host tensors are refused before any launch, there is no torch fallback behind them
    from multimodalfilter_amd import engine

    with pytest.raises(_abi.MmfError):
        engine.Fc64Function.apply(torch.zeros(4, 8192), torch.zeros(64, 8192), torch.zeros(64))
    with pytest.raises(_abi.MmfError):
        dyn.encode_controls(torch.zeros(2, 7))
    assert dyn._ctrl_prog is not None  # the program is built before the tensors are looked at
    with pytest.raises(_abi.MmfError):
        dyn._ctrl_prog.run_autograd({"controls": torch.zeros(2, 7)}, {"bias": 64}, 2)


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


def test_every_entry_point_rejects_null_arguments():
    """Error behaviour of the boundary, uniformly: every ``int``-returning entry point called with null pointers
    and zero sizes returns ``MMF_EINVAL`` (-1) before any HIP call (so this runs without a GPU) -- no crash, no
    launch.  The reference raises ``AssertionError`` on bad shapes; the host side turns a non-zero
    return into ``MmfError``."""
    import ctypes

    from multimodalfilter_amd import _abi

    lib = _abi.load()
    checked = 0
    for name, (res, args) in _abi.SIGNATURES.items():
        if res is not ctypes.c_int or not args:
            continue
        vals = [0 if a in (ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_size_t) else
                (0.0 if a is ctypes.c_float else None) for a in args]
        assert getattr(lib, name)(*vals) == -1, name
        checked += 1
    assert checked >= 35


def test_size_limits_and_empty_batches_at_the_boundary():
    """Sizes beyond what a kernel indexes -> ``MMF_ETOOLARGE`` (-2); unsupported ``d`` / precision / aliasing ->
    ``MMF_EINVAL``; an EMPTY batch (N = 0 trajectories, R = 0 rows) is a successful no-op.  All decided on the
    host before any HIP call: the pointers here are never dereferenced."""
    import ctypes

    from multimodalfilter_amd import _abi

    lib = _abi.load()
    bufs = [(ctypes.c_float * 16)() for _ in range(9)]
    P = [ctypes.cast(b, ctypes.c_void_p) for b in bufs]
    F32 = _abi.PREC_F32
    EINVAL, ETOOLARGE = -1, -2
    # too large
    assert lib.mmf_pf_measure(P[0], 2, F32, P[1], P[2], None, 0, P[3], 0, None, 1 << 20, 1 << 12, 3, None) == ETOOLARGE
    assert lib.mmf_pf_dynamics(P[0], 3, F32, P[1], P[2], P[3], P[4], P[5], None, 1 << 20, 1 << 12, 3, None) == ETOOLARGE
    assert lib.mmf_dynamics_jacobian(P[0], 3, F32, P[1], P[2], P[3], P[4], None, 1 << 30, 3, None) == ETOOLARGE
    # systematic resampling keeps an 8-byte CDF entry per particle in the 160 KiB LDS: 20,400 particles fit, 20,480 do not
    rs = lambda M, mode, out: lib.mmf_pf_reweight_resample(P[0], P[1], P[2], P[3], P[4], out, P[6], None, 4, M, M, 3, mode, None)
    assert lib.mmf_pf_reweight_resample_lds_bytes(20400, 1) <= 160 * 1024 < lib.mmf_pf_reweight_resample_lds_bytes(20480, 1)
    assert rs(20480, 1, P[5]) == ETOOLARGE and rs(41000, 0, P[1]) == ETOOLARGE and rs(65537, 0, P[1]) == ETOOLARGE
    assert rs(64, 1, P[2]) == EINVAL            # in-place gather
    assert lib.mmf_pf_reweight_resample(P[0], P[1], P[2], P[3], P[4], P[5], P[6], None, 4, 64, 64, 5, 1, None) == EINVAL  # d > 4
    # unsupported precision code / state dimension
    assert lib.mmf_pf_measure(P[0], 2, 77, P[1], P[2], None, 0, P[3], 0, None, 4, 64, 3, None) == EINVAL
    assert lib.mmf_pf_measure(P[0], 2, F32, P[1], P[2], None, 0, P[3], 0, None, 4, 64, 5, None) == EINVAL
    # empty batches
    assert lib.mmf_pf_measure(P[0], 2, F32, P[1], P[2], None, 0, P[3], 0, None, 0, 64, 3, None) == 0
    assert lib.mmf_pf_dynamics(P[0], 3, F32, P[1], P[2], P[3], P[4], P[5], None, 0, 64, 3, None) == 0
    assert lib.mmf_dynamics_jacobian(P[0], 3, F32, P[1], P[2], P[3], P[4], None, 0, 3, None) == 0
    assert lib.mmf_image_encoder((ctypes.c_void_p * 1)(P[0]), 1, P[1], P[2], P[3], None, F32, 0, 0, None) == 0
    assert lib.mmf_traj_program(P[0], 1, P[1], (ctypes.c_void_p * 8)(), 0, 1, 64, None) == 0
    # round 5's entry points: contraction length of the 8192 -> 64 layer, row slices, networks per launch, state dimension
    assert lib.mmf_fc64_train_forward(P[0], P[1], P[2], P[3], P[4], 4, 8192 + 256, None) == EINVAL      # K % 512
    assert lib.mmf_fc64_train_forward(P[0], P[1], P[2], P[3], P[4], 0, 8192, None) == 0               # no rows
    assert lib.mmf_fc64_train_backward(P[0], P[1], P[2], P[3], P[4], P[5], 4, 100, None) == EINVAL
    assert lib.mmf_traj_weight_grads(P[0], 1, P[1], 8, P[2], 8, P[3], 16, P[4], 65, 4, None) == EINVAL  # > 64 row slices
    assert lib.mmf_traj_weight_grads(P[0], 1, P[1], 8, P[2], 8, P[3], 16, None, 2, 4, None) == EINVAL   # slices without scratch
    assert lib.mmf_traj_pack(P[0], 0, P[1], None) == EINVAL
    nets = (_abi.MmfTrainFusedArgs * 5)()
    assert lib.mmf_particle_net_train_fused_multi(nets, 5, None) == EINVAL                               # > MMF_LOOP_MAX_MEAS
    assert lib.mmf_particle_net_train_fused_multi(nets, 2, None) == EINVAL                               # null fields
    four = (ctypes.c_void_p * 5)(*[P[0]] * 5)
    assert lib.mmf_pf_measure_multi(four, 5, 2, F32, P[1], four, four, 0, four, None, 4, 64, 3, None) == EINVAL
    assert lib.mmf_pf_measure_multi(four, 2, 2, F32, P[1], four, four, 0, four, None, 0, 64, 3, None) == 0  # empty batch
    fa = _abi.MmfPfTrainFinalizeArgs()
    for k in ("pw", "pb", "p_first", "p_head", "p_dout", "p_traj", "grads", "bias_grad", "scratch"):
        setattr(fa, k, P[0])
    fa.T, fa.N, fa.SL, fa.S, fa.n_res, fa.d, fa.n_out, fa.join_in, fa.join_state_off = 2, 2, 1, 1, 2, 4, 1, 128, 64
    assert lib.mmf_pf_train_finalize(ctypes.byref(fa), None) == EINVAL                                   # d > 3
    fa.d, fa.join_state_off = 3, 100
    assert lib.mmf_pf_train_finalize(ctypes.byref(fa), None) == EINVAL                                   # state columns outside the join layer


def test_every_host_struct_of_the_binding_matches_the_header_field_by_field(tmp_path):
    """The ctypes structures of ``_abi.py`` against ``include/mmf.h`` as a C compiler lays it out: a small C program
    (gcc, the header compiled as plain C) prints ``sizeof`` and the ``offsetof`` of every field, which must equal
    ctypes' -- a field added to one side only, or in another order, fails here instead of corrupting a launch."""
    import shutil
    import subprocess

    from multimodalfilter_amd import _abi

    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    structs = ["MmfParticleNetDesc", "MmfImageEncoderDesc", "MmfPfLoopArgs", "MmfTrainNet", "MmfPfTrainArgs", "MmfTrajInstr",
               "MmfEkfLoopArgs"]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "mmf.h")}"', "int main(void) {"]
    for name in structs:
        cls = getattr(_abi, name)
        lines.append(f'  printf("{name} size %zu\\n", sizeof({name}));')
        for field, _t in cls._fields_:
            lines.append(f'  printf("{name} {field} %zu\\n", offsetof({name}, {field}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    out = subprocess.run([gcc, "-std=c99", "-o", str(exe), str(src)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]  # a field the binding names and the header lacks does not compile
    got = {}
    for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines():
        name, field, value = line.split()
        got[(name, field)] = int(value)
    for name in structs:
        cls = getattr(_abi, name)
        assert got[(name, "size")] == ctypes.sizeof(cls), (name, got[(name, "size")], ctypes.sizeof(cls))
        for field, _t in cls._fields_:
            assert got[(name, field)] == getattr(cls, field).offset, (name, field)
    # and the header has no field the binding lacks: sizes are equal and every binding field sits where C puts it,
    # so an extra C field would have to hide in padding -- the structs are checked to have none at their end
    for name in structs:
        cls = getattr(_abi, name)
        last, last_t = cls._fields_[-1]
        assert getattr(cls, last).offset + ctypes.sizeof(last_t) + 8 > ctypes.sizeof(cls), name


def test_every_entry_point_of_the_header_cites_what_it_replaces():
    """The boundary contract: each entry point of ``include/mmf.h`` stands in a section whose comment names the
    reference interface it replaces -- a ``file.py:line`` under ``/root/reference`` or, for the recursion the reference
    imports, ``torchfilter`` (un-vendored: SURVEY appendix A.2)."""
    with open(os.path.join(ROOT, "include", "mmf.h")) as fh:
        text = fh.read()
    decls = list(re.finditer(r"^(?:int|size_t|const char\*|void)\s+(mmf_\w+)\s*\(", text, re.M))
    assert len(decls) >= 46
    for m in decls:
        if m.group(1) == "mmf_version":
            continue
        start = text.rfind("/* ----", 0, m.start())
        assert start >= 0, m.group(1)
        section = text[start:m.start()]
        assert re.search(r"[\w/]+\.py:\d+", section) or "torchfilter" in section, m.group(1)
