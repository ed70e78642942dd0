"""A minimal HDF5 reader (and the writer its tests need) on the HDF5 C library itself, bound with ctypes.

The reference reads its recordings with ``fannypack.data.TrajectoriesFile`` -- ``h5py`` on ``libhdf5``
(``/root/reference/crossmodal/tasks/_door.py:121-126``, ``tasks/_push.py``): one group per trajectory
(``trajectory0``, ``trajectory1``, ...), one dataset per key.  ``h5py`` is not installed in this image,
``libhdf5`` is: this module calls the same C API (``H5Fopen``, ``H5Literate``, ``H5Dread`` with the
dataset's native memory type, so contiguous, chunked and deflate-compressed numeric datasets all read),
which makes ``data.load_hdf5`` a REAL HDF5 read instead of one exercised only against a stand-in module.
Host-side, nothing here is on the timed path.
"""
import ctypes
import ctypes.util
import glob
import os
from ctypes import POINTER, c_char_p, c_int, c_int64, c_size_t, c_uint, c_ulonglong, c_void_p
from typing import Dict, List

import numpy as np

hid_t = c_int64          # HDF5 >= 1.10
_lib = None

H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5T_DIR_ASCEND = 1
H5O_TYPE_GROUP, H5O_TYPE_DATASET = 0, 1


def _find() -> str:
    cands = []
    for root in (os.environ.get("CONDA_PREFIX"), "/opt/conda", "/usr", "/usr/local"):
        if root:
            for sub in ("lib", "lib/x86_64-linux-gnu", "lib64"):
                cands += sorted(glob.glob(os.path.join(root, sub, "libhdf5.so*")))
                cands += sorted(glob.glob(os.path.join(root, sub, "libhdf5_serial.so*")))
    found = ctypes.util.find_library("hdf5")
    if found:
        cands.append(found)
    if not cands:
        raise ImportError("neither h5py nor a libhdf5 shared library is available")
    return cands[0]


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_find())
        sig = {
            "H5open": (c_int, []), "H5Fopen": (hid_t, [c_char_p, c_uint, hid_t]), "H5Fclose": (c_int, [hid_t]),
            "H5Fcreate": (hid_t, [c_char_p, c_uint, hid_t, hid_t]),
            "H5Gopen2": (hid_t, [hid_t, c_char_p, hid_t]), "H5Gclose": (c_int, [hid_t]),
            "H5Gcreate2": (hid_t, [hid_t, c_char_p, hid_t, hid_t, hid_t]),
            "H5Dopen2": (hid_t, [hid_t, c_char_p, hid_t]), "H5Dclose": (c_int, [hid_t]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
            "H5Dread": (c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p]),
            "H5Dcreate2": (hid_t, [hid_t, c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dwrite": (c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, c_void_p]),
            "H5Sget_simple_extent_ndims": (c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (c_int, [hid_t, POINTER(c_ulonglong), POINTER(c_ulonglong)]),
            "H5Screate_simple": (hid_t, [c_int, POINTER(c_ulonglong), POINTER(c_ulonglong)]), "H5Sclose": (c_int, [hid_t]),
            "H5Tget_class": (c_int, [hid_t]), "H5Tget_size": (c_size_t, [hid_t]), "H5Tget_sign": (c_int, [hid_t]),
            "H5Tget_native_type": (hid_t, [hid_t, c_int]), "H5Tclose": (c_int, [hid_t]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pset_chunk": (c_int, [hid_t, c_int, POINTER(c_ulonglong)]),
            "H5Pset_deflate": (c_int, [hid_t, c_uint]), "H5Pclose": (c_int, [hid_t]),
            "H5get_libversion": (c_int, [POINTER(c_uint), POINTER(c_uint), POINTER(c_uint)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.H5open() < 0:
            raise ImportError("H5open failed")
        maj, mnr, rel = c_uint(0), c_uint(0), c_uint(0)
        L.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
        if (maj.value, mnr.value) < (1, 10):
            # 1.8 has a 32-bit hid_t: every handle of this binding would be mis-sized
            raise ImportError(f"libhdf5 {maj.value}.{mnr.value}.{rel.value}: this binding needs HDF5 >= 1.10 (64-bit hid_t); install h5py instead")
        # HDF5 >= 1.12 turned H5Literate into a macro over H5Literate1 / H5Literate2: the library exports only those.
        # H5Literate1 has the 1.10 signature (its callback gets the H5L_info1_t this binding ignores anyway)
        it = getattr(L, "H5Literate", None) or getattr(L, "H5Literate1", None)
        if it is None:
            raise ImportError(f"libhdf5 {maj.value}.{mnr.value}.{rel.value} exports neither H5Literate nor H5Literate1")
        it.restype, it.argtypes = c_int, [hid_t, c_int, c_int, POINTER(c_ulonglong), c_void_p, c_void_p]
        L._mmf_iterate = it
        _lib = L
    return _lib


def _global(name: str) -> int:
    return hid_t.in_dll(lib(), name).value


_ITER_CB = ctypes.CFUNCTYPE(c_int, hid_t, c_char_p, c_void_p, c_void_p)


def _links(loc: int) -> List[str]:
    names: List[str] = []

    def cb(_group, name, _info, _data):
        names.append(name.decode())
        return 0

    idx = c_ulonglong(0)
    if lib()._mmf_iterate(loc, H5_INDEX_NAME, H5_ITER_INC, ctypes.byref(idx), ctypes.cast(_ITER_CB(cb), c_void_p), None) < 0:
        raise OSError("H5Literate failed")
    return names


def _read_dataset(loc: int, name: str) -> np.ndarray:
    L = lib()
    d = L.H5Dopen2(loc, name.encode(), 0)
    if d < 0:
        raise OSError(f"{name}: not a dataset")
    space = ftype = mtype = -1
    try:
        space, ftype = L.H5Dget_space(d), L.H5Dget_type(d)
        nd = L.H5Sget_simple_extent_ndims(space)
        dims = (c_ulonglong * max(nd, 1))()
        if nd > 0:
            L.H5Sget_simple_extent_dims(space, dims, None)
        mtype = L.H5Tget_native_type(ftype, H5T_DIR_ASCEND)
        cls, size, sign = L.H5Tget_class(mtype), L.H5Tget_size(mtype), L.H5Tget_sign(mtype)
        if cls == H5T_FLOAT:
            dtype = {2: np.float16, 4: np.float32, 8: np.float64}[size]
        elif cls == H5T_INTEGER:
            dtype = np.dtype(f"{'i' if sign else 'u'}{size}")
        else:
            raise TypeError(f"{name}: only numeric datasets are supported (HDF5 type class {cls})")
        out = np.empty(tuple(int(x) for x in dims[:nd]), dtype=dtype)
        if L.H5Dread(d, mtype, 0, 0, 0, out.ctypes.data_as(c_void_p)) < 0:
            raise OSError(f"{name}: H5Dread failed")
        return out
    finally:  # every id, on the error paths too
        if mtype >= 0:
            L.H5Tclose(mtype)
        if ftype >= 0:
            L.H5Tclose(ftype)
        if space >= 0:
            L.H5Sclose(space)
        L.H5Dclose(d)


def read_groups(path: str) -> Dict[str, Dict[str, np.ndarray]]:
    """``{group name: {dataset name: array}}`` for every top-level group of the file."""
    L = lib()
    f = L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, 0)
    if f < 0:
        raise OSError(f"cannot open {path} as HDF5")
    try:
        out = {}
        for g in _links(f):
            gid = L.H5Gopen2(f, g.encode(), 0)
            if gid < 0:
                continue  # a top-level dataset, not a trajectory group
            try:
                out[g] = {k: _read_dataset(gid, k) for k in _links(gid)}
            finally:
                L.H5Gclose(gid)
        return out
    finally:
        L.H5Fclose(f)


def write_groups(path: str, groups: Dict[str, Dict[str, np.ndarray]], *, compress: bool = False) -> None:
    """Write ``{group: {dataset: array}}`` (tests: a recording in ``TrajectoriesFile`` layout).  ``compress``:
    chunked + deflate, as ``fannypack.data.TrajectoriesFile(compress=True)`` stores its datasets."""
    L = lib()
    native = {np.dtype(np.float32): "H5T_NATIVE_FLOAT_g", np.dtype(np.float64): "H5T_NATIVE_DOUBLE_g",
              np.dtype(np.uint8): "H5T_NATIVE_UCHAR_g", np.dtype(np.int64): "H5T_NATIVE_LLONG_g",
              np.dtype(np.int32): "H5T_NATIVE_INT_g"}
    f = L.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, 0, 0)
    if f < 0:
        raise OSError(f"cannot create {path}")
    try:
        for g, datasets in groups.items():
            gid = L.H5Gcreate2(f, g.encode(), 0, 0, 0)
            for k, a in datasets.items():
                a = np.ascontiguousarray(a)
                t = _global(native[a.dtype])
                dims = (c_ulonglong * a.ndim)(*a.shape)
                space = L.H5Screate_simple(a.ndim, dims, None)
                dcpl = 0
                if compress and a.size > 0:
                    dcpl = L.H5Pcreate(_global("H5P_CLS_DATASET_CREATE_ID_g"))
                    chunk = (c_ulonglong * a.ndim)(*[min(s, 64) if i == 0 else s for i, s in enumerate(a.shape)])
                    L.H5Pset_chunk(dcpl, a.ndim, chunk)
                    L.H5Pset_deflate(dcpl, 4)
                d = L.H5Dcreate2(gid, k.encode(), t, space, 0, dcpl, 0)
                if d < 0 or L.H5Dwrite(d, t, 0, 0, 0, a.ctypes.data_as(c_void_p)) < 0:
                    raise OSError(f"writing {g}/{k} failed")
                L.H5Dclose(d); L.H5Sclose(space)
                if dcpl:
                    L.H5Pclose(dcpl)
            L.H5Gclose(gid)
    finally:
        L.H5Fclose(f)
