"""TEST-ONLY reader of the reference's HDF5 files through ctypes on the image's libhdf5 (no h5py here): datasets as numpy
arrays, root attributes "time step" / "total time" / "ghost zone included"."""
import ctypes as C
import os

import numpy as np

CANDIDATES = [os.environ.get("RGPU_HDF5_LIB", ""), "libhdf5.so", "libhdf5.so.103", "/opt/conda/lib/libhdf5.so.103"]
_lib = None


def lib():
    global _lib
    if _lib is None:
        for name in CANDIDATES:
            if not name:
                continue
            try:
                _lib = C.CDLL(name)
                break
            except OSError:
                continue
        if _lib is None:
            return None
        hid = C.c_int64
        L = _lib
        L.H5open.restype = C.c_int
        L.H5Fopen.restype = hid; L.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
        L.H5Fclose.argtypes = [hid]
        L.H5Dopen2.restype = hid; L.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
        L.H5Dget_space.restype = hid; L.H5Dget_space.argtypes = [hid]
        L.H5Sget_simple_extent_ndims.argtypes = [hid]
        L.H5Sget_simple_extent_dims.argtypes = [hid, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
        L.H5Sclose.argtypes = [hid]
        L.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
        L.H5Dclose.argtypes = [hid]
        L.H5Aopen.restype = hid; L.H5Aopen.argtypes = [hid, C.c_char_p, hid]
        L.H5Aread.argtypes = [hid, hid, C.c_void_p]
        L.H5Aclose.argtypes = [hid]
        L.H5Lexists.argtypes = [hid, C.c_char_p, hid]
        L.H5open()
    return _lib


def available():
    return lib() is not None


NAMES = ["density", "energy", "momentum_x", "momentum_y", "momentum_z", "magnetic_field_x", "magnetic_field_y", "magnetic_field_z"]


def read(path):
    """-> (dict name -> array, dict of attributes)"""
    L = lib()
    hid = C.c_int64
    f64 = hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
    i32 = hid.in_dll(L, "H5T_NATIVE_INT_g").value
    f = L.H5Fopen(path.encode(), 0, 0)
    assert f >= 0, path
    out = {}
    for n in NAMES:
        if L.H5Lexists(f, n.encode(), 0) <= 0:
            continue
        d = L.H5Dopen2(f, ("/" + n).encode(), 0)
        sp = L.H5Dget_space(d)
        rank = L.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_ulonglong * rank)()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        a = np.empty(tuple(int(x) for x in dims), dtype=np.float64)
        assert L.H5Dread(d, f64, 0, 0, 0, a.ctypes.data) >= 0
        L.H5Sclose(sp); L.H5Dclose(d)
        out[n] = a
    attrs = {}
    for name, ct, ty in (("time step", C.c_int, i32), ("total time", C.c_double, f64), ("ghost zone included", C.c_int, i32),
                         ("nx", C.c_int, i32), ("ny", C.c_int, i32), ("nz", C.c_int, i32)):
        a = L.H5Aopen(f, name.encode(), 0)
        assert a >= 0, name
        v = ct()
        L.H5Aread(a, ty, C.byref(v))
        L.H5Aclose(a)
        attrs[name] = v.value
    L.H5Fclose(f)
    return out, attrs
