"""TEST INFRASTRUCTURE — ctypes loader for the CPU oracle (oracle/interpn_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; nothing under interpn_amd/ does.  The oracle restates the reference's algorithm
(jlogan03/interpn v0.8.2, src/multilinear/*.rs, src/multicubic/*.rs); see the header of
interpn_oracle.cpp for the pinning status.
"""

from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_double, c_float, c_int, c_size_t, c_uint8, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_interpn.so")

UNREPRESENTABLE = 7


class OracleError(AssertionError):
    """Mirrors PyAssertionError(msg) raised by src/python.rs:77-79."""

    def __init__(self, status: int, msg: str, first_bad: int | None = None):
        super().__init__(msg)
        self.status = status
        self.first_bad = first_bad


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "interpn_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_interpn.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("INTERPN_ORACLE_LIB")  # e.g. an ASan/UBSan build of the same source
        if override:
            _lib = ctypes.CDLL(override)
        else:
            build()
            _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_strerror.restype = ctypes.c_char_p
        _lib.oracle_strerror.argtypes = [c_int]
    return _lib


def _ct(dtype):
    if dtype == np.float64:
        return c_double, "f64"
    if dtype == np.float32:
        return c_float, "f32"
    raise TypeError(f"Unexpected data type: {dtype}")


def _arr(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _ptrs(arrs, ct):
    n = len(arrs)
    ptrs = (POINTER(ct) * max(n, 1))()
    lens = (c_size_t * max(n, 1))()
    for i, a in enumerate(arrs):
        ptrs[i] = a.ctypes.data_as(POINTER(ct))
        lens[i] = a.size
    return ptrs, lens


def _finish(status, first_bad):
    if status != 0:
        msg = lib().oracle_strerror(status).decode()
        raise OracleError(status, msg, first_bad.value if status == UNREPRESENTABLE else None)


def _sizes(dims):
    d = (c_size_t * max(len(dims), 1))()
    for i, v in enumerate(dims):
        d[i] = int(v)
    return d


def linear_regular(dims, starts, steps, vals, obs, out, fma=True):
    dtype = out.dtype
    ct, sfx = _ct(dtype)
    starts, steps, vals = _arr(starts, dtype), _arr(steps, dtype), _arr(vals, dtype)
    obs = [_arr(o, dtype) for o in obs]
    optr, olen = _ptrs(obs, ct)
    fb = c_size_t(0)
    fn = getattr(lib(), f"oracle_linear_regular_{sfx}")
    st = fn(c_int(int(fma)), _sizes(dims), c_size_t(len(dims)), starts.ctypes.data_as(POINTER(ct)),
            c_size_t(starts.size), steps.ctypes.data_as(POINTER(ct)), c_size_t(steps.size),
            vals.ctypes.data_as(POINTER(ct)), c_size_t(vals.size), optr, olen, c_size_t(len(obs)),
            out.ctypes.data_as(POINTER(ct)), c_size_t(out.size), ctypes.byref(fb))
    _finish(st, fb)
    return out


def linear_rectilinear(grids, vals, obs, out, fma=True):
    dtype = out.dtype
    ct, sfx = _ct(dtype)
    grids = [_arr(g, dtype) for g in grids]
    vals = _arr(vals, dtype)
    obs = [_arr(o, dtype) for o in obs]
    gptr, glen = _ptrs(grids, ct)
    optr, olen = _ptrs(obs, ct)
    fb = c_size_t(0)
    fn = getattr(lib(), f"oracle_linear_rectilinear_{sfx}")
    st = fn(c_int(int(fma)), gptr, glen, c_size_t(len(grids)), vals.ctypes.data_as(POINTER(ct)),
            c_size_t(vals.size), optr, olen, c_size_t(len(obs)), out.ctypes.data_as(POINTER(ct)),
            c_size_t(out.size), ctypes.byref(fb))
    _finish(st, fb)
    return out


def cubic_regular(dims, starts, steps, vals, linearize_extrapolation, obs, out, fma=True):
    dtype = out.dtype
    ct, sfx = _ct(dtype)
    starts, steps, vals = _arr(starts, dtype), _arr(steps, dtype), _arr(vals, dtype)
    obs = [_arr(o, dtype) for o in obs]
    optr, olen = _ptrs(obs, ct)
    fb = c_size_t(0)
    fn = getattr(lib(), f"oracle_cubic_regular_{sfx}")
    st = fn(c_int(int(fma)), _sizes(dims), c_size_t(len(dims)), starts.ctypes.data_as(POINTER(ct)),
            c_size_t(starts.size), steps.ctypes.data_as(POINTER(ct)), c_size_t(steps.size),
            vals.ctypes.data_as(POINTER(ct)), c_size_t(vals.size), c_int(int(bool(linearize_extrapolation))),
            optr, olen, c_size_t(len(obs)), out.ctypes.data_as(POINTER(ct)), c_size_t(out.size),
            ctypes.byref(fb))
    _finish(st, fb)
    return out


def cubic_rectilinear(grids, vals, linearize_extrapolation, obs, out, fma=True):
    dtype = out.dtype
    ct, sfx = _ct(dtype)
    grids = [_arr(g, dtype) for g in grids]
    vals = _arr(vals, dtype)
    obs = [_arr(o, dtype) for o in obs]
    gptr, glen = _ptrs(grids, ct)
    optr, olen = _ptrs(obs, ct)
    fb = c_size_t(0)
    fn = getattr(lib(), f"oracle_cubic_rectilinear_{sfx}")
    st = fn(c_int(int(fma)), gptr, glen, c_size_t(len(grids)), vals.ctypes.data_as(POINTER(ct)),
            c_size_t(vals.size), c_int(int(bool(linearize_extrapolation))), optr, olen, c_size_t(len(obs)),
            out.ctypes.data_as(POINTER(ct)), c_size_t(out.size), ctypes.byref(fb))
    _finish(st, fb)
    return out


def nearest_regular(dims, starts, steps, vals, obs, out, fma=True):
    dtype = out.dtype
    ct, sfx = _ct(dtype)
    starts, steps, vals = _arr(starts, dtype), _arr(steps, dtype), _arr(vals, dtype)
    obs = [_arr(o, dtype) for o in obs]
    optr, olen = _ptrs(obs, ct)
    fb = c_size_t(0)
    fn = getattr(lib(), f"oracle_nearest_regular_{sfx}")
    st = fn(c_int(int(fma)), _sizes(dims), c_size_t(len(dims)), starts.ctypes.data_as(POINTER(ct)),
            c_size_t(starts.size), steps.ctypes.data_as(POINTER(ct)), c_size_t(steps.size),
            vals.ctypes.data_as(POINTER(ct)), c_size_t(vals.size), optr, olen, c_size_t(len(obs)),
            out.ctypes.data_as(POINTER(ct)), c_size_t(out.size), ctypes.byref(fb))
    _finish(st, fb)
    return out


def nearest_rectilinear(grids, vals, obs, out, fma=True):
    dtype = out.dtype
    ct, sfx = _ct(dtype)
    grids = [_arr(g, dtype) for g in grids]
    vals = _arr(vals, dtype)
    obs = [_arr(o, dtype) for o in obs]
    gptr, glen = _ptrs(grids, ct)
    optr, olen = _ptrs(obs, ct)
    fb = c_size_t(0)
    fn = getattr(lib(), f"oracle_nearest_rectilinear_{sfx}")
    st = fn(c_int(int(fma)), gptr, glen, c_size_t(len(grids)), vals.ctypes.data_as(POINTER(ct)),
            c_size_t(vals.size), optr, olen, c_size_t(len(obs)), out.ctypes.data_as(POINTER(ct)),
            c_size_t(out.size), ctypes.byref(fb))
    _finish(st, fb)
    return out


def check_bounds_regular(dims, starts, steps, obs, atol, out):
    dtype = np.asarray(starts).dtype
    ct, sfx = _ct(dtype)
    starts, steps = _arr(starts, dtype), _arr(steps, dtype)
    obs = [_arr(o, dtype) for o in obs]
    optr, olen = _ptrs(obs, ct)
    flags = np.zeros(len(out), dtype=np.uint8)
    fn = getattr(lib(), f"oracle_check_bounds_regular_{sfx}")
    st = fn(_sizes(dims), c_size_t(len(dims)), starts.ctypes.data_as(POINTER(ct)),
            steps.ctypes.data_as(POINTER(ct)), optr, olen, c_size_t(len(obs)), ct(float(atol)),
            flags.ctypes.data_as(POINTER(c_uint8)), c_size_t(flags.size))
    _finish(st, c_size_t(0))
    out[:] = flags.astype(bool)
    return out


def check_bounds_rectilinear(grids, obs, atol, out):
    dtype = np.asarray(grids[0]).dtype
    ct, sfx = _ct(dtype)
    grids = [_arr(g, dtype) for g in grids]
    obs = [_arr(o, dtype) for o in obs]
    gptr, glen = _ptrs(grids, ct)
    optr, olen = _ptrs(obs, ct)
    flags = np.zeros(len(out), dtype=np.uint8)
    fn = getattr(lib(), f"oracle_check_bounds_rectilinear_{sfx}")
    st = fn(gptr, glen, c_size_t(len(grids)), optr, olen, c_size_t(len(obs)), ct(float(atol)),
            flags.ctypes.data_as(POINTER(c_uint8)), c_size_t(flags.size))
    _finish(st, c_size_t(0))
    out[:] = flags.astype(bool)
    return out
