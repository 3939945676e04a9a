"""Raw bindings with the names, argument order and error behaviour of the reference's
`interpn.raw` module (src/interpn/raw.py:6-42, signatures src/interpn/raw.pyi:32-147, PyO3
bodies src/python.rs:55-292) — evaluated on the MI355X through libinterpn_hip.so.

Arrays must be 1-D, C-contiguous numpy arrays of the function's dtype, as the reference
requires (`PyReadonlyArray1<T>` + `as_slice()?`, src/python.rs:49,71-75).  `obs` is a list with
one array per dimension; `out` is written in place.  Reference errors raise
AssertionError(message) with the reference's message text.
"""

from __future__ import annotations

import ctypes
from ctypes import POINTER, c_double, c_float, c_size_t, c_uint8

import numpy as np

from . import _lib

MAXDIMS = 8  # src/python.rs:10


def _ctype(dtype):
    return c_double if dtype == np.float64 else c_float


def _check_arr(name, a, dtype, writable=False):
    if not isinstance(a, np.ndarray):
        raise TypeError(f"argument '{name}': expected a numpy array, got {type(a).__name__}")
    if a.dtype != dtype:
        raise TypeError(f"argument '{name}': expected dtype {np.dtype(dtype).name}, got {a.dtype.name}")
    if a.ndim != 1:
        raise TypeError(f"argument '{name}': expected a 1-D array, got {a.ndim}-D")
    if not a.flags.c_contiguous:
        # numpy::NotContiguousError in the reference (as_slice() on a strided view)
        raise ValueError(f"argument '{name}': The given array is not contiguous")
    if writable and not a.flags.writeable:
        raise ValueError(f"argument '{name}': array is read-only")
    return a


def _slice_of_slices(name, arrs, dtype):
    arrs = list(arrs)
    ct = _ctype(dtype)
    n = len(arrs)
    if n > MAXDIMS:
        # unpack_vec_of_arr! writes into a [&[T]; MAXDIMS] (src/python.rs:46-50): index out of bounds
        raise _lib.ReferencePanic(f"argument '{name}': more than {MAXDIMS} arrays")
    ptrs = (POINTER(ct) * max(n, 1))()
    lens = (c_size_t * max(n, 1))()
    keep = []
    for i, a in enumerate(arrs):
        a = _check_arr(f"{name}[{i}]", a, dtype)
        keep.append(a)
        ptrs[i] = a.ctypes.data_as(POINTER(ct))
        lens[i] = a.size
    return ptrs, lens, n, keep


def _dims(dims):
    d = [int(v) for v in dims]
    if any(v < 0 for v in d):
        raise OverflowError("can't convert negative int to unsigned")  # Vec<usize> extraction
    arr = (c_size_t * max(len(d), 1))()
    for i, v in enumerate(d):
        arr[i] = v
    return arr, len(d)


def _ptr(a, dtype):
    return a.ctypes.data_as(POINTER(_ctype(dtype)))


def _linear_regular(dtype, sfx, dims, starts, steps, vals, obs, out, fam="linear"):
    lib = _lib.load()
    d, nd = _dims(dims)
    starts = _check_arr("starts", starts, dtype)
    steps = _check_arr("steps", steps, dtype)
    vals = _check_arr("vals", vals, dtype)
    out = _check_arr("out", out, dtype, writable=True)
    optr, olen, nobs, _keep = _slice_of_slices("obs", obs, dtype)
    st = getattr(lib, f"interpn_hip_{fam}_regular_{sfx}")(
        d, nd, _ptr(starts, dtype), starts.size, _ptr(steps, dtype), steps.size, _ptr(vals, dtype), vals.size,
        optr, olen, nobs, _ptr(out, dtype), out.size)
    _lib.raise_for_status(st)


def _linear_rectilinear(dtype, sfx, grids, vals, obs, out, fam="linear"):
    lib = _lib.load()
    vals = _check_arr("vals", vals, dtype)
    out = _check_arr("out", out, dtype, writable=True)
    gptr, glen, ng, _k1 = _slice_of_slices("grids", grids, dtype)
    optr, olen, nobs, _k2 = _slice_of_slices("obs", obs, dtype)
    st = getattr(lib, f"interpn_hip_{fam}_rectilinear_{sfx}")(
        gptr, glen, ng, _ptr(vals, dtype), vals.size, optr, olen, nobs, _ptr(out, dtype), out.size)
    _lib.raise_for_status(st)


def _cubic_regular(dtype, sfx, dims, starts, steps, vals, linearize_extrapolation, obs, out):
    lib = _lib.load()
    d, nd = _dims(dims)
    starts = _check_arr("starts", starts, dtype)
    steps = _check_arr("steps", steps, dtype)
    vals = _check_arr("vals", vals, dtype)
    out = _check_arr("out", out, dtype, writable=True)
    optr, olen, nobs, _keep = _slice_of_slices("obs", obs, dtype)
    st = getattr(lib, f"interpn_hip_cubic_regular_{sfx}")(
        d, nd, _ptr(starts, dtype), starts.size, _ptr(steps, dtype), steps.size, _ptr(vals, dtype), vals.size,
        int(bool(linearize_extrapolation)), optr, olen, nobs, _ptr(out, dtype), out.size)
    _lib.raise_for_status(st)


def _cubic_rectilinear(dtype, sfx, grids, vals, linearize_extrapolation, obs, out):
    lib = _lib.load()
    vals = _check_arr("vals", vals, dtype)
    out = _check_arr("out", out, dtype, writable=True)
    gptr, glen, ng, _k1 = _slice_of_slices("grids", grids, dtype)
    optr, olen, nobs, _k2 = _slice_of_slices("obs", obs, dtype)
    st = getattr(lib, f"interpn_hip_cubic_rectilinear_{sfx}")(
        gptr, glen, ng, _ptr(vals, dtype), vals.size, int(bool(linearize_extrapolation)), optr, olen, nobs,
        _ptr(out, dtype), out.size)
    _lib.raise_for_status(st)


def _bool_out(out):
    if not isinstance(out, np.ndarray) or out.dtype != np.bool_ or out.ndim != 1 or not out.flags.c_contiguous:
        raise TypeError("argument 'out': expected a contiguous 1-D numpy bool array")
    return out


def _check_bounds_regular(dtype, sfx, dims, starts, steps, obs, atol, out):
    lib = _lib.load()
    d, nd = _dims(dims)
    starts = _check_arr("starts", starts, dtype)
    steps = _check_arr("steps", steps, dtype)
    out = _bool_out(out)
    optr, olen, nobs, _keep = _slice_of_slices("obs", obs, dtype)
    flags = np.zeros(out.size, dtype=np.uint8)
    st = getattr(lib, f"interpn_hip_check_bounds_regular_{sfx}")(
        d, nd, _ptr(starts, dtype), starts.size, _ptr(steps, dtype), steps.size, optr, olen, nobs,
        _ctype(dtype)(float(atol)), flags.ctypes.data_as(POINTER(c_uint8)), flags.size)
    _lib.raise_for_status(st)
    out[:] = flags.astype(bool)


def _check_bounds_rectilinear(dtype, sfx, grids, obs, atol, out):
    lib = _lib.load()
    out = _bool_out(out)
    gptr, glen, ng, _k1 = _slice_of_slices("grids", grids, dtype)
    optr, olen, nobs, _k2 = _slice_of_slices("obs", obs, dtype)
    flags = np.zeros(out.size, dtype=np.uint8)
    st = getattr(lib, f"interpn_hip_check_bounds_rectilinear_{sfx}")(
        gptr, glen, ng, optr, olen, nobs, _ctype(dtype)(float(atol)), flags.ctypes.data_as(POINTER(c_uint8)),
        flags.size)
    _lib.raise_for_status(st)
    out[:] = flags.astype(bool)


# --- the reference's names (src/interpn/raw.pyi:32-147) ---------------------------------------
def interpn_linear_regular_f64(dims, starts, steps, vals, obs, out) -> None:
    _linear_regular(np.float64, "f64", dims, starts, steps, vals, obs, out)


def interpn_linear_regular_f32(dims, starts, steps, vals, obs, out) -> None:
    _linear_regular(np.float32, "f32", dims, starts, steps, vals, obs, out)


def interpn_linear_rectilinear_f64(grids, vals, obs, out) -> None:
    _linear_rectilinear(np.float64, "f64", grids, vals, obs, out)


def interpn_linear_rectilinear_f32(grids, vals, obs, out) -> None:
    _linear_rectilinear(np.float32, "f32", grids, vals, obs, out)


def interpn_nearest_regular_f64(dims, starts, steps, vals, obs, out) -> None:
    _linear_regular(np.float64, "f64", dims, starts, steps, vals, obs, out, fam="nearest")


def interpn_nearest_regular_f32(dims, starts, steps, vals, obs, out) -> None:
    _linear_regular(np.float32, "f32", dims, starts, steps, vals, obs, out, fam="nearest")


def interpn_nearest_rectilinear_f64(grids, vals, obs, out) -> None:
    _linear_rectilinear(np.float64, "f64", grids, vals, obs, out, fam="nearest")


def interpn_nearest_rectilinear_f32(grids, vals, obs, out) -> None:
    _linear_rectilinear(np.float32, "f32", grids, vals, obs, out, fam="nearest")


def interpn_cubic_regular_f64(dims, starts, steps, vals, linearize_extrapolation, obs, out) -> None:
    _cubic_regular(np.float64, "f64", dims, starts, steps, vals, linearize_extrapolation, obs, out)


def interpn_cubic_regular_f32(dims, starts, steps, vals, linearize_extrapolation, obs, out) -> None:
    _cubic_regular(np.float32, "f32", dims, starts, steps, vals, linearize_extrapolation, obs, out)


def interpn_cubic_rectilinear_f64(grids, vals, linearize_extrapolation, obs, out) -> None:
    _cubic_rectilinear(np.float64, "f64", grids, vals, linearize_extrapolation, obs, out)


def interpn_cubic_rectilinear_f32(grids, vals, linearize_extrapolation, obs, out) -> None:
    _cubic_rectilinear(np.float32, "f32", grids, vals, linearize_extrapolation, obs, out)


def check_bounds_regular_f64(dims, starts, steps, obs, atol, out) -> None:
    _check_bounds_regular(np.float64, "f64", dims, starts, steps, obs, atol, out)


def check_bounds_regular_f32(dims, starts, steps, obs, atol, out) -> None:
    _check_bounds_regular(np.float32, "f32", dims, starts, steps, obs, atol, out)


def check_bounds_rectilinear_f64(grids, obs, atol, out) -> None:
    _check_bounds_rectilinear(np.float64, "f64", grids, obs, atol, out)


def check_bounds_rectilinear_f32(grids, obs, atol, out) -> None:
    _check_bounds_rectilinear(np.float32, "f32", grids, obs, atol, out)


__all__ = [
    "interpn_linear_regular_f64",
    "interpn_linear_regular_f32",
    "interpn_linear_rectilinear_f64",
    "interpn_linear_rectilinear_f32",
    "interpn_nearest_regular_f64",
    "interpn_nearest_regular_f32",
    "interpn_nearest_rectilinear_f64",
    "interpn_nearest_rectilinear_f32",
    "interpn_cubic_regular_f64",
    "interpn_cubic_regular_f32",
    "interpn_cubic_rectilinear_f64",
    "interpn_cubic_rectilinear_f32",
    "check_bounds_regular_f64",
    "check_bounds_regular_f32",
    "check_bounds_rectilinear_f64",
    "check_bounds_rectilinear_f32",
]
