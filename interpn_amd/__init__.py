"""interpn_amd — MI355X (gfx950) implementation of the batched per-observation-point hot path of
jlogan03/interpn (multilinear / multicubic interpolation on regular and rectilinear grids).

The public names mirror the reference's Python package (src/interpn/__init__.py): `interpn()`,
`raw`, and the `Multilinear*/Multicubic*` classes.  Everything evaluates through the HIP kernels
behind the C ABI of include/interpn_hip.h; there is no CPU evaluation path in this package.
"""

from __future__ import annotations

from collections.abc import Sequence
from typing import Literal

import numpy as np

from . import _lib, raw
from .classes import (MulticubicRectilinear, MulticubicRegular, MultilinearRectilinear, MultilinearRegular,
                      NearestRectilinear, NearestRegular)
from .handle import Interpolator, eval_device_sharded, eval_host_sharded

__version__ = "0.1.0"

def trim(device: int = -1) -> int:
    """Release the device memory of destroyed interpolators that the library keeps for reuse on
    `device` (`interpn_hip_trim`; -1 = the current device).  Returns the bytes released."""
    import ctypes

    freed = ctypes.c_size_t(0)
    _lib.raise_for_status(_lib.load().interpn_hip_trim(int(device), ctypes.byref(freed)))
    return int(freed.value)


__all__ = [
    "trim",
    "eval_host_sharded",
    "eval_device_sharded",
    "__version__",
    "raw",
    "interpn",
    "Interpolator",
    "MultilinearRegular",
    "MultilinearRectilinear",
    "MulticubicRegular",
    "MulticubicRectilinear",
    "NearestRegular",
    "NearestRectilinear",
]


def interpn(
    obs: Sequence,
    grids: Sequence,
    vals,
    *,
    method: Literal["linear", "cubic", "nearest"] = "linear",
    out=None,
    linearize_extrapolation: bool = True,
    assume_regular: bool = False,
    check_bounds: bool = False,
    bounds_atol: float = 1e-8,
):
    """Evaluate an N-dimensional grid at the supplied observation points.

    Same contract as the reference's helper (src/interpn/__init__.py:48-194): inputs are
    ravelled and made contiguous, the dtype is taken from `vals` (float64 / float32), a grid is
    treated as regular iff every axis has exactly equal spacing (`_check_regular`, :197-203)
    or `assume_regular` is set, and the call dispatches to the matching raw function (:135-192).
    """
    if len(obs) and _is_cuda_tensor(obs[0]):
        return _interpn_on_device(obs, grids, vals, method, out, linearize_extrapolation, assume_regular,
                                  check_bounds, bounds_atol)
    # src/interpn/__init__.py:86-88 (the reference's `out or ...` raises on multi-element arrays;
    # `is None` is what it means)
    out = out if out is not None else np.zeros_like(obs[0])
    outshape = out.shape
    out = out.ravel()

    obs = [np.ascontiguousarray(x.ravel()) for x in obs]
    grids = [np.ascontiguousarray(x.ravel()) for x in grids]
    vals = np.ascontiguousarray(vals.ravel())

    dtype = vals.dtype
    assert dtype in [np.float64, np.float32], "`interpn` defined only for float32 and float64 data"

    is_regular = assume_regular or _check_regular(grids)

    if is_regular:
        dims = [len(grid) for grid in grids]
        starts = np.array([grid[0] for grid in grids], dtype=dtype)
        steps = np.array([grid[1] - grid[0] for grid in grids], dtype=dtype)

    sfx = "f64" if dtype == np.float64 else "f32"

    if check_bounds:
        outb = np.zeros(len(grids), dtype=bool)
        if is_regular:
            getattr(raw, f"check_bounds_regular_{sfx}")(dims, starts, steps, obs, bounds_atol, outb)
        else:
            getattr(raw, f"check_bounds_rectilinear_{sfx}")(grids, obs, bounds_atol, outb)
        if any(outb):
            raise ValueError("Observation points violate interpolator bounds")

    if method == "linear":
        if is_regular:
            getattr(raw, f"interpn_linear_regular_{sfx}")(dims, starts, steps, vals, obs, out)
        else:
            getattr(raw, f"interpn_linear_rectilinear_{sfx}")(grids, vals, obs, out)
    elif method == "nearest":
        if is_regular:
            getattr(raw, f"interpn_nearest_regular_{sfx}")(dims, starts, steps, vals, obs, out)
        else:
            getattr(raw, f"interpn_nearest_rectilinear_{sfx}")(grids, vals, obs, out)
    elif method == "cubic":
        if is_regular:
            getattr(raw, f"interpn_cubic_regular_{sfx}")(dims, starts, steps, vals, linearize_extrapolation, obs, out)
        else:
            getattr(raw, f"interpn_cubic_rectilinear_{sfx}")(grids, vals, linearize_extrapolation, obs, out)
    else:
        raise ValueError(f"Unsupported interpolation configuration: {dtype}, {is_regular}, {method}")

    return out.reshape(outshape)


def _is_cuda_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "is_cuda") and bool(x.is_cuda)


def _interpn_on_device(obs, grids, vals, method, out, linearize_extrapolation, assume_regular, check_bounds,
                       bounds_atol):
    """`interpn()` for observation points that already live on the GPU (torch CUDA tensors): same
    rules as the host form (ravelled inputs, dtype from `vals`, exact-spacing regularity test,
    optional bounds check), the points and the result never cross PCIe.  Returns a tensor."""
    import torch

    if method not in ("linear", "cubic", "nearest"):
        raise ValueError(f"Unsupported interpolation configuration: {method}")
    shape = (out if out is not None else obs[0]).shape
    obs_t = [x.reshape(-1).contiguous() for x in obs]
    grids = [np.ascontiguousarray(np.asarray(x).ravel()) for x in grids]
    vals = vals if _is_cuda_tensor(vals) else np.ascontiguousarray(np.asarray(vals).ravel())
    dtype = np.dtype(np.float64 if str(vals.dtype).endswith("64") else np.float32)
    assert str(vals.dtype).endswith(("float64", "float32")), "`interpn` defined only for float32 and float64 data"
    grids = [g.astype(dtype, copy=False) for g in grids]
    if _is_cuda_tensor(vals):
        vals = vals.reshape(-1).contiguous()
    # The interpolator lives where the points are (not on whatever device happens to be current).
    device = obs_t[0].device.index if obs_t[0].device.index is not None else torch.cuda.current_device()
    if _is_cuda_tensor(vals) and vals.device.index not in (None, device):
        raise ValueError(f"vals is on {vals.device} but the observation points are on cuda:{device}")
    if assume_regular or _check_regular(grids):
        starts = np.array([g[0] for g in grids], dtype=dtype)
        steps = np.array([g[1] - g[0] for g in grids], dtype=dtype)
        it = Interpolator.regular(method, [len(g) for g in grids], starts, steps, vals,
                                  linearize_extrapolation=linearize_extrapolation, device=device, dtype=dtype)
    else:
        it = Interpolator.rectilinear(method, grids, vals, linearize_extrapolation=linearize_extrapolation,
                                      device=device, dtype=dtype)
    try:
        if check_bounds and it.check_bounds_tensors(obs_t, bounds_atol).any():
            raise ValueError("Observation points violate interpolator bounds")
        if out is not None:
            if not out.is_contiguous():
                raise ValueError("out: expected a contiguous CUDA tensor")
            out_t = out.reshape(-1)
        else:
            out_t = torch.empty_like(obs_t[0])
        it.eval_tensors(obs_t, out_t)
        it.finish()
    finally:
        it.close()
    return out_t.reshape(shape)


def _check_regular(grids) -> bool:
    """src/interpn/__init__.py:197-203 — exact equality of all spacings."""
    is_regular = True
    for grid in grids:
        dgrid = np.diff(grid)
        is_regular = is_regular and np.all(dgrid == dgrid[0])
    return bool(is_regular)
