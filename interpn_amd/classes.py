"""Interpolator classes with the surface of the reference's Python wrappers
(src/interpn/multilinear_regular.py:24-212, multilinear_rectilinear.py, multicubic_regular.py,
multicubic_rectilinear.py): `.new(...)`, `.eval(obs, out=None)`, `.eval_unchecked`,
`.check_bounds(obs, atol)`, `.ndims()` (+ `.dims()` on rectilinear grids), JSON round trip via
`.model_dump_json()` / `.model_validate_json()`.

Differences from the reference, all behind the same interface: the grid is uploaded to HBM once
(first evaluation) and stays resident, where the reference rebuilds its Rust struct on every
call (src/multilinear/regular.rs:65-71); `.eval` also accepts torch CUDA tensors, in which case
the observation points never leave the device.
"""

from __future__ import annotations

import json
import threading
from functools import reduce

import numpy as np

from . import raw
from .handle import Interpolator

_HANDLE_LOCK = threading.Lock()


def _as_flat(a, dtype):
    """ArrayF64/ArrayF32 validator: contiguous copy in the target dtype (serialization.py:28-39)."""
    if isinstance(a, str):
        return np.ascontiguousarray(np.array(json.loads(a), dtype=dtype))
    return np.ascontiguousarray(np.asarray(a).astype(dtype))


def _is_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class _Base:
    _method = "linear"
    _kind = "regular"

    def __setattr__(self, key, value):
        if getattr(self, "_frozen", False):
            raise TypeError(f"{type(self).__name__} is immutable")  # ConfigDict(frozen=True)
        object.__setattr__(self, key, value)

    def _freeze(self):
        object.__setattr__(self, "_handle", None)
        object.__setattr__(self, "_frozen", True)

    # -- device residency -------------------------------------------------------------------
    def _interp(self) -> Interpolator:
        if self._handle is None:
            with _HANDLE_LOCK:  # first eval from several threads: upload the grid once
                if self._handle is None:
                    object.__setattr__(self, "_handle", self._make_handle())
        return self._handle

    def to_device(self, device: int = -1) -> "_Base":
        """Upload the grid now (otherwise done lazily by the first eval)."""
        object.__setattr__(self, "_handle", self._make_handle(device))
        return self

    @property
    def dtype(self):
        return self.vals.dtype

    # -- reference surface ------------------------------------------------------------------
    def eval(self, obs, out=None):
        """Evaluate at observation points; allocates the output if not given
        (multilinear_regular.py:101-123)."""
        if obs and _is_tensor(obs[0]):
            res = self._interp().eval_tensors(obs, out)
            self._interp().finish()
            return res
        out_inner = out if out is not None else np.zeros_like(obs[0])
        self.eval_unchecked(obs, out_inner)
        return out_inner

    def eval_unchecked(self, obs, out=None):
        dtype = self.vals.dtype
        if dtype not in (np.float64, np.float32):
            raise TypeError(f"Unexpected data type: {dtype}")
        out_inner = out if out is not None else np.zeros_like(obs[0])
        self._interp().eval_host(list(obs), out_inner)
        return out_inner

    def model_dump_json(self) -> str:
        def arr(a):
            return {"data": json.dumps(a.tolist()), "dtype": a.dtype.name}

        d = {}
        for k in self._fields:
            v = getattr(self, k)
            if isinstance(v, np.ndarray):
                d[k] = arr(v)
            elif isinstance(v, list) and v and isinstance(v[0], np.ndarray):
                d[k] = [arr(x) for x in v]
            else:
                d[k] = v
        return json.dumps(d)

    @classmethod
    def model_validate_json(cls, s: str):
        d = json.loads(s)

        def unarr(a):
            return _as_flat(a["data"], np.float64 if a["dtype"] == "float64" else np.float32)

        kw = {}
        for k, v in d.items():
            if isinstance(v, dict) and "data" in v:
                kw[k] = unarr(v)
            elif isinstance(v, list) and v and isinstance(v[0], dict):
                kw[k] = [unarr(x) for x in v]
            else:
                kw[k] = v
        return cls(**kw)


class _RegularBase(_Base):
    _kind = "regular"
    _fields = ("dims", "starts", "steps", "vals")

    def __init__(self, dims, starts, steps, vals, **kw):
        dtype = np.asarray(vals).dtype if not isinstance(vals, str) else np.float64
        self.dims = [int(x) for x in dims]
        self.starts = _as_flat(starts, starts.dtype if isinstance(starts, np.ndarray) else dtype)
        self.steps = _as_flat(steps, steps.dtype if isinstance(steps, np.ndarray) else dtype)
        self.vals = _as_flat(vals, dtype)
        for k, v in kw.items():
            setattr(self, k, v)
        self._validate()
        self._freeze()

    def _validate(self):
        # multilinear_regular.py:73-96 / multicubic_regular.py:93-120
        ndims = self.ndims()
        assert ndims <= 8 and ndims >= 1, "Number of dimensions must be at least 1 and no more than 8"
        assert self.starts.size == ndims, "Grid dimension mismatch"
        assert self.steps.size == ndims, "Grid dimension mismatch"
        assert self.vals.size == reduce(lambda acc, x: acc * x, self.dims), (
            "Size of value array does not match grid dims"
        )
        assert all([x > 0.0 for x in self.steps]), "All grid steps must be positive and nonzero"
        assert all([x.dtype == self.vals.dtype for x in [self.steps, self.vals]]), (
            "All grid inputs must be of the same data type (np.float32 or np.float64)"
        )

    def ndims(self) -> int:
        return len(self.dims)

    def _make_handle(self, device: int = -1) -> Interpolator:
        dtype = self.vals.dtype
        return Interpolator.regular(self._method, self.dims, self.starts.astype(dtype), self.steps, self.vals,
                                    getattr(self, "linearize_extrapolation", False), device, dtype)

    def check_bounds(self, obs, atol: float):
        if obs and _is_tensor(obs[0]):
            return self._interp().check_bounds_tensors(obs, atol)
        ndims = self.ndims()
        out = np.array([False] * ndims)
        dtype = self.vals.dtype
        fn = raw.check_bounds_regular_f64 if dtype == np.float64 else raw.check_bounds_regular_f32
        if dtype not in (np.float64, np.float32):
            raise TypeError(f"Unexpected data type: {dtype}")
        fn(self.dims, self.starts, self.steps, [np.asarray(x).flatten() for x in obs], atol, out)
        return out


class _RectilinearBase(_Base):
    _kind = "rectilinear"
    _fields = ("grids", "vals")

    def __init__(self, grids, vals, **kw):
        dtype = np.asarray(vals).dtype if not isinstance(vals, str) else np.float64
        self.grids = [_as_flat(g, g.dtype if isinstance(g, np.ndarray) else dtype) for g in grids]
        self.vals = _as_flat(vals, dtype)
        for k, v in kw.items():
            setattr(self, k, v)
        self._validate()
        self._freeze()

    def _validate(self):
        # multilinear_rectilinear.py:67-89
        dims = self.dims()
        ndims = self.ndims()
        assert ndims <= 8 and ndims >= 1, "Number of dimensions must be at least 1 and no more than 8"
        assert self.vals.size == reduce(lambda acc, x: acc * x, dims), "Size of value array does not match grid dims"
        assert all([np.all(np.diff(x) > 0.0) for x in self.grids]), "All grids must be monotonically increasing"
        assert all([x.dtype == self.vals.dtype for x in self.grids]), (
            "All grid inputs must be of the same data type (np.float32 or np.float64)"
        )

    def ndims(self) -> int:
        return len(self.grids)

    def dims(self) -> list:
        return [x.size for x in self.grids]

    def _make_handle(self, device: int = -1) -> Interpolator:
        return Interpolator.rectilinear(self._method, self.grids, self.vals,
                                        getattr(self, "linearize_extrapolation", False), device, self.vals.dtype)

    def check_bounds(self, obs, atol: float):
        if obs and _is_tensor(obs[0]):
            return self._interp().check_bounds_tensors(obs, atol)
        ndims = self.ndims()
        out = np.array([False] * ndims)
        dtype = self.vals.dtype
        if dtype not in (np.float64, np.float32):
            raise TypeError(f"Unexpected data type: {dtype}")
        fn = raw.check_bounds_rectilinear_f64 if dtype == np.float64 else raw.check_bounds_rectilinear_f32
        fn(self.grids, [np.asarray(x).flatten() for x in obs], atol, out)
        return out


class MultilinearRegular(_RegularBase):
    """Multilinear interpolation on a regular grid in up to 8 dimensions
    (src/interpn/multilinear_regular.py:24)."""

    _method = "linear"

    @classmethod
    def new(cls, dims, starts, steps, vals) -> "MultilinearRegular":
        dtype = vals.dtype
        return cls(dims=dims, starts=_as_flat(starts.flatten(), dtype), steps=_as_flat(steps.flatten(), dtype),
                   vals=_as_flat(vals.flatten(), dtype))


class MultilinearRectilinear(_RectilinearBase):
    """Multilinear interpolation on a rectilinear grid (src/interpn/multilinear_rectilinear.py:24)."""

    _method = "linear"

    @classmethod
    def new(cls, grids, vals) -> "MultilinearRectilinear":
        dtype = vals.dtype
        return cls(grids=[_as_flat(x, dtype) for x in grids], vals=_as_flat(vals.flatten(), dtype))


class NearestRegular(_RegularBase):
    """Nearest-neighbour interpolation on a regular grid in up to 6 dimensions
    (src/interpn/nearest_regular.py)."""

    _method = "nearest"

    def _validate(self):
        assert self.ndims() <= 6 and self.ndims() >= 1, "Number of dimensions must be at least 1 and no more than 6"
        super()._validate()

    @classmethod
    def new(cls, dims, starts, steps, vals) -> "NearestRegular":
        dtype = vals.dtype
        return cls(dims=dims, starts=_as_flat(starts.flatten(), dtype), steps=_as_flat(steps.flatten(), dtype),
                   vals=_as_flat(vals.flatten(), dtype))


class NearestRectilinear(_RectilinearBase):
    """Nearest-neighbour interpolation on a rectilinear grid in up to 6 dimensions
    (src/interpn/nearest_rectilinear.py)."""

    _method = "nearest"

    def _validate(self):
        assert self.ndims() <= 6 and self.ndims() >= 1, "Number of dimensions must be at least 1 and no more than 6"
        super()._validate()

    @classmethod
    def new(cls, grids, vals) -> "NearestRectilinear":
        dtype = vals.dtype
        return cls(grids=[_as_flat(x, dtype) for x in grids], vals=_as_flat(vals.flatten(), dtype))


class MulticubicRegular(_RegularBase):
    """Cubic Hermite interpolation on a regular grid (src/interpn/multicubic_regular.py:24).
    `linearize_extrapolation` defaults to True as in the reference (:59)."""

    _method = "cubic"
    _fields = ("dims", "starts", "steps", "vals", "linearize_extrapolation")

    def __init__(self, dims, starts, steps, vals, linearize_extrapolation=True):
        super().__init__(dims, starts, steps, vals, linearize_extrapolation=bool(linearize_extrapolation))

    @classmethod
    def new(cls, dims, starts, steps, vals, linearize_extrapolation: bool = True) -> "MulticubicRegular":
        dtype = vals.dtype
        return cls(dims=dims, starts=_as_flat(starts.flatten(), dtype), steps=_as_flat(steps.flatten(), dtype),
                   vals=_as_flat(vals.flatten(), dtype), linearize_extrapolation=linearize_extrapolation)


class MulticubicRectilinear(_RectilinearBase):
    """Cubic Hermite interpolation on a rectilinear grid (src/interpn/multicubic_rectilinear.py:24)."""

    _method = "cubic"
    _fields = ("grids", "vals", "linearize_extrapolation")

    def __init__(self, grids, vals, linearize_extrapolation=True):
        super().__init__(grids, vals, linearize_extrapolation=bool(linearize_extrapolation))

    @classmethod
    def new(cls, grids, vals, linearize_extrapolation: bool = True) -> "MulticubicRectilinear":
        dtype = vals.dtype
        return cls(grids=[_as_flat(x, dtype) for x in grids], vals=_as_flat(vals.flatten(), dtype),
                   linearize_extrapolation=linearize_extrapolation)
