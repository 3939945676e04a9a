"""Persistent, device-resident interpolators over the handle API of include/interpn_hip.h.

`Interpolator` is the counterpart of the reference's interpolator structs
(`MultilinearRegular::new(..).interp(obs, out)`, src/multilinear/regular.rs:225-283 and the
three siblings) with the grid kept in HBM between evaluations.  Observation points may be host
numpy arrays (`eval_host`) or device buffers (`eval_device`, e.g. torch tensors on the GPU —
torch is only plumbing for device memory and streams here).
"""

from __future__ import annotations

import ctypes
from ctypes import POINTER, c_double, c_float, c_size_t, c_uint64, c_void_p

import numpy as np

from . import _lib
from .raw import _check_arr, _dims, _slice_of_slices


class Interpolator:
    def __init__(self, handle: int, dtype, ndims: int, keepalive=None):
        self._h = c_void_p(handle)
        self.dtype = np.dtype(dtype)
        self._ndims = ndims
        self._keepalive = keepalive  # e.g. a torch tensor whose storage the handle borrows
        # Streams that evaluations were enqueued on since the last finish(): raw handle -> the
        # object that owns it (a torch Stream kept alive here; None for a caller-supplied integer).
        self._pending_streams = {}
        self.last_path = None        # "in_place" | "binned" | "sweep": what the most recent device evaluation did
        self.last_path_reason = ""   # why a handle that can sort its points did not

    # -- construction ---------------------------------------------------------------------
    @staticmethod
    def _vals_arg(vals, dtype):
        """Return (void* address, nvals, mem kind, keepalive) for a numpy array or a torch tensor."""
        if isinstance(vals, np.ndarray):
            v = _check_arr("vals", vals, dtype)
            return v.ctypes.data_as(c_void_p), v.size, _lib.MEM_HOST, v
        # torch tensor (duck-typed so that torch stays an optional import)
        if hasattr(vals, "data_ptr") and hasattr(vals, "is_cuda"):
            if not vals.is_contiguous() or vals.dim() != 1:
                raise ValueError("argument 'vals': expected a contiguous 1-D tensor")
            want = "torch.float64" if dtype == np.float64 else "torch.float32"
            if str(vals.dtype) != want:
                raise TypeError(f"argument 'vals': expected {want}, got {vals.dtype}")
            mem = _lib.MEM_DEVICE if vals.is_cuda else _lib.MEM_HOST
            return c_void_p(vals.data_ptr()), vals.numel(), mem, vals
        raise TypeError("argument 'vals': expected a numpy array or a torch tensor")

    @staticmethod
    def _method_arg(method: str, fma) -> int:
        """`method` of interpn_hip_create_*: the method plus the per-interpolator flavour of the
        reference's `fma` cargo feature (None = the process default, on)."""
        m = _lib.METHODS[method]
        if fma is None:
            return m
        return m | (_lib.FLAVOUR_FMA if fma else _lib.FLAVOUR_NO_FMA)

    @classmethod
    def regular(cls, method: str, dims, starts, steps, vals, linearize_extrapolation: bool = False,
                device: int = -1, dtype=None, fma=None) -> "Interpolator":
        dtype = np.dtype(dtype or starts.dtype)
        sfx = "f64" if dtype == np.float64 else "f32"
        ct = c_double if dtype == np.float64 else c_float
        lib = _lib.load()
        d, nd = _dims(dims)
        starts = _check_arr("starts", starts, dtype)
        steps = _check_arr("steps", steps, dtype)
        vptr, nvals, mem, keep = cls._vals_arg(vals, dtype)
        h = c_void_p()
        st = getattr(lib, f"interpn_hip_create_regular_{sfx}")(
            cls._method_arg(method, fma), d, nd, starts.ctypes.data_as(POINTER(ct)),
            starts.size, steps.ctypes.data_as(POINTER(ct)), steps.size, vptr, nvals, mem,
            int(bool(linearize_extrapolation)), int(device), ctypes.byref(h))
        _lib.raise_for_status(st)
        return cls(h.value, dtype, nd, keep if mem == _lib.MEM_DEVICE else None)

    @classmethod
    def rectilinear(cls, method: str, grids, vals, linearize_extrapolation: bool = False, device: int = -1,
                    dtype=None, fma=None) -> "Interpolator":
        dtype = np.dtype(dtype or grids[0].dtype)
        sfx = "f64" if dtype == np.float64 else "f32"
        lib = _lib.load()
        gptr, glen, ng, _keep_grids = _slice_of_slices("grids", grids, dtype)
        vptr, nvals, mem, keep = cls._vals_arg(vals, dtype)
        h = c_void_p()
        st = getattr(lib, f"interpn_hip_create_rectilinear_{sfx}")(
            cls._method_arg(method, fma), gptr, glen, ng, vptr, nvals, mem,
            int(bool(linearize_extrapolation)), int(device), ctypes.byref(h))
        _lib.raise_for_status(st)
        return cls(h.value, dtype, ng, keep if mem == _lib.MEM_DEVICE else None)

    def replicate(self, device: int = -1) -> "Interpolator":
        """Clone onto another GPU of this process, grid copied device to device
        (`interpn_hip_replicate`); the clone owns its copy."""
        h = c_void_p()
        _lib.raise_for_status(_lib.load().interpn_hip_replicate(self._h, int(device), ctypes.byref(h)))
        return Interpolator(h.value, self.dtype, self._ndims)

    # -- evaluation -----------------------------------------------------------------------
    def ndims(self) -> int:
        return self._ndims

    def device(self) -> int:
        return _lib.load().interpn_hip_device(self._h)

    def set_blocks_per_cu(self, n: int) -> None:
        _lib.raise_for_status(_lib.load().interpn_hip_set_blocks_per_cu(self._h, int(n)))

    def set_option(self, name: str, value: int) -> None:
        """Per-handle tuning / testing option (`interpn_hip_set_option`; names in include/interpn_hip.h)."""
        _lib.raise_for_status(_lib.load().interpn_hip_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = ctypes.c_longlong(0)
        _lib.raise_for_status(_lib.load().interpn_hip_get_option(self._h, name.encode(), ctypes.byref(v)))
        return int(v.value)

    def kernel_name(self) -> str:
        """Instantiation the most recent evaluation launched, in rocprofv3's spelling ("" before any)."""
        buf = ctypes.create_string_buffer(256)
        _lib.raise_for_status(_lib.load().interpn_hip_kernel_name(self._h, buf, len(buf)))
        return buf.value.decode()

    def table_layout(self):
        """(bytes, step_i, step_j) of the re-laid grid copy the kernels read; (0, 0, 0) = C order.
        After an evaluation that took the 3-D multilinear sweep kernel (`k_linear_sweep<`): that kernel's table (a handle
        may keep two brick tables, e.g. 64^3 f64: (1,2) for the brick kernel, (1,1) for the sweep).  The steps alone do
        not tell the f64 2 x 2 x KW bricks from the f32 2 x 4 x 4 ones (both report (1, 1)): `sweep_brick_form()` does.
        The 2-D, nearest-neighbour and multicubic sweep kernels read the handle's one table (brick / C order / tile):
        for them this returns that table, as for the one-pass kernels."""
        if self.kernel_name().startswith("interpn::k_linear_sweep<"):
            lay = self.get_option("sweep_layout")
            return self.get_option("sweep_table_bytes"), lay // 10, lay % 10
        si, sj = ctypes.c_int(0), ctypes.c_int(0)
        nbytes = _lib.load().interpn_hip_table_bytes(self._h, ctypes.byref(si), ctypes.byref(sj))
        return int(nbytes), int(si.value), int(sj.value)

    def sweep_brick_form(self) -> str:
        """Brick shape of the 3-D multilinear sweep kernel's table: "2x2xKW" (f64; f32 with KW = 8), "2x4x4" (f32), or ""
        when the handle has no such table."""
        if not self.get_option("sweep_table_bytes"):
            return ""
        return "2x4x4" if self.get_option("sweep_cell") == 2 else "2x2xKW"

    def pci_address(self):
        """(domain, bus, device) of the GPU this interpolator lives on, or None if the runtime does not say."""
        v = self.get_option("dev_pci")
        return None if v < 0 else (v >> 16, (v >> 8) & 0xFF, v & 0xFF)

    def _check_same_device(self, what: str, tensor) -> None:
        """Kernels run under the handle's device with the grid resident there: a tensor on another
        GPU would be read across devices on a stream of the wrong device."""
        idx = tensor.device.index
        if idx is None:
            import torch

            idx = torch.cuda.current_device()
        if idx != self.device():
            raise ValueError(f"{what} is on cuda:{idx} but this interpolator lives on cuda:{self.device()}")

    def eval_host(self, obs, out: np.ndarray) -> np.ndarray:
        """`.interp(obs, out)` on host arrays (synchronous)."""
        lib = _lib.load()
        out = _check_arr("out", out, self.dtype, writable=True)
        optr, olen, nobs, _keep = _slice_of_slices("obs", obs, self.dtype)
        vp = (c_void_p * max(nobs, 1))()
        for i in range(nobs):
            vp[i] = ctypes.cast(optr[i], c_void_p)
        st = lib.interpn_hip_eval_host(self._h, vp, olen, nobs, out.ctypes.data_as(c_void_p), out.size)
        _lib.raise_for_status(st)
        return out

    def reserve(self, npoints: int, nstreams: int = 1) -> None:
        """Pre-allocate what device evaluations of up to `npoints` points on up to `nstreams`
        concurrent streams need (`interpn_hip_reserve`); afterwards such evaluations allocate
        nothing, also with `no_alloc=True`."""
        _lib.raise_for_status(_lib.load().interpn_hip_reserve(self._h, int(npoints), int(nstreams)))

    def stage_ms(self):
        """Durations (ms) of the most recent sorted evaluation's stages — histogram, scan, scatter,
        evaluation kernel — recorded when option "stage_timing" is 1 (`interpn_hip_stage_ms`)."""
        ms = (c_double * 4)()
        _lib.raise_for_status(_lib.load().interpn_hip_stage_ms(self._h, ms, 4))
        return {"hist": ms[0], "scan": ms[1], "scatter": ms[2], "kernel": ms[3]}

    def eval_device_ptrs(self, obs_ptrs, out_ptr: int, npoints: int, stream: int = 0, no_alloc: bool = False) -> str:
        """Enqueue one evaluation on device buffers given as raw addresses (asynchronous).
        Returns the path taken, "in_place", "binned" or "sweep" (also kept in `.last_path`, with
        `.last_path_reason` saying why a handle that can sort its points evaluated in place)."""
        lib = _lib.load()
        n = len(obs_ptrs)
        vp = (c_void_p * max(n, 1))()
        for i, p in enumerate(obs_ptrs):
            vp[i] = c_void_p(int(p))
        path, why = ctypes.c_int(0), ctypes.c_int(0)
        st = lib.interpn_hip_eval_device_ex(self._h, vp, n, c_void_p(int(out_ptr)), int(npoints), c_void_p(int(stream)),
                                            _lib.EVAL_NO_ALLOC if no_alloc else 0, ctypes.byref(path), ctypes.byref(why))
        _lib.raise_for_status(st)
        self.last_path = {_lib.PATH_BINNED: "binned", _lib.PATH_SWEEP: "sweep"}.get(path.value, "in_place")
        self.last_path_reason = _lib.WHY.get(why.value, str(why.value))
        self._pending_streams.setdefault(int(stream), None)
        return self.last_path

    def eval_tensors(self, obs, out=None, stream=None, no_alloc: bool = False):
        """Evaluate on torch CUDA tensors (asynchronous on torch's current stream unless given: a
        torch Stream or a raw hipStream_t integer).  Call `finish()` to synchronise and surface
        "Unrepresentable coordinate value"."""
        import torch

        want = torch.float64 if self.dtype == np.float64 else torch.float32
        obs = list(obs)
        for i, t in enumerate(obs):
            if not (t.is_cuda and t.is_contiguous() and t.dim() == 1 and t.dtype == want):
                raise TypeError(f"obs[{i}]: expected a contiguous 1-D {want} CUDA tensor")
            self._check_same_device(f"obs[{i}]", t)
        n = obs[0].numel() if obs else 0
        for t in obs:
            if t.numel() != n:
                raise AssertionError("Dimension mismatch")
        if out is None:
            out = torch.empty(n, dtype=want, device=torch.device("cuda", self.device()))
        elif not (out.is_cuda and out.is_contiguous() and out.dim() == 1 and out.dtype == want):
            raise TypeError(f"out: expected a contiguous 1-D {want} CUDA tensor")
        elif out.numel() != n:
            raise AssertionError("Dimension mismatch")
        else:
            self._check_same_device("out", out)
        owner = torch.cuda.current_stream(self.device()) if stream is None else stream
        raw = owner.cuda_stream if hasattr(owner, "cuda_stream") else int(owner)
        self.eval_device_ptrs([t.data_ptr() for t in obs], out.data_ptr(), n, raw, no_alloc)
        # keep the Stream OBJECT: a raw handle whose torch Stream was collected would dangle
        self._pending_streams[raw] = owner if hasattr(owner, "cuda_stream") else None
        return out

    def check_bounds_tensors(self, obs, atol: float, stream=None) -> np.ndarray:
        """`check_bounds` on torch CUDA tensors against this interpolator's grid: one flag per
        dimension, True where any coordinate lies outside the grid by `atol` or more
        (src/multilinear/regular.rs:145-182).  The points stay on the device."""
        import torch

        want = torch.float64 if self.dtype == np.float64 else torch.float32
        obs = list(obs)
        for i, t in enumerate(obs):
            if not (t.is_cuda and t.is_contiguous() and t.dim() == 1 and t.dtype == want):
                raise TypeError(f"obs[{i}]: expected a contiguous 1-D {want} CUDA tensor")
            self._check_same_device(f"obs[{i}]", t)
        n = obs[0].numel() if obs else 0
        for t in obs:
            if t.numel() != n:
                raise AssertionError(_lib.strerror(_lib.ERR_DIM_MISMATCH))
        if stream is None:
            stream = torch.cuda.current_stream(self.device()).cuda_stream
        vp = (c_void_p * max(len(obs), 1))()
        for i, t in enumerate(obs):
            vp[i] = c_void_p(t.data_ptr())
        flags = (ctypes.c_uint8 * max(len(obs), 1))()
        st = _lib.load().interpn_hip_check_bounds_device(self._h, vp, len(obs), n, float(atol), flags, len(obs),
                                                         c_void_p(int(stream)))
        _lib.raise_for_status(st)
        return np.array([bool(flags[i]) for i in range(len(obs))])

    def finish(self, stream=None) -> None:
        """Wait for the evaluations enqueued since the last finish and raise
        AssertionError("Unrepresentable coordinate value") if any of them hit a NaN/inf/out-of-range
        coordinate.  Without `stream`, EVERY stream used since the last finish is waited for (the
        status word is sticky per handle, not per stream); with `stream` (a torch Stream or a raw
        integer) only that one."""
        lib = _lib.load()
        if stream is not None:
            raws = [stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)]
        else:
            raws = list(self._pending_streams)
        if not raws:
            try:
                import torch

                raws = [torch.cuda.current_stream(self.device()).cuda_stream if torch.cuda.is_available() else 0]
            except ImportError:
                raws = [0]
        first_bad = None
        status = _lib.OK
        for raw in raws:
            bad = c_uint64(0)
            st = lib.interpn_hip_finish(self._h, c_void_p(int(raw)), ctypes.byref(bad))
            self._pending_streams.pop(raw, None)
            if st == _lib.ERR_UNREPRESENTABLE:
                first_bad = bad.value if first_bad is None else min(first_bad, bad.value)
            elif st != _lib.OK and status == _lib.OK:
                status = st
        _lib.raise_for_status(status)
        if first_bad is not None:
            err = AssertionError(_lib.strerror(_lib.ERR_UNREPRESENTABLE))
            err.first_bad_index = first_bad
            raise err

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.load().interpn_hip_destroy(self._h)
            self._h = c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def eval_host_sharded(interps, obs, out: np.ndarray) -> np.ndarray:
    """Single-process multi-GPU `.interp(obs, out)`: contiguous ranges of the observation index,
    one per interpolator in `interps` (create one per device from the same grid), evaluated
    concurrently (`interpn_hip_eval_host_sharded`).  On "Unrepresentable coordinate value" the
    AssertionError carries `first_bad_index` (global)."""
    interps = list(interps)
    if not interps:
        raise ValueError("eval_host_sharded needs at least one interpolator")
    lib = _lib.load()
    dtype = interps[0].dtype
    out = _check_arr("out", out, dtype, writable=True)
    optr, olen, nobs, _keep = _slice_of_slices("obs", obs, dtype)
    vp = (c_void_p * max(nobs, 1))()
    for i in range(nobs):
        vp[i] = ctypes.cast(optr[i], c_void_p)
    hs = (c_void_p * len(interps))()
    for i, it in enumerate(interps):
        hs[i] = it._h
    bad = c_uint64(0)
    st = lib.interpn_hip_eval_host_sharded(hs, len(interps), vp, olen, nobs, out.ctypes.data_as(c_void_p), out.size,
                                           ctypes.byref(bad))
    if st == _lib.ERR_UNREPRESENTABLE:
        err = AssertionError(_lib.strerror(st))
        err.first_bad_index = bad.value
        raise err
    _lib.raise_for_status(st)
    return out


def eval_device_sharded(interps, obs_shards, out_shards=None):
    """Single-process multi-GPU evaluation of device-resident shards (`interpn_hip_eval_device_sharded`):
    `obs_shards[r]` is the list of N coordinate tensors of shard r on the device of `interps[r]`
    (clones made with `.replicate(device)`), `out_shards[r]` its result tensor (allocated when
    omitted).  Every shard is enqueued on its device's current torch stream before the first is
    waited for; returns the list of result tensors.  On "Unrepresentable coordinate value" the
    AssertionError carries `first_bad_index`, counting the shards' points in shard order."""
    import torch

    interps = list(interps)
    if not interps or len(obs_shards) != len(interps):
        raise ValueError("eval_device_sharded needs one list of coordinate tensors per interpolator")
    lib = _lib.load()
    n = len(interps)
    nd = interps[0].ndims()
    tdt = torch.float64 if interps[0].dtype == np.float64 else torch.float32
    outs = list(out_shards) if out_shards is not None else [None] * n
    hs = (c_void_p * n)()
    obs_pp = (ctypes.POINTER(c_void_p) * n)()
    out_p = (c_void_p * n)()
    npts = (c_size_t * n)()
    streams = (c_void_p * n)()
    keep = []
    for r, it in enumerate(interps):
        shard = list(obs_shards[r])
        if len(shard) != nd:
            raise AssertionError("Dimension mismatch")
        count = int(shard[0].numel())
        for t in shard:
            if t.dtype != tdt or not t.is_contiguous() or int(t.numel()) != count:
                raise ValueError("coordinate tensors of a shard must be contiguous, of the interpolator's dtype and equally long")
            it._check_same_device("obs", t)
        if outs[r] is None:
            outs[r] = torch.empty(count, dtype=tdt, device=shard[0].device)
        it._check_same_device("out", outs[r])
        arr = (c_void_p * max(nd, 1))(*[t.data_ptr() for t in shard])
        keep.append(arr)
        hs[r] = it._h
        obs_pp[r] = ctypes.cast(arr, ctypes.POINTER(c_void_p))
        out_p[r] = outs[r].data_ptr()
        npts[r] = count
        streams[r] = torch.cuda.current_stream(shard[0].device).cuda_stream
    bad = c_uint64(0)
    st = lib.interpn_hip_eval_device_sharded(hs, n, obs_pp, nd, out_p, npts, streams, ctypes.byref(bad))
    if st == _lib.ERR_UNREPRESENTABLE:
        err = AssertionError(_lib.strerror(st))
        err.first_bad_index = bad.value
        raise err
    _lib.raise_for_status(st)
    return outs
