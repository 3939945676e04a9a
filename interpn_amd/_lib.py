"""Loader for libinterpn_hip.so (the C ABI declared in include/interpn_hip.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded this module
raises, and every entry point of the package fails loudly.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_uint8, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# INTERPN_AMD_LIB: another build of the same library (A/B measurements of compiler flags, tools/)
LIB_PATH = os.environ.get("INTERPN_AMD_LIB") or os.path.join(_HERE, "libinterpn_hip.so")

# interpn_hip_status (include/interpn_hip.h)
OK = 0
ERR_DIM_MISMATCH = 1
ERR_UNREPRESENTABLE = 7
ERR_REFERENCE_PANIC = 9
ERR_TOO_MANY_DIMS_6 = 10
ERR_INVALID_ARGUMENT = 32

LINEAR, CUBIC, NEAREST = 0, 1, 2
METHODS = {"linear": LINEAR, "cubic": CUBIC, "nearest": NEAREST}
MEM_HOST, MEM_DEVICE = 0, 1
FLAVOUR_FMA, FLAVOUR_NO_FMA = 0x100, 0x200  # OR-ed into `method` of interpn_hip_create_*
PATH_IN_PLACE, PATH_BINNED, PATH_SWEEP = 0, 1, 2
EVAL_NO_ALLOC = 1
WHY = {0: "", 1: "batch below the break-even size or option binned = 0", 2: "stream under graph capture",
       3: "no reserved scratch block free and allocation not allowed", 4: "scratch allocation failed",
       5: "out or a coordinate array is not 16-byte aligned (sweep evaluation)"}

_lib = None


class InterpnHipError(RuntimeError):
    """Failure of the HIP implementation itself (no device, out of memory, HIP runtime error)."""


class ReferencePanic(RuntimeError):
    """The reference implementation panics on this input (pyo3 surfaces it as PanicException)."""


def _preload_hip_runtime() -> None:
    """One HIP/HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so and
    libhsa-runtime64.so (same sonames as /opt/rocm's).  If libinterpn_hip.so were loaded first it
    would bind to /opt/rocm's copies, and a later `import torch` would bring up a second runtime:
    observed as "no ROCm-capable device is detected" or as a hang inside `import torch`.  Loading
    only torch's libamdhip64.so by path is not enough (its HSA runtime then comes from /opt/rocm
    and torch later loads its own).  So when torch is installed it is imported first — the order
    every working configuration has — and libinterpn_hip.so binds to the runtime torch loaded.
    INTERPN_AMD_NO_TORCH_PRELOAD=1 skips this (processes that never import torch)."""
    import importlib.util
    import sys

    if "torch" in sys.modules or os.environ.get("INTERPN_AMD_NO_TORCH_PRELOAD"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None:
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass  # fall back to the system runtime


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    _preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C interpn_amd/csrc`. interpn_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.interpn_hip_strerror.restype = c_char_p
    lib.interpn_hip_strerror.argtypes = [c_int]
    lib.interpn_hip_last_hip_error.restype = c_char_p
    lib.interpn_hip_version.restype = c_char_p
    lib.interpn_hip_set_fma.argtypes = [c_int]
    lib.interpn_hip_set_fma.restype = c_int
    lib.interpn_hip_device_count.restype = c_int
    lib.interpn_hip_trim.argtypes = [c_int, POINTER(c_size_t)]
    lib.interpn_hip_trim.restype = c_int
    for sfx, ct in (("f64", c_double), ("f32", c_float)):
        pp = POINTER(POINTER(ct))
        p = POINTER(ct)
        sz = POINTER(c_size_t)
        getattr(lib, f"interpn_hip_linear_regular_{sfx}").argtypes = [
            sz, c_size_t, p, c_size_t, p, c_size_t, p, c_size_t, pp, sz, c_size_t, p, c_size_t]
        getattr(lib, f"interpn_hip_linear_rectilinear_{sfx}").argtypes = [
            pp, sz, c_size_t, p, c_size_t, pp, sz, c_size_t, p, c_size_t]
        getattr(lib, f"interpn_hip_nearest_regular_{sfx}").argtypes = [
            sz, c_size_t, p, c_size_t, p, c_size_t, p, c_size_t, pp, sz, c_size_t, p, c_size_t]
        getattr(lib, f"interpn_hip_nearest_rectilinear_{sfx}").argtypes = [
            pp, sz, c_size_t, p, c_size_t, pp, sz, c_size_t, p, c_size_t]
        getattr(lib, f"interpn_hip_cubic_regular_{sfx}").argtypes = [
            sz, c_size_t, p, c_size_t, p, c_size_t, p, c_size_t, c_int, pp, sz, c_size_t, p, c_size_t]
        getattr(lib, f"interpn_hip_cubic_rectilinear_{sfx}").argtypes = [
            pp, sz, c_size_t, p, c_size_t, c_int, pp, sz, c_size_t, p, c_size_t]
        getattr(lib, f"interpn_hip_create_regular_{sfx}").argtypes = [
            c_int, sz, c_size_t, p, c_size_t, p, c_size_t, c_void_p, c_size_t, c_int, c_int, c_int,
            POINTER(c_void_p)]
        getattr(lib, f"interpn_hip_create_rectilinear_{sfx}").argtypes = [
            c_int, pp, sz, c_size_t, c_void_p, c_size_t, c_int, c_int, c_int, POINTER(c_void_p)]
        getattr(lib, f"interpn_hip_check_bounds_regular_{sfx}").argtypes = [
            sz, c_size_t, p, c_size_t, p, c_size_t, pp, sz, c_size_t, ct, POINTER(c_uint8), c_size_t]
        getattr(lib, f"interpn_hip_check_bounds_rectilinear_{sfx}").argtypes = [
            pp, sz, c_size_t, pp, sz, c_size_t, ct, POINTER(c_uint8), c_size_t]
    lib.interpn_hip_replicate.argtypes = [c_void_p, c_int, POINTER(c_void_p)]
    lib.interpn_hip_elem_size.argtypes = [c_void_p]
    lib.interpn_hip_ndims.argtypes = [c_void_p]
    lib.interpn_hip_device.argtypes = [c_void_p]
    lib.interpn_hip_eval_host.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_size_t), c_size_t, c_void_p, c_size_t]
    lib.interpn_hip_eval_host_sharded.argtypes = [POINTER(c_void_p), c_size_t, POINTER(c_void_p), POINTER(c_size_t),
                                                  c_size_t, c_void_p, c_size_t, POINTER(c_uint64)]
    lib.interpn_hip_eval_device_sharded.argtypes = [POINTER(c_void_p), c_size_t, POINTER(POINTER(c_void_p)), c_size_t,
                                                    POINTER(c_void_p), POINTER(c_size_t), POINTER(c_void_p), POINTER(c_uint64)]
    lib.interpn_hip_eval_device.argtypes = [c_void_p, POINTER(c_void_p), c_size_t, c_void_p, c_size_t, c_void_p]
    lib.interpn_hip_eval_device_ex.argtypes = [c_void_p, POINTER(c_void_p), c_size_t, c_void_p, c_size_t, c_void_p,
                                               ctypes.c_uint, POINTER(c_int), POINTER(c_int)]
    lib.interpn_hip_reserve.argtypes = [c_void_p, c_size_t, c_int]
    lib.interpn_hip_stage_ms.argtypes = [c_void_p, POINTER(c_double), c_size_t]
    lib.interpn_hip_check_bounds_device.argtypes = [c_void_p, POINTER(c_void_p), c_size_t, c_size_t, ctypes.c_double,
                                                    POINTER(ctypes.c_uint8), c_size_t, c_void_p]
    lib.interpn_hip_finish.argtypes = [c_void_p, c_void_p, POINTER(c_uint64)]
    lib.interpn_hip_set_blocks_per_cu.argtypes = [c_void_p, c_int]
    lib.interpn_hip_set_option.argtypes = [c_void_p, c_char_p, ctypes.c_longlong]
    lib.interpn_hip_get_option.argtypes = [c_void_p, c_char_p, POINTER(ctypes.c_longlong)]
    lib.interpn_hip_kernel_name.argtypes = [c_void_p, ctypes.c_char_p, c_size_t]
    lib.interpn_hip_table_bytes.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.interpn_hip_table_bytes.restype = c_size_t
    lib.interpn_hip_destroy.argtypes = [c_void_p]
    lib.interpn_hip_destroy.restype = None
    _lib = lib
    return lib


def strerror(status: int) -> str:
    return load().interpn_hip_strerror(status).decode()


def raise_for_status(status: int) -> None:
    """Map a status to the exception the reference's Python surface raises.

    Reference errors (`Err(&'static str)`) become AssertionError(msg), src/python.rs:77-79."""
    if status == OK:
        return
    msg = strerror(status)
    if status < ERR_REFERENCE_PANIC or status == ERR_TOO_MANY_DIMS_6:
        raise AssertionError(msg)
    if status == ERR_REFERENCE_PANIC:
        raise ReferencePanic(msg)
    if status == ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    detail = load().interpn_hip_last_hip_error().decode()
    raise InterpnHipError(f"{msg}: {detail}" if detail else msg)
