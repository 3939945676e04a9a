"""CPU-only checks of the C ABI: the library loads, exports every symbol include/interpn_hip.h
declares, maps statuses to the reference's exact error strings, and validates arguments in the
reference's order WITHOUT touching a device.  No compute calls here (no GPU in this tier)."""

import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "interpn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set()
    # macro-generated families
    for macro, body in re.findall(r"#define (INTERPN_HIP_DECLARE_\w+)\(T, SUFFIX\)(.*?)\n\n", text, flags=re.S):
        fams = re.findall(r"(interpn_hip_\w+?)_##SUFFIX", body)
        for sfx in re.findall(macro + r"\(\w+, (\w+)\)", text):
            if sfx == "SUFFIX":
                continue  # the #define line itself
            names.update(f"{f}_{sfx}" for f in fams)
    # plain declarations
    plain = re.sub(r"#define.*?\n\n", "\n", text, flags=re.S)
    names.update(re.findall(r"\b(interpn_hip_\w+)\s*\(", plain))
    names.discard("interpn_hip_status")
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    from interpn_amd import _lib

    return _lib.load()


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 29, syms
    for fam in ("linear_regular", "linear_rectilinear", "cubic_regular", "cubic_rectilinear", "create_regular",
                "create_rectilinear", "check_bounds_regular", "check_bounds_rectilinear"):
        for sfx in ("f64", "f32"):
            assert f"interpn_hip_{fam}_{sfx}" in syms
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/interpn_hip.h but not exported"


def test_signatures_have_no_torch_types():
    text = open(os.path.join(ROOT, "include", "interpn_hip.h")).read()
    assert "torch" not in text and "at::" not in text and "#include <hip" not in text


def test_strerror_matches_reference_strings(lib):
    """Messages of src/multilinear/regular.rs:61,240,245,250,418,112; rectilinear.rs:192;
    multicubic/regular.rs:261; multicubic/rectilinear.rs:214."""
    want = {
        1: "Dimension mismatch",
        2: "All grids must have at least two entries",
        3: "All grids must have at least 2 entries",
        4: "All grids must have at least four entries",
        5: "All grids must have at least 4 entries",
        6: "All grids must be monotonically increasing",
        7: "Unrepresentable coordinate value",
        8: "Dimension exceeds maximum (8). Use interpolator struct directly for higher dimensions.",
    }
    for code, msg in want.items():
        assert lib.interpn_hip_strerror(code).decode() == msg
    assert lib.interpn_hip_strerror(0).decode() == ""


def _lin_reg(dims, starts, steps, vals, obs, out):
    from interpn_amd import raw

    raw.interpn_linear_regular_f64(dims, starts, steps, vals, obs, out)


def test_validation_order_without_device():
    """Every rejected call must fail with the reference's message before any device work."""
    from interpn_amd import raw
    from interpn_amd._lib import ReferencePanic

    z3 = [np.zeros(3), np.zeros(3)]
    o3 = np.zeros(3)
    with pytest.raises(AssertionError, match="^Dimension mismatch$"):  # starts.len() != ndims
        _lin_reg([2, 2], np.zeros(1), np.ones(2), np.zeros(4), z3, o3)
    with pytest.raises(AssertionError, match="^Dimension mismatch$"):  # obs.len() != ndims
        _lin_reg([2, 2], np.zeros(2), np.ones(2), np.zeros(4), z3[:1], o3)
    with pytest.raises(AssertionError, match="^Dimension mismatch$"):  # vals.len() != prod(dims)
        _lin_reg([2, 2], np.zeros(2), np.ones(2), np.zeros(5), z3, o3)
    with pytest.raises(AssertionError, match="at least two entries"):
        _lin_reg([2, 1], np.zeros(2), np.ones(2), np.zeros(2), z3, o3)
    with pytest.raises(AssertionError, match="monotonically increasing"):
        _lin_reg([2, 2], np.zeros(2), np.array([1.0, 0.0]), np.zeros(4), z3, o3)
    with pytest.raises(AssertionError, match="monotonically increasing"):  # NaN step: !(x > 0)
        _lin_reg([2, 2], np.zeros(2), np.array([1.0, np.nan]), np.zeros(4), z3, o3)
    with pytest.raises(AssertionError, match="^Dimension mismatch$"):  # obs lengths != out length
        _lin_reg([2, 2], np.zeros(2), np.ones(2), np.zeros(4), [np.zeros(3), np.zeros(2)], o3)
    with pytest.raises(AssertionError, match="Dimension exceeds maximum"):
        _lin_reg([], np.zeros(0), np.zeros(0), np.zeros(1), [], o3)
    # the vals check comes before the degenerate-grid check, as in `new` (regular.rs:238-246)
    with pytest.raises(AssertionError, match="^Dimension mismatch$"):
        _lin_reg([2, 1], np.zeros(2), np.ones(2), np.zeros(3), z3, o3)

    g2 = [np.array([0.0, 1.0]), np.array([0.0, 1.0])]
    with pytest.raises(AssertionError, match="at least 2 entries"):
        raw.interpn_linear_rectilinear_f64([np.array([0.0]), g2[1]], np.zeros(2), z3, o3)
    with pytest.raises(AssertionError, match="monotonically increasing"):
        raw.interpn_linear_rectilinear_f64([np.array([1.0, 1.0]), g2[1]], np.zeros(4), z3, o3)
    with pytest.raises(AssertionError, match="^Dimension mismatch$"):
        raw.interpn_linear_rectilinear_f64(g2, np.zeros(4), z3[:1], o3)

    with pytest.raises(AssertionError, match="at least four entries"):
        raw.interpn_cubic_regular_f64([4, 3], np.zeros(2), np.ones(2), np.zeros(12), False, z3, o3)
    with pytest.raises(AssertionError, match="at least 4 entries"):
        raw.interpn_cubic_rectilinear_f64([np.arange(4.0), np.arange(3.0)], np.zeros(12), False, z3, o3)
    # multicubic::regular::interpn panics on mismatched slice lengths for N <= 4
    # (`starts.try_into().unwrap()`, multicubic/regular.rs:66-73) ...
    with pytest.raises(ReferencePanic):
        raw.interpn_cubic_regular_f64([4, 4], np.zeros(1), np.ones(2), np.zeros(16), False, z3, o3)
    # ... but reports "Dimension mismatch" from the recursive arm (N >= 5)
    with pytest.raises(AssertionError, match="^Dimension mismatch$"):
        raw.interpn_cubic_regular_f64([4] * 5, np.zeros(4), np.ones(5), np.zeros(4**5), False, [np.zeros(3)] * 5, o3)


def test_raw_rejects_wrong_dtype_and_layout():
    from interpn_amd import raw

    z = [np.zeros(3), np.zeros(3)]
    with pytest.raises(TypeError):
        raw.interpn_linear_regular_f64([2, 2], np.zeros(2, dtype=np.float32), np.ones(2), np.zeros(4), z, np.zeros(3))
    with pytest.raises(ValueError, match="not contiguous"):
        raw.interpn_linear_regular_f64([2, 2], np.zeros(2), np.ones(2), np.zeros(8)[::2], z, np.zeros(3))
    with pytest.raises(TypeError):
        raw.interpn_linear_regular_f64([2, 2], np.zeros(2), np.ones(2), np.zeros((2, 2)), z, np.zeros(3))


def test_valid_call_fails_loudly_without_gpu(lib):
    """No CPU fallback: on a machine without a HIP device a *valid* call must raise, not compute."""
    if lib.interpn_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    from interpn_amd import raw
    from interpn_amd._lib import InterpnHipError

    out = np.full(3, -7.0)
    with pytest.raises(InterpnHipError):
        raw.interpn_linear_regular_f64([2, 2], np.zeros(2), np.ones(2), np.zeros(4), [np.zeros(3), np.zeros(3)], out)
    assert np.all(out == -7.0)


def test_python_classes_validate_like_the_reference():
    """Validators of src/interpn/multilinear_regular.py:73-96 and multilinear_rectilinear.py:67-89."""
    import interpn_amd

    with pytest.raises(AssertionError, match="Size of value array does not match grid dims"):
        interpn_amd.MultilinearRegular.new([2, 2], np.zeros(2), np.ones(2), np.zeros(5))
    with pytest.raises(AssertionError, match="All grid steps must be positive and nonzero"):
        interpn_amd.MultilinearRegular.new([2, 2], np.zeros(2), np.array([1.0, 0.0]), np.zeros(4))
    with pytest.raises(AssertionError, match="monotonically increasing"):
        interpn_amd.MultilinearRectilinear.new([np.array([0.0, 1.0, 0.5]), np.array([0.0, 1.0])], np.zeros(6))
    it = interpn_amd.MulticubicRegular.new([4, 4], np.zeros(2), np.ones(2), np.zeros(16))
    assert it.linearize_extrapolation is True  # multicubic_regular.py:59
    assert it.ndims() == 2
    rt = interpn_amd.MulticubicRegular.model_validate_json(it.model_dump_json())
    assert rt.dims == it.dims and np.array_equal(rt.vals, it.vals) and rt.linearize_extrapolation is True
    with pytest.raises(TypeError):
        it.dims = [1]


def test_header_is_valid_c99_and_c_consumer_compiles(tmp_path):
    """The boundary is a C ABI: `include/interpn_hip.h` must compile as plain C99 (no C++isms, no
    HIP/torch types) and the pure-C consumer in examples/ must compile and link against the
    library (running it needs a GPU: tests/test_gpu_parity.py::test_c_consumer_runs)."""
    import subprocess

    inc = os.path.join(ROOT, "include")
    src = os.path.join(ROOT, "examples", "c_abi_demo.c")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", inc, "-fsyntax-only", src])
    probe = tmp_path / "only_header.c"
    probe.write_text('#include "interpn_hip.h"\nint main(void) { return INTERPN_HIP_OK; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Werror", "-I", inc, "-c", str(probe), "-o", str(tmp_path / "p.o")])
    libdir = os.path.join(ROOT, "interpn_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I", inc, src, "-L", libdir, "-linterpn_hip", f"-Wl,-rpath,{libdir}",
                           "-lm", "-o", str(tmp_path / "demo")])


def test_raw_stub_matches_module():
    """interpn_amd/raw.pyi (+ py.typed) mirrors the reference's typed surface
    (src/interpn/raw.pyi:32-147): same 16 names, same parameter names in the same order as the
    functions raw.py really defines; the committed stub is what tools/gen_raw_stub.py renders."""
    import ast
    import inspect

    from interpn_amd import raw
    from tools.gen_raw_stub import render

    here = os.path.join(ROOT, "interpn_amd")
    assert os.path.exists(os.path.join(here, "py.typed"))
    text = open(os.path.join(here, "raw.pyi")).read()
    assert text == render()
    stub = {n.name: [a.arg for a in n.args.args] for n in ast.parse(text).body if isinstance(n, ast.FunctionDef)}
    assert sorted(stub) == sorted(raw.__all__) and len(stub) == 16
    for name, params in stub.items():
        assert list(inspect.signature(getattr(raw, name)).parameters) == params, name
