"""The C++ mirror of the reference crate's Rust API (include/interpn_hip.hpp) and the reference's
own Rust unit tests / doctests re-created against it (tests/cpp/reference_tests.cpp): the
compiled-language host side above the C ABI, since no Rust toolchain exists in this image.

CPU tier: the header and the test program compile and link with plain g++ (no hipcc, no HIP
headers) and the program refuses to run without a device.  GPU tier: every re-created test passes.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "interpn_amd")


def build(tmp_path, extra=()):
    exe = str(tmp_path / "reference_tests")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp"), "-L", LIBDIR, "-linterpn_hip",
                           f"-Wl,-rpath,{LIBDIR}", "-o", exe, *extra])
    return exe


def test_cpp_mirror_builds_with_plain_gxx(tmp_path):
    exe = build(tmp_path)
    assert os.path.exists(exe)
    # the header is also valid on its own, in C++17 pedantic mode
    probe = tmp_path / "probe.cpp"
    probe.write_text('#include "interpn_hip.hpp"\nint main() { return interpn_hip::utils::linspace(0.0, 1.0, 3).size() == 3 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(probe), "-L", LIBDIR, "-linterpn_hip", f"-Wl,-rpath,{LIBDIR}", "-o", str(tmp_path / "probe")])
    assert subprocess.run([str(tmp_path / "probe")]).returncode == 0


def test_cpp_mirror_refuses_without_a_device(tmp_path):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    res = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 2 and "no HIP device" in res.stdout


@pytest.mark.gpu
def test_reference_rust_tests_through_the_cpp_mirror(tmp_path):
    """multilinear / multicubic / nearest x regular / rectilinear: the extrapolation-corner sweeps
    (N = 1..8), hat functions (exact), quadratic and sine reproduction, doctests, error strings,
    the abort-at-first-bad-point contract, check_bounds, f32 — src/*/**.rs `mod test`."""
    res = subprocess.run([build(tmp_path)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ALL PASSED" in res.stdout and "FAIL" not in res.stdout
    assert res.stdout.count("PASS ") == 20


@pytest.mark.gpu
@pytest.mark.parametrize("feature", ["1", "0"], ids=["fma", "nofma"])
def test_reference_rust_tests_with_the_fma_feature_chosen_at_compile_time(tmp_path, feature):
    """`-DINTERPN_HIP_FEATURE_FMA=1|0` is the C++ counterpart of building the crate with or without
    `--features fma` (Cargo.toml:34-38): the structs then carry that flavour (a per-handle property
    of the ABI) and the crate's tests — whose tolerances hold for both cargo flavours — pass."""
    res = subprocess.run([build(tmp_path, extra=(f"-DINTERPN_HIP_FEATURE_FMA={feature}",))], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ALL PASSED" in res.stdout and "FAIL" not in res.stdout
