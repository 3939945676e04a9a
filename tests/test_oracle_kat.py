"""Pins the CPU oracle against every hot-path known-answer test of the reference
(SURVEY.md §4 / §8(c)).  CPU only."""

import numpy as np
import pytest

from tests import kat


def run_oracle(oracle, case, fma, dtype=None):
    dtype = dtype or case.vals.dtype
    out = np.zeros(case.obs[0].size, dtype=dtype)
    if case.method == "nearest" and case.kind == "regular":
        oracle.nearest_regular(case.dims, case.starts, case.steps, case.vals, case.obs, out, fma=fma)
    elif case.method == "nearest":
        oracle.nearest_rectilinear(case.grids, case.vals, case.obs, out, fma=fma)
    elif case.method == "linear" and case.kind == "regular":
        oracle.linear_regular(case.dims, case.starts, case.steps, case.vals, case.obs, out, fma=fma)
    elif case.method == "linear":
        oracle.linear_rectilinear(case.grids, case.vals, case.obs, out, fma=fma)
    elif case.kind == "regular":
        oracle.cubic_regular(case.dims, case.starts, case.steps, case.vals, case.linearize, case.obs, out, fma=fma)
    else:
        oracle.cubic_rectilinear(case.grids, case.vals, case.linearize, case.obs, out, fma=fma)
    return out


RUST_CASES = [c for c in kat.all_cases(8, 6) if not c.name.startswith("py_")]
PY_CASES = [c for c in kat.all_cases(1, 1) if c.name.startswith("py_")]  # incl. py_near_*


@pytest.mark.parametrize("fma", [False, True], ids=["nofma", "fma"])
@pytest.mark.parametrize("case", RUST_CASES, ids=lambda c: c.name)
def test_rust_known_answers(oracle, case, fma):
    # CI of the reference runs `cargo test` and `cargo test --features=fma`
    # (.github/workflows/test-rust.yml:32-36): both flavours must satisfy the assertion.
    kat.check(case, run_oracle(oracle, case, fma))


@pytest.mark.parametrize("case", PY_CASES, ids=lambda c: c.name)
def test_python_known_answers(oracle, case):
    # The published wheels are built with the `fma` feature (pyproject.toml:72).
    kat.check(case, run_oracle(oracle, case, True))


def test_flattened_and_recursive_fma_sites(oracle):
    """The two arms of `interpn` differ only at documented FMA sites; without FMA a 6-D and a
    7-D evaluation of the same separable field must agree to rounding, and with FMA the
    N<=6 arm fuses index_zero_loc (multilinear/regular.rs:337) while N>=7 does not
    (regular_recursive.rs:310-313)."""
    rng = np.random.default_rng(7)
    # A step that is not exactly representable makes the fused and unfused index_zero_loc differ.
    starts, steps = np.array([0.1]), np.array([0.3])
    vals = rng.uniform(-1, 1, 8)
    obs = [rng.uniform(0.1, 0.1 + 0.3 * 7, 4096)]
    a = np.zeros(4096)
    b = np.zeros(4096)
    oracle.linear_regular([8], starts, steps, vals, obs, a, fma=True)
    oracle.linear_regular([8], starts, steps, vals, obs, b, fma=False)
    assert np.max(np.abs(a - b)) < 1e-14
    assert np.any(a != b)  # the feature does change bits, i.e. the switch is live
