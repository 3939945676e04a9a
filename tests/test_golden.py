"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU tier: the oracle still reproduces them bit for bit (no silent drift of the checker) and the
known-answer expectations hold.  GPU tier: the HIP path reproduces them bit for bit, using only
the committed data (nothing from /root/reference exists on the GPU box)."""

import os

import numpy as np
import pytest

from tests import kat
from tests.helpers import run_hip_raw, run_oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases(fname):
    z = np.load(os.path.join(GOLD, fname))
    names = sorted({k.split("/")[0] for k in z.files})
    cases = []
    for name in names:
        g = lambda key: z[f"{name}/{key}"]
        n = int(g("ndims"))
        c = kat.Case(name, str(g("method")), str(g("kind")), [g(f"grid{i}") for i in range(n)], g("vals"),
                     [g(f"obs{i}") for i in range(n)], g("expected"), float(g("atol")), linearize=bool(g("linearize")))
        c.extra["oracle_fma1"] = g("oracle_fma1")
        c.extra["oracle_fma0"] = g("oracle_fma0")
        if f"{name}/exact" in z.files:
            c.extra["exact"] = g("exact")
            c.extra["exact_scale"] = g("exact_scale")
        cases.append(c)
    return cases


KAT = load_cases("kat_cases.npz")
RND = load_cases("random_cases.npz")


def same(a, b):
    return bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


@pytest.mark.parametrize("case", KAT + RND, ids=lambda c: c.name)
def test_oracle_reproduces_golden(oracle, case):
    assert same(run_oracle(oracle, case, True), case.extra["oracle_fma1"])
    assert same(run_oracle(oracle, case, False), case.extra["oracle_fma0"])


# Bound against the exact-rational values stored in random_cases.npz (oracle/exact_rational.py,
# `fractions.Fraction`, tensor-product form: shares no code with the C++ oracle): a result computed
# in floating point with unit round-off u = eps/2 of its dtype lies within EXACT_K * u * scale of the
# exact value, scale = sum|w||v| + sum_d (|x_d| + max|g_d|) |dI/dx_d| per point (stored with the
# fixture).  Measured: the oracle's worst ratio over all 60 workloads is 3.5 (cubic rectilinear:
# ~10 divides per node); typical 0.05.  A transcription error in the Hermite c3 term
# (multicubic/mod.rs:72-91) changes results by ~1e-2 * scale: 13 orders of magnitude above this.
EXACT_K = 16.0


def within_exact_bound(got, case):
    u = np.finfo(case.vals.dtype).eps / 2
    err = np.abs(got.astype(np.float64) - case.extra["exact"])
    return err <= EXACT_K * u * case.extra["exact_scale"], float(np.max(err / (u * case.extra["exact_scale"])))


@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("case", RND, ids=lambda c: c.name)
def test_oracle_within_exact_rational_bound(oracle, case, fma):
    ok, worst = within_exact_bound(run_oracle(oracle, case, fma), case)
    assert np.all(ok), (case.name, worst)


@pytest.mark.gpu
@pytest.mark.parametrize("case", RND, ids=lambda c: c.name)
def test_hip_within_exact_rational_bound(case):
    """The one check of the HIP path that does not pass through the builder's C++ restatement:
    every linear and cubic workload of random_cases.npz against the exact interpolant."""
    got = run_hip_raw(case)
    ok, worst = within_exact_bound(got, case)
    assert np.all(ok), (case.name, worst)
    # and in north_star's terms (<= 1e-12 linear / 1e-10 cubic, relative, f64)
    if case.vals.dtype == np.float64:
        rel = np.abs(got - case.extra["exact"]) / np.maximum(np.abs(case.extra["exact"]), 1.0)
        assert float(rel.max()) <= (1e-12 if case.method == "linear" else 1e-10), (case.name, float(rel.max()))


@pytest.mark.gpu
def test_hip_within_exact_rational_bound_without_fma():
    from interpn_amd import _lib

    lib = _lib.load()
    prev = lib.interpn_hip_set_fma(0)
    try:
        for case in RND:
            ok, worst = within_exact_bound(run_hip_raw(case), case)
            assert np.all(ok), (case.name, worst)
    finally:
        lib.interpn_hip_set_fma(prev)


@pytest.mark.gpu
@pytest.mark.parametrize("case", KAT + RND, ids=lambda c: c.name)
def test_hip_reproduces_golden(case):
    got = run_hip_raw(case)
    assert got.dtype == case.vals.dtype
    assert same(got, case.extra["oracle_fma1"]), case.name


@pytest.mark.gpu
def test_hip_reproduces_golden_without_fma():
    from interpn_amd import _lib

    lib = _lib.load()
    prev = lib.interpn_hip_set_fma(0)
    try:
        for case in RND:
            assert same(run_hip_raw(case), case.extra["oracle_fma0"]), case.name
    finally:
        lib.interpn_hip_set_fma(prev)
