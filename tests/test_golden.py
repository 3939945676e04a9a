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
