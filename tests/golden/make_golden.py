#!/usr/bin/env python3
"""Generates tests/golden/*.npz.

The reference (jlogan03/interpn v0.8.2) can be neither compiled (no rustc) nor imported (its
Python package needs the compiled cdylib) in this environment, so these fixtures are NOT
outputs of the reference itself.  They freeze
  (a) the inputs and analytic expectations of the reference's own known-answer tests
      (re-created by tests/kat.py, each citing the reference test it mirrors), and
  (b) outputs of the pinned CPU oracle (oracle/interpn_oracle.cpp, both `fma` flavours) on small
      seeded random workloads with extrapolation and special points,
  (c) for every workload of (b): the EXACT value of the interpolant at each point
      (oracle/exact_rational.py: `fractions.Fraction`, tensor-product form — no rounding, and no
      code shared with the C++ oracle) rounded once to f64, and the rounding-error scale
      sum|w||v| + sum_d (|x_d| + max|g_d|) |dI/dx_d| the tests multiply by 4 u,
so that the oracle cannot drift silently and the GPU path can be checked — against the oracle AND
against something that is not the oracle — on a box where only the committed data travels.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import exact_rational, pyoracle  # noqa: E402
from tests import kat  # noqa: E402
from tests.helpers import run_oracle, synthetic_case  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def pack(case, extra=None):
    d = {
        "method": np.array(case.method),
        "kind": np.array(case.kind),
        "linearize": np.array(case.linearize),
        "vals": case.vals,
        "ndims": np.array(len(case.grids)),
        "expected": case.expected,
        "atol": np.array(case.atol),
    }
    for i, g in enumerate(case.grids):
        d[f"grid{i}"] = g
    for i, o in enumerate(case.obs):
        d[f"obs{i}"] = o
    d.update(extra or {})
    return d


def main():
    pyoracle.build()
    # (a) known-answer cases, small dimensions only (the big meshes are regenerated at test time)
    kats = [c for c in kat.all_cases(4, 3)]
    blob = {}
    for c in kats:
        for k, v in pack(c, {"oracle_fma1": run_oracle(pyoracle, c, True),
                             "oracle_fma0": run_oracle(pyoracle, c, False)}).items():
            blob[f"{c.name}/{k}"] = v
    np.savez_compressed(os.path.join(OUT, "kat_cases.npz"), **blob)
    # (b) seeded random workloads
    blob = {}
    specs = [("linear", "regular", 1, [33]), ("linear", "regular", 3, [9, 7, 8]), ("linear", "rectilinear", 2, [12, 9]),
             ("linear", "rectilinear", 4, [5, 4, 6, 3]), ("cubic", "regular", 2, [9, 8]), ("cubic", "regular", 4, [5, 6, 4, 7]),
             ("cubic", "rectilinear", 1, [17]), ("cubic", "rectilinear", 3, [6, 5, 7]), ("linear", "regular", 7, [2, 3, 2, 2, 3, 2, 2]),
             ("cubic", "rectilinear", 5, [4, 4, 5, 4, 4])]
    for dtype in (np.float64, np.float32):
        for lin in (False, True):
            for m, k, n, axis in specs:
                if m == "linear" and lin:
                    continue
                c = synthetic_case(m, k, n, axis, 500, 7000 + n, dtype, linearize=lin, extrap=0.25)
                c.name = f"{m}_{k}_N{n}_{'f64' if dtype == np.float64 else 'f32'}_lin{int(lin)}"
                ex = exact_rational.evaluate_with_condition(m, k, c.grids, c.vals, c.obs, lin, c.starts, c.steps)
                extra = {"oracle_fma1": run_oracle(pyoracle, c, True), "oracle_fma0": run_oracle(pyoracle, c, False),
                         "exact": np.array([float(v) for v, _ in ex]),  # float(Fraction) rounds correctly, once
                         "exact_scale": np.array([float(sc) for _, sc in ex])}
                for key, v in pack(c, extra).items():
                    blob[f"{c.name}/{key}"] = v
                print(c.name, flush=True)
    np.savez_compressed(os.path.join(OUT, "random_cases.npz"), **blob)
    for f in ("kat_cases.npz", "random_cases.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
