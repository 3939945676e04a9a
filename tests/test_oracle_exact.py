"""Oracle vs. an independent exact-rational evaluation of the same interpolants (CPU only).

Random (non-polynomial) data, ~15 % of the points extrapolated, both FMA flavours, all four
methods, N = 1..3.  The bound is a few ulp times the size of the data and the extrapolation
distance — orders of magnitude below north_star's 1e-12 / 1e-10."""

import numpy as np
import pytest

from oracle import exact_rational
from tests.test_oracle_kat import run_oracle
from tests import kat


def _case(method, kind, n, seed, linearize):
    rng = np.random.default_rng(seed)
    npts = 6
    if kind == "regular":
        starts = rng.uniform(-1, 1, n)
        steps = rng.uniform(0.1, 0.7, n)
        grids = [starts[d] + steps[d] * np.arange(npts) for d in range(n)]
    else:
        grids = [np.cumsum(rng.uniform(0.1, 0.7, npts)) - 1.0 for _ in range(n)]
    vals = rng.uniform(-1, 1, npts**n)
    obs = [rng.uniform(g[0] - 0.15 * (g[-1] - g[0]), g[-1] + 0.15 * (g[-1] - g[0]), 40) for g in grids]
    # add exact nodes and the domain ends
    for d in range(n):
        obs[d][:npts] = grids[d]
    c = kat.Case(f"{method}_{kind}_{n}", method, kind, grids, vals, obs, np.zeros(40), 0.0, linearize=linearize)
    if kind == "regular":
        c.extra["starts"], c.extra["steps"] = starts, steps
    return c


@pytest.mark.parametrize("fma", [False, True], ids=["nofma", "fma"])
@pytest.mark.parametrize("linearize", [False, True], ids=["quad", "lin"])
@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("method", ["linear", "cubic"])
def test_oracle_matches_exact(oracle, method, kind, n, linearize, fma):
    if method == "linear" and linearize:
        pytest.skip("flag only exists for cubic")
    c = _case(method, kind, n, 1000 + 10 * n + (method == "cubic"), linearize)
    if kind == "regular":
        # evaluate the oracle with the exact starts/steps (not grids[1]-grids[0])
        out = np.zeros(40)
        if method == "linear":
            oracle.linear_regular(c.dims, c.extra["starts"], c.extra["steps"], c.vals, c.obs, out, fma=fma)
        else:
            oracle.cubic_regular(c.dims, c.extra["starts"], c.extra["steps"], c.vals, linearize, c.obs, out, fma=fma)
        exact = exact_rational.evaluate(method, kind, c.grids, c.vals, c.obs, linearize,
                                        c.extra["starts"], c.extra["steps"])
    else:
        out = run_oracle(oracle, c, fma)
        exact = exact_rational.evaluate(method, kind, c.grids, c.vals, c.obs, linearize)
    err = np.array([abs(float(e - type(e)(float(o)))) for e, o in zip(exact, out)])
    scale = np.maximum(np.abs(np.array([float(e) for e in exact])), 1.0)
    tol = 2e-14 if method == "linear" else 2e-13
    assert np.max(err / scale) < tol, float(np.max(err / scale))
