"""Shared helpers for the parity tests: seeded synthetic workloads and
dispatch of a `kat.Case` to either the oracle or the HIP path."""

from __future__ import annotations

import numpy as np


def run_oracle(oracle, case, fma=True, dtype=None, out=None):
    dtype = dtype or case.vals.dtype
    if out is None:
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


def run_hip_raw(case, dtype=None, out=None):
    """Through the reference-named raw functions -> C ABI one-shot entry points."""
    from interpn_amd import raw

    dtype = np.dtype(dtype or case.vals.dtype)
    sfx = "f64" if dtype == np.float64 else "f32"
    cv = lambda a: np.ascontiguousarray(a, dtype=dtype)
    if out is None:
        out = np.zeros(case.obs[0].size, dtype=dtype)
    obs = [cv(o) for o in case.obs]
    if case.method == "nearest" and case.kind == "regular":
        getattr(raw, f"interpn_nearest_regular_{sfx}")(case.dims, cv(case.starts), cv(case.steps), cv(case.vals), obs, out)
    elif case.method == "nearest":
        getattr(raw, f"interpn_nearest_rectilinear_{sfx}")([cv(g) for g in case.grids], cv(case.vals), obs, out)
    elif case.method == "linear" and case.kind == "regular":
        getattr(raw, f"interpn_linear_regular_{sfx}")(case.dims, cv(case.starts), cv(case.steps), cv(case.vals), obs, out)
    elif case.method == "linear":
        getattr(raw, f"interpn_linear_rectilinear_{sfx}")([cv(g) for g in case.grids], cv(case.vals), obs, out)
    elif case.kind == "regular":
        getattr(raw, f"interpn_cubic_regular_{sfx}")(case.dims, cv(case.starts), cv(case.steps), cv(case.vals),
                                                      case.linearize, obs, out)
    else:
        getattr(raw, f"interpn_cubic_rectilinear_{sfx}")([cv(g) for g in case.grids], cv(case.vals), case.linearize,
                                                          obs, out)
    return out


def synthetic_case(method, kind, n, npts_axis, nobs, seed, dtype=np.float64, linearize=False, extrap=0.05,
                   specials=True):
    """Synthetic workload of SURVEY.md §8(d): axes linspace(-1,1,n) (rectilinear: interior nodes
    jittered by up to a quarter step), vals U(-1,1), obs i.i.d. uniform over the grid extent
    widened by `extrap` on each side, plus injected special points (exact nodes, domain ends,
    +-0)."""
    from tests.kat import Case

    rng = np.random.default_rng(seed)
    grids = []
    for d in range(n):
        g = np.linspace(-1.0, 1.0, npts_axis[d])
        if kind == "rectilinear":
            step = g[1] - g[0]
            j = (rng.random(g.size) - 0.5) * 0.5 * step
            j[0] = j[-1] = 0.0
            g = g + j
        g = g.astype(dtype)
        assert np.all(np.diff(g) > 0)
        grids.append(g)
    vals = rng.uniform(-1.0, 1.0, int(np.prod(npts_axis))).astype(dtype)
    obs = []
    for d in range(n):
        lo, hi = float(grids[d][0]), float(grids[d][-1])
        w = (hi - lo) * extrap
        o = rng.uniform(lo - w, hi + w, nobs).astype(dtype)
        if specials and nobs >= 64:
            k = min(grids[d].size, 24)
            o[:k] = grids[d][:k]  # exact nodes
            o[k] = grids[d][-1]
            o[k + 1] = grids[d][0]
            o[k + 2] = 0.0
            o[k + 3] = -0.0
            o[k + 4] = np.nextafter(grids[d][-1], np.inf, dtype=dtype)
            o[k + 5] = np.nextafter(grids[d][0], -np.inf, dtype=dtype)
            o[k + 6] = grids[d][-2]
            o[k + 7] = grids[d][1]
        obs.append(o)
    if specials and nobs >= 64 and n > 1:
        # de-correlate the special rows across dimensions
        for d in range(1, n):
            obs[d][:40] = np.roll(obs[d][:40], 3 * d)
    return Case(f"syn_{method}_{kind}_N{n}", method, kind, grids, vals, obs, np.zeros(nobs, dtype=dtype), 0.0,
                linearize=linearize)


def rel_err(a, b):
    """|a-b| / max(|b|, 1) — the reference's normalisation (test/test_multicubic_regular.py:97-100)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)
