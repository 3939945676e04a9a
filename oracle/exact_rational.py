"""TEST INFRASTRUCTURE — exact-rational evaluation of the interpolants (small cases only).

Independent of the C++ oracle: it is derived from the *mathematical* definition of the
reference's interpolants (docs of src/multilinear/regular.rs, src/multicubic/mod.rs:1-46,
src/multicubic/regular.rs:167-193), written in Hermite-basis form rather than the reference's
Horner form, and evaluated in `fractions.Fraction` so that there is no rounding at all.  It
bounds the oracle's rounding error and catches transcription errors the polynomial-field
known answers cannot see (a wrong cubic term on non-polynomial data).

Cell selection is done in exact arithmetic; the interpolants are continuous (linear) / C1
(cubic) across cells, so a point that floating point assigns to the neighbouring cell still
agrees to rounding.
"""

from __future__ import annotations

import math
from fractions import Fraction as F


def _cell_regular(x, start, step, n, footprint):
    # lower index of the cell [i, i+1] that contains x, clamped to the grid
    i = math.floor((x - start) / step)
    return max(0, min(n - 2, i))


def _cell_rect(x, g):
    n = len(g)
    cnt = sum(1 for v in g if v < x)  # partition_point(g < x)
    return max(0, min(n - 2, cnt - 1))


def _lin1d(x, g, y, i):
    t = (x - g[i]) / (g[i + 1] - g[i])
    return y[i] + t * (y[i + 1] - y[i])


def _slope(g, y, i):
    """Physical-units slope at interior node i: distance-weighted central difference
    (Veldman & Rinzema method B, src/multicubic/mod.rs:93-117); equals (y[i+1]-y[i-1])/(2h)
    on a uniform grid (src/multicubic/regular.rs:507-508)."""
    h0 = g[i] - g[i - 1]
    h1 = g[i + 1] - g[i]
    return (h0 / (h0 + h1)) * (y[i + 1] - y[i]) / h1 + (h1 / (h0 + h1)) * (y[i] - y[i - 1]) / h0


def _cubic1d(x, g, y, i, linearize):
    """Hermite cubic on cell [i, i+1]; natural-spline (zero third derivative) closure in the
    first and last cell; quadratic continuation or slope-hold outside the grid."""
    n = len(g)
    h = g[i + 1] - g[i]
    dy = y[i + 1] - y[i]
    if i == 0:
        m1 = _slope(g, y, 1)
        m0 = 2 * dy / h - m1  # q''' = 0 at the boundary node
    elif i == n - 2:
        m0 = _slope(g, y, i)
        m1 = 2 * dy / h - m0
    else:
        m0 = _slope(g, y, i)
        m1 = _slope(g, y, i + 1)
    if linearize and x < g[0]:
        return y[0] + m0 * (x - g[0])
    if linearize and x > g[n - 1]:
        return y[n - 1] + m1 * (x - g[n - 1])
    t = (x - g[i]) / h
    h00 = 2 * t**3 - 3 * t**2 + 1
    h10 = t**3 - 2 * t**2 + t
    h01 = -2 * t**3 + 3 * t**2
    h11 = t**3 - t**2
    return h00 * y[i] + h10 * h * m0 + h01 * y[i + 1] + h11 * h * m1


def _eval(method, grids, vals, shape, x, linearize, regular):
    """Tensor-product evaluation, reducing the LAST dimension first (any order is exact)."""
    nd = len(grids)

    def rec(d, prefix):
        # value of the interpolant restricted to fixed indices `prefix` on dims < d... reduce dim d..nd-1
        if d == nd:
            idx = 0
            for k in range(nd):
                idx = idx * shape[k] + prefix[k]
            return vals[idx]
        g = grids[d]
        if regular is not None:
            start, step = regular[d]
            i = _cell_regular(x[d], start, step, len(g), 2)
        else:
            i = _cell_rect(x[d], g)
        if method == "linear":
            need = [i, i + 1]
        else:
            need = sorted({k for k in (i - 1, i, i + 1, i + 2) if 0 <= k < len(g)})
        y = {k: rec(d + 1, prefix + [k]) for k in need}
        if method == "linear":
            return _lin1d(x[d], g, y, i)
        return _cubic1d(x[d], g, y, i, linearize)

    return rec(0, [])


def evaluate(method, kind, grids, vals, obs, linearize=False, starts=None, steps=None):
    """Return a list of Fractions, one per observation point.

    For kind == "regular" the grid is the one the reference *implies*: node i of dim d sits at
    starts[d] + i*steps[d] exactly (not at the rounded f64 grids[d][i])."""
    shape = [len(g) for g in grids]
    fv = [F(float(v)) for v in vals]
    if kind == "regular":
        reg = [(F(float(starts[d])), F(float(steps[d]))) for d in range(len(grids))]
        fg = [[reg[d][0] + k * reg[d][1] for k in range(shape[d])] for d in range(len(grids))]
    else:
        reg = None
        fg = [[F(float(v)) for v in g] for g in grids]
    out = []
    for k in range(len(obs[0])):
        x = [F(float(obs[d][k])) for d in range(len(grids))]
        out.append(_eval(method, fg, fv, shape, x, linearize, reg))
    return out


# ---------------------------------------------------------------------------------------------
# Tensor-product form with a condition number (round 3).  Every 1-D operator above is linear in
# the node values, so the N-D interpolant is  sum_k  prod_d W_d[k_d](x_d) * vals[k]  with per-
# dimension weight vectors W_d that depend on x_d only.  This second formulation (weights first,
# one contraction) shares nothing with the dimension-by-dimension recursion of `_eval` except the
# 1-D formulas, returns exactly the same rational, and yields the quantities a rounding-error bound
# needs:  sum |W...| |vals|  (sensitivity to rounding in the value tree)  and  |dI/dx_d|
# (sensitivity to the rounding of t = (x - x_i) / h, whose absolute error is ~ u (|x| + |x_i|) / h).


def _weights_1d(method, g, x, linearize, reg):
    n = len(g)
    if reg is not None:
        i = _cell_regular(x, reg[0], reg[1], n, 2)
    else:
        i = _cell_rect(x, g)
    if method == "linear":
        t = (x - g[i]) / (g[i + 1] - g[i])
        return {i: 1 - t, i + 1: t}
    need = sorted({k for k in (i - 1, i, i + 1, i + 2) if 0 <= k < n})
    w = {}
    for k in need:
        unit = {j: (F(1) if j == k else F(0)) for j in need}
        w[k] = _cubic1d(x, g, unit, i, linearize)
    return w


def _contract(ws, vals, shape, absolute):
    nd = len(ws)

    def rec(d, base):
        tot = F(0)
        for k, w in ws[d].items():
            idx = base * shape[d] + k
            sub = (abs(vals[idx]) if absolute else vals[idx]) if d == nd - 1 else rec(d + 1, idx)
            tot += (abs(w) if absolute else w) * sub
        return tot

    return rec(0, 0)


def evaluate_with_condition(method, kind, grids, vals, obs, linearize=False, starts=None, steps=None):
    """Per observation point: (exact value, bound scale), both Fractions.

    `bound scale` = sum_k |prod_d W_d[k_d]| |vals[k]|  +  sum_d (|x_d| + max|g_d|) * |dI/dx_d|,
    the derivative taken as an exact central difference over 2^-24 of the local cell width.  A
    floating-point evaluation with unit round-off u that performs the reference's operations in
    any order differs from the exact value by a modest multiple of u * scale."""
    nd = len(grids)
    shape = [len(g) for g in grids]
    fv = [F(float(v)) for v in vals]
    if kind == "regular":
        reg = [(F(float(starts[d])), F(float(steps[d]))) for d in range(nd)]
        fg = [[reg[d][0] + k * reg[d][1] for k in range(shape[d])] for d in range(nd)]
    else:
        reg = [None] * nd
        fg = [[F(float(v)) for v in g] for g in grids]
    gmax = [max(abs(fg[d][0]), abs(fg[d][-1])) for d in range(nd)]
    out = []
    for p in range(len(obs[0])):
        x = [F(float(obs[d][p])) for d in range(nd)]
        if any(not math.isfinite(float(obs[d][p])) for d in range(nd)):
            out.append((None, None))
            continue
        ws = [_weights_1d(method, fg[d], x[d], linearize, reg[d]) for d in range(nd)]
        val = _contract(ws, fv, shape, False)
        scale = _contract(ws, fv, shape, True)
        for d in range(nd):
            i0 = min(ws[d])
            h = fg[d][i0 + 1] - fg[d][i0]
            delta = h / (1 << 24)
            wp = list(ws)
            wm = list(ws)
            wp[d] = _weights_1d(method, fg[d], x[d] + delta, linearize, reg[d])
            wm[d] = _weights_1d(method, fg[d], x[d] - delta, linearize, reg[d])
            deriv = abs(_contract(wp, fv, shape, False) - _contract(wm, fv, shape, False)) / (2 * delta)
            scale += (abs(x[d]) + gmax[d]) * deriv
        out.append((val, scale))
    return out
