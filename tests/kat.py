"""Known-answer cases re-created from the reference's own tests.

The reference's tests cannot run here (no rustc, no interpn wheel), so each builder below
re-creates the *inputs* of one reference test with the reference's own helper semantics
(`linspace` = start + i*dx, src/utils.rs:8-14; `meshgrid` = C-ordered cartesian product,
src/utils.rs:17-25) and returns the *analytic expectation* that test asserts, with the
tolerance it asserts.  The same cases are used for the oracle (CPU) and the HIP path (GPU).

Where the reference perturbs axes with its fixed-seed StdRng (src/testing.rs:6-25, not
reproducible here), any strictly increasing perturbation is a valid stand-in because the
assertion is an analytic identity; numpy's PCG64 with a fixed seed is used instead.
"""

from __future__ import annotations

import itertools
from dataclasses import dataclass, field

import numpy as np


def linspace_ref(start: float, stop: float, n: int) -> np.ndarray:
    """src/utils.rs:8-14 — NOT np.linspace (differs in the last bits)."""
    dx = (np.float64(stop) - np.float64(start)) / np.float64(n - 1)
    return np.array([np.float64(start) + np.float64(i) * dx for i in range(n)], dtype=np.float64)


def meshgrid_ref(axes) -> np.ndarray:
    """src/utils.rs:17-25 — rows are points, C ordering (last axis fastest)."""
    return np.array(list(itertools.product(*[list(a) for a in axes])), dtype=np.float64).reshape(-1, len(axes))


def seq_sum(rows: np.ndarray) -> np.ndarray:
    """Left-to-right f64 sum of each row, as `x.iter().sum()` does."""
    acc = np.zeros(rows.shape[0], dtype=np.float64)
    for j in range(rows.shape[1]):
        acc = acc + rows[:, j]
    return acc


def seq_sum_sq(rows: np.ndarray) -> np.ndarray:
    acc = np.zeros(rows.shape[0], dtype=np.float64)
    for j in range(rows.shape[1]):
        acc = acc + rows[:, j] * rows[:, j]
    return acc


def seq_sum_sin(rows: np.ndarray) -> np.ndarray:
    acc = np.zeros(rows.shape[0], dtype=np.float64)
    for j in range(rows.shape[1]):
        acc = acc + np.sin(rows[:, j] * 6.28 / 10.0)
    return acc


@dataclass
class Case:
    name: str
    method: str  # "linear" | "cubic"
    kind: str  # "regular" | "rectilinear"
    grids: list  # axis coordinate arrays (always present; regular cases derive dims/starts/steps)
    vals: np.ndarray
    obs: list  # SoA: one array per dim
    expected: np.ndarray
    atol: float  # absolute tolerance asserted by the reference test; 0.0 => exact ==
    linearize: bool = False
    ref: str = ""  # reference test file:line
    extra: dict = field(default_factory=dict)

    @property
    def dims(self):
        return [len(g) for g in self.grids]

    @property
    def starts(self):
        return np.array([g[0] for g in self.grids], dtype=self.vals.dtype)

    @property
    def steps(self):
        # the reference tests use x[1] - x[0] (e.g. multilinear/regular.rs:449)
        return np.array([g[1] - g[0] for g in self.grids], dtype=self.vals.dtype)


def _jitter(rng, x):
    dx = rng.random(x.size)
    y = x + (dx - 0.5) / 10.0  # multilinear/rectilinear.rs:425-426
    assert np.all(np.diff(y) > 0)
    return y


def _axes(n, npts, jitter_rng=None):
    xs = [linspace_ref(-5.0 * i, 5.0 * (i + 1), npts) for i in range(n)]
    if jitter_rng is not None:
        xs = [_jitter(jitter_rng, x) for x in xs]
    return xs


def _obs_mesh(n, npts, lo=-7.0, hi=7.0):
    xobs = [linspace_ref(lo * i, hi * (i + 1), npts) for i in range(n)]
    pts = meshgrid_ref(xobs)
    return [np.ascontiguousarray(pts[:, j]) for j in range(n)], pts


def linear_sum_cases(kind: str, max_n: int = 8):
    """multilinear/regular.rs:438-477, regular_recursive.rs:402-441 (N=1..8);
    rectilinear.rs:414-456, rectilinear_recursive.rs:380-423."""
    rng = np.random.default_rng(20260103)
    out = []
    for n in range(1, max_n + 1):
        xs = _axes(n, 2, rng if kind == "rectilinear" else None)
        u = seq_sum(meshgrid_ref(xs))
        obs, pts = _obs_mesh(n, 3)
        ref = "src/multilinear/regular.rs:438" if kind == "regular" else "src/multilinear/rectilinear.rs:414"
        out.append(Case(f"lin_{kind}_sum_N{n}", "linear", kind, xs, u, obs, seq_sum(pts), 1e-12, ref=ref))
    return out


def hat_cases():
    """multilinear/regular.rs:481-495 and rectilinear.rs:460-476 — exact equality."""
    y = np.array([0.0, 1.0, 0.0])
    x = np.array([0.0, 1.0, 2.0])
    obs = linspace_ref(-2.0, 4.0, 100)
    exp = np.where(obs <= 1.0, obs, 2.0 - obs)
    return [
        Case("lin_regular_hat", "linear", "regular", [x], y, [obs], exp, 0.0, ref="src/multilinear/regular.rs:481"),
        Case("lin_rectilinear_hat", "linear", "rectilinear", [x], y, [obs], exp, 0.0,
             ref="src/multilinear/rectilinear.rs:460"),
    ]


def rect_2d_small_case():
    """multilinear/rectilinear.rs:381-407."""
    x = linspace_ref(-1.0, 1.0, 3)
    y = np.array([0.5, 0.6])
    xy = meshgrid_ref([x, y])
    z = xy[:, 0] + xy[:, 1]
    xo = linspace_ref(-10.0, 10.0, 5)
    pts = meshgrid_ref([xo, xo])
    return Case("lin_rectilinear_2d_small", "linear", "rectilinear", [x, y], z,
                [np.ascontiguousarray(pts[:, 0]), np.ascontiguousarray(pts[:, 1])], pts[:, 0] + pts[:, 1], 1e-12,
                ref="src/multilinear/rectilinear.rs:381")


def cubic_poly_cases(kind: str, max_n: int = 5):
    """multicubic/regular.rs:635-730 (+ regular_recursive.rs:622-716: `1..6` = N<=5) and
    multicubic/rectilinear.rs:558-667 (+ rectilinear_recursive.rs:552-660: linear `1..=6`,
    quadratic `1..6`).  Linear field with both `linearize` flags; quadratic field incl.
    extrapolation, linearize=false.  The reference asserts nothing for N>6 (nor for N=6 on
    the regular grid), so no case is generated there."""
    rng = np.random.default_rng(20260104)
    out = []
    for n in range(1, max_n + 1):
        xs = _axes(n, 4, rng if kind == "rectilinear" else None)
        g = meshgrid_ref(xs)
        obs, pts = _obs_mesh(n, 6)
        tol_lin = 1e-12 if kind == "regular" else 1e-10
        ref = f"src/multicubic/{kind}.rs:" + ("635" if kind == "regular" else "558")
        if n <= (5 if kind == "regular" else 6):
            for lin in (False, True):
                out.append(Case(f"cub_{kind}_linear_N{n}_lin{int(lin)}", "cubic", kind, xs, seq_sum(g), obs,
                                seq_sum(pts), tol_lin, linearize=lin, ref=ref))
        if n <= 5:
            out.append(Case(f"cub_{kind}_quadratic_N{n}", "cubic", kind, xs, seq_sum_sq(g), obs, seq_sum_sq(pts),
                            1e-10, linearize=False, ref=ref.replace("635", "681").replace("558", "609")))
    return out


def cubic_sine_cases(kind: str):
    """multicubic/regular.rs:737-792, rectilinear.rs:674-736 — N=1,2, tolerance 2e-2*N."""
    rng = np.random.default_rng(20260105)
    out = []
    for n in (1, 2):
        xs = _axes(n, 10, rng if kind == "rectilinear" else None)
        g = meshgrid_ref(xs)
        npts = 12 if kind == "regular" else 11
        xobs = [linspace_ref(-5.0 * i, 5.0 * (i + 1), npts) for i in range(n)]
        pts = meshgrid_ref(xobs)
        obs = [np.ascontiguousarray(pts[:, j]) for j in range(n)]
        out.append(Case(f"cub_{kind}_sine_N{n}", "cubic", kind, xs, seq_sum_sin(g), obs, seq_sum_sin(pts),
                        2e-2 * n, linearize=False,
                        ref=f"src/multicubic/{kind}.rs:" + ("737" if kind == "regular" else "674")))
    return out


def const2_cases():
    """Doctests: src/lib.rs:24-80, multilinear/regular.rs:3-28, multicubic/regular.rs:3-32.
    A constant field must come back exactly, including at extrapolated points."""
    out = []
    xl, yl = np.array([1.0, 2.0]), np.array([1.0, 1.5])
    obs = [np.array([0.0, 5.0]), np.array([-1.0, 3.0])]
    for kind in ("regular", "rectilinear"):
        out.append(Case(f"const2_lin_{kind}", "linear", kind, [xl, yl], np.full(4, 2.0), obs, np.full(2, 2.0), 0.0,
                        ref="src/multilinear/regular.rs:3-28"))
    xc, yc = np.array([1.0, 2.0, 3.0, 4.0]), np.array([1.0, 1.5, 2.0, 2.5])
    for kind in ("regular", "rectilinear"):
        for lin in (False, True):
            out.append(Case(f"const2_cub_{kind}_lin{int(lin)}", "cubic", kind, [xc, yc], np.full(16, 2.0), obs,
                            np.full(2, 2.0), 0.0, linearize=lin, ref="src/multicubic/regular.rs:3-32"))
    return out


def py_on_grid_cases():
    """test/test_multilinear_regular.py:5-93, test_multilinear_rectilinear.py:5-84,
    test_multicubic_regular.py:5-100, test_multicubic_rectilinear.py:5-85 — z = x + 2y sampled at
    every grid node, np.linspace axes, f64 and f32."""
    out = []
    for dtype in (np.float64, np.float32):
        tag = "f64" if dtype == np.float64 else "f32"

        def mk(name, method, kind, x, y, atol, rel=None):
            xg, yg = np.meshgrid(x, y, indexing="ij")
            z = (xg + 2.0 * yg).astype(dtype)
            obs = [xg.flatten().astype(dtype), yg.flatten().astype(dtype)]
            c = Case(f"py_{name}_{tag}", method, kind, [x, y], z.flatten(), obs, z.flatten(), atol, linearize=False,
                     ref=f"test/test_{'multilinear' if method == 'linear' else 'multicubic'}_{kind}.py")
            if rel is not None:
                c.extra["rel"] = rel
            return c

        x = np.linspace(0.0, 10.0, 5).astype(dtype)
        y = np.linspace(20.0, 30.0, 3).astype(dtype)
        out.append(mk("lin_regular", "linear", "regular", x, y, 0.0))
        out.append(mk("lin_rectilinear", "linear", "rectilinear", x, y, 0.0))
        x = np.linspace(0.0, 10.0, 7).astype(dtype)
        y = np.linspace(20.0, 30.0, 5).astype(dtype)
        # test_multicubic_regular.py:6,97-100: rel 1e-12 (f64) / 1e-6 (f32), normalised by max(|ref|, 1)
        out.append(mk("cub_regular", "cubic", "regular", x, y, -1.0, rel=1e-12 if dtype == np.float64 else 1e-6))
        x = np.linspace(0.0, 10.0, 5).astype(dtype)
        y = np.linspace(20.0, 30.0, 4).astype(dtype)
        out.append(mk("cub_rectilinear", "cubic", "rectilinear", x, y, 0.0))
    return out


def _nearest_regular_index(value, start, step, size):
    """The reference tests' own helper: src/nearest/regular.rs:324-338 (unfused arithmetic) and
    test/test_nearest_regular.py:5-10."""
    floc = np.floor((value - start) / step)
    origin = int(max(0, min(floc, size - 2)))
    dt = (value - (start + step * origin)) / step
    return origin if dt <= 0.5 else min(origin + 1, size - 1)


def _nearest_rect_index(value, grid):
    """src/nearest/rectilinear.rs:274-283."""
    iloc = int(np.sum(grid < value)) - 1  # partition_point(|x| x < value) - 1 on a sorted grid
    origin = int(max(0, min(iloc, grid.size - 2)))
    dt = (value - grid[origin]) / (grid[origin + 1] - grid[origin])
    return origin if dt <= 0.5 else origin + 1


def nearest_cases():
    """src/nearest/regular.rs:344-417, rectilinear.rs:285-390 and test/test_nearest_*.py."""
    out = []
    rng = np.random.default_rng(20260106)
    for kind in ("regular", "rectilinear"):
        for n in range(1, 7):
            xs = _axes(n, 2, rng if kind == "rectilinear" else None)
            u = seq_sum(meshgrid_ref(xs))
            obs, pts = _obs_mesh(n, 3)
            exp = np.zeros(pts.shape[0])
            for d in range(n):
                if kind == "regular":
                    start, step = xs[d][0], xs[d][1] - xs[d][0]
                    idx = [_nearest_regular_index(v, start, step, 2) for v in pts[:, d]]
                    exp = exp + np.array([start + step * i for i in idx])
                else:
                    idx = [_nearest_rect_index(v, xs[d]) for v in pts[:, d]]
                    exp = exp + xs[d][idx]
            out.append(Case(f"near_{kind}_sum_N{n}", "nearest", kind, xs, u, obs, exp, 1e-12,
                            ref=f"src/nearest/{kind}.rs"))
        # hat function, exact equality
        y = np.array([0.0, 1.0, 0.0])
        x = np.array([0.0, 1.0, 2.0])
        o = linspace_ref(-2.0, 4.0, 100)
        if kind == "regular":
            exp = np.array([y[_nearest_regular_index(v, 0.0, 1.0, 3)] for v in o])
        else:
            exp = np.array([y[_nearest_rect_index(v, x)] for v in o])
        out.append(Case(f"near_{kind}_hat", "nearest", kind, [x], y, [o], exp, 0.0, ref=f"src/nearest/{kind}.rs"))
    # rectilinear 2d small (rectilinear.rs:285-313)
    x = linspace_ref(-1.0, 1.0, 3)
    y = np.array([0.5, 0.6])
    xy = meshgrid_ref([x, y])
    z = xy[:, 0] + xy[:, 1]
    xo = linspace_ref(-10.0, 10.0, 5)
    pts = meshgrid_ref([xo, xo])
    exp = np.array([x[_nearest_rect_index(p[0], x)] + y[_nearest_rect_index(p[1], y)] for p in pts])
    out.append(Case("near_rectilinear_2d_small", "nearest", "rectilinear", [x, y], z,
                    [np.ascontiguousarray(pts[:, 0]), np.ascontiguousarray(pts[:, 1])], exp, 1e-12,
                    ref="src/nearest/rectilinear.rs:285"))
    # Python tests (exact equality, f64 and f32)
    for dtype in (np.float64, np.float32):
        tag = "f64" if dtype == np.float64 else "f32"
        x = np.linspace(0.0, 6.0, 4).astype(dtype)
        y = np.linspace(-3.0, 3.0, 3).astype(dtype)
        xg, yg = np.meshgrid(x, y, indexing="ij")
        zg = (xg - 2.0 * yg).astype(dtype)
        obs = [np.array([0.1, 1.6, 2.9, 5.0], dtype=dtype), np.array([-3.0, -1.2, 0.4, 2.4], dtype=dtype)]
        starts = np.array([x[0], y[0]]).astype(dtype)
        steps = np.array([x[1] - x[0], y[1] - y[0]]).astype(dtype)
        exp = np.array([zg[_nearest_regular_index(float(a), float(starts[0]), float(steps[0]), 4),
                           _nearest_regular_index(float(b), float(starts[1]), float(steps[1]), 3)]
                        for a, b in zip(obs[0], obs[1])], dtype=dtype)
        out.append(Case(f"py_near_regular_{tag}", "nearest", "regular", [x, y], zg.flatten(), obs, exp, 0.0,
                        ref="test/test_nearest_regular.py"))
        x = np.array([0.0, 1.0, 3.5, 4.0], dtype=dtype)
        y = np.array([-2.0, -0.5, 0.1], dtype=dtype)
        xg, yg = np.meshgrid(x, y, indexing="ij")
        zg = (xg + yg**2).astype(dtype)
        obs = [np.array([0.2, 2.8, 3.8], dtype=dtype), np.array([-1.5, -0.2, 0.4], dtype=dtype)]
        exp = np.array([zg[_nearest_rect_index(float(a), x), _nearest_rect_index(float(b), y)]
                        for a, b in zip(obs[0], obs[1])], dtype=dtype)
        out.append(Case(f"py_near_rectilinear_{tag}", "nearest", "rectilinear", [x, y], zg.flatten(), obs, exp, 0.0,
                        ref="test/test_nearest_rectilinear.py"))
    return out


def all_cases(max_lin_n: int = 8, max_cub_n: int = 5):
    cases = []
    for kind in ("regular", "rectilinear"):
        cases += linear_sum_cases(kind, max_lin_n)
        cases += cubic_poly_cases(kind, max_cub_n)
        cases += cubic_sine_cases(kind)
    cases += hat_cases()
    cases.append(rect_2d_small_case())
    cases += const2_cases()
    cases += py_on_grid_cases()
    cases += nearest_cases()
    return cases


def check(case: Case, out: np.ndarray):
    exp = case.expected
    if "rel" in case.extra:
        err = np.abs(out.astype(np.float64) - exp.astype(np.float64)) / np.maximum(np.abs(exp.astype(np.float64)), 1.0)
        assert np.all(err < case.extra["rel"]), (case.name, float(err.max()))
    elif case.atol == 0.0:
        assert np.array_equal(out, exp), (case.name, out, exp)
    else:
        err = np.abs(out - exp)
        assert np.all(err < case.atol), (case.name, float(err.max()))
