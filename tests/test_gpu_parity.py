"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's
known answers.  Run with `pytest -m gpu` on an MI355X.

Tolerances (north_star): <= 1e-12 relative for linear, <= 1e-10 for cubic in f64, normalised by
max(|ref|, 1); f32: 1e-6 (test/test_multicubic_regular.py:6).  The kernels reproduce the
reference's operation order and FMA sites, so the *expected* difference to the oracle is zero
ulp; the tests assert bit equality and would report the tolerance margin if that ever broke.
"""

import numpy as np
import pytest

from tests import kat
from tests.helpers import rel_err, run_hip_raw, run_oracle, synthetic_case

pytestmark = pytest.mark.gpu

TOL = {("nearest", np.float64): 0.0, ("nearest", np.float32): 0.0, ("linear", np.float64): 1e-12, ("cubic", np.float64): 1e-10, ("linear", np.float32): 1e-6,
       ("cubic", np.float32): 1e-6}


def _ndev():
    """GPUs this process can use, as the C ABI counts them (1 on the pool's test boxes, 8 on a node):
    every multi-handle test spreads its handles over them (interpn_amd.sharded.device_for_shard)."""
    import interpn_amd

    return max(1, interpn_amd._lib.load().interpn_hip_device_count())


def assert_parity(case, got, want):
    dtype = np.dtype(got.dtype).type
    err = rel_err(got, want)
    finite = np.isfinite(want)
    assert np.array_equal(np.isnan(got), np.isnan(want)), case.name
    tol = TOL[(case.method, dtype)]
    assert np.all(err[finite] <= tol), (case.name, float(err[finite].max()))
    # stronger: bit-identical
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    assert np.all(same), (case.name, int((~same).sum()), float(err[finite].max()))


ALL_KATS = kat.all_cases(8, 6)


@pytest.mark.parametrize("case", ALL_KATS, ids=lambda c: c.name)
def test_known_answers(oracle, case):
    """Every hot-path known-answer test of the reference, through the raw (one-shot) ABI."""
    got = run_hip_raw(case)
    kat.check(case, got)
    assert_parity(case, got, run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
def test_linear_random(oracle, kind, n, dtype):
    axis = {1: [257], 2: [33, 64], 3: [17, 9, 32], 4: [7, 9, 5, 12], 5: [4, 5, 3, 6, 7], 6: [3, 4, 2, 5, 3, 4]}[n]
    case = synthetic_case("linear", kind, n, axis, 200_003, 100 + n, dtype)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("linearize", [False, True], ids=["quad", "lin"])
@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
def test_cubic_random(oracle, kind, n, linearize, dtype):
    axis = {1: [129], 2: [17, 32], 3: [9, 6, 16], 4: [5, 7, 4, 9]}[n]
    case = synthetic_case("cubic", kind, n, axis, 100_003, 200 + n, dtype, linearize=linearize, extrap=0.2)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("form", ["static", "runtime"])
@pytest.mark.parametrize("linearize", [False, True], ids=["quad", "lin"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("method,n", [("linear", 7), ("linear", 8), ("cubic", 5), ("cubic", 6), ("cubic", 7)])
def test_recursive_arms(oracle, monkeypatch, method, kind, n, linearize, form, dtype):
    """N = 7,8 (linear) and 5..8 (cubic) take the reference's recursive arm, with its own FMA
    sites (regular_recursive.rs:310-313; rectilinear_recursive.rs:467,527).  Served by the
    compile-time-N vertex loop (k_generic_n); the runtime-N kernel must agree bit for bit."""
    if method == "linear" and linearize:
        pytest.skip("linearize_extrapolation is a cubic argument")
    if form == "runtime":
        monkeypatch.setenv("INTERPN_HIP_GENERIC_RUNTIME", "1")
    m = 2 if method == "linear" else 4
    axis = [m + (d % 2) for d in range(n)]
    nobs = 3001
    case = synthetic_case(method, kind, n, axis, nobs, 300 + n, dtype, linearize=linearize, extrap=0.3)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("vec", ["0", "1"], ids=["onetree", "rowvec"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("method,n", [("linear", 7), ("linear", 8), ("cubic", 5), ("cubic", 6), ("cubic", 7), ("cubic", 8)])
def test_recursive_arm_forms(oracle, monkeypatch, method, kind, n, vec, dtype):
    """Both forms of k_generic_n (one tree per leaf element / FP trees side by side over one row
    load) for every shape of the recursive arms.  The row-vector form is compiled only where it
    fits the register file (k_generic.hip::generic_vec_ok, asserted on the build by
    tests/test_build_resources.py); asking for it elsewhere must fall back to the one-tree form,
    never to a spilled kernel (round 1: f64 cubic regular N = 8 returned wrong results)."""
    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_GENERIC_VEC", vec)
    m = 2 if method == "linear" else 4
    axis = [m + ((d + 1) % 2) for d in range(n)] if n < 8 or method == "linear" else [4] * 8
    nobs = 1501 if n < 8 or method == "linear" else 301
    case = synthetic_case(method, kind, n, axis, nobs, 900 + n, dtype, linearize=True, extrap=0.3)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))
    # which instantiation ran: query it from a handle built the same way
    cv = lambda a: np.ascontiguousarray(a, dtype=dtype)
    if kind == "regular":
        it = interpn_amd.Interpolator.regular(method, case.dims, cv(case.starts), cv(case.steps), cv(case.vals), True)
    else:
        it = interpn_amd.Interpolator.rectilinear(method, [cv(g) for g in case.grids], cv(case.vals), True)
    out = it.eval_host([cv(o) for o in case.obs], np.zeros(nobs, dtype=dtype))
    name = it.kernel_name()
    it.close()
    assert_parity(case, out, run_oracle(oracle, case, True))
    assert name.startswith("interpn::k_generic_n<"), name
    f64_cubic = method == "cubic" and dtype == np.float64
    spilled = f64_cubic and ((kind == "regular" and n >= 7) or (kind == "rectilinear" and n >= 6))
    want_vec = vec == "1" and not spilled
    assert name.endswith(", true>") == want_vec, (name, vec)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("linearize", [False, True], ids=["quad", "lin"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
def test_recursive_arm_cubic_8d(oracle, kind, linearize, dtype):
    """The largest shape the reference dispatches (multicubic N = 8: 65 536 grid reads per point);
    30 % of the coordinates extrapolate, so every saturation case meets every level."""
    case = synthetic_case("cubic", kind, 8, [4] * 8, 601, 77, dtype, linearize=linearize, extrap=0.3,
                          specials=False)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("layout", ["off", "11", "12", "22", "c4", "j4"])
@pytest.mark.parametrize("axis", [[2, 2, 2], [3, 4, 5], [17, 9, 32], [64, 64, 64], [8, 7, 3], [5, 9, 8, 17],
                                  [2, 3, 2, 2, 9], [3, 2, 4, 3, 5, 4]], ids=str)
def test_linear_brick_layouts(oracle, monkeypatch, dtype, kind, layout, axis):
    """Multilinear N = 3..6 runs on a bricked copy of the grid with a quad-cooperative gather
    (k_linear_brick.hip); every brick overlap scheme and the C-order kernel must give the same
    bits, including 2-point axes, odd sizes, leading dimensions and NaN / out-of-range
    coordinates.  `j4` = the f32-only 2 x 4 x 4 bricks (ignored for f64)."""
    monkeypatch.setenv("INTERPN_HIP_BRICKS", layout)
    n = len(axis)
    case = synthetic_case("linear", kind, n, axis, 40_001, 900 + sum(axis), dtype, extrap=0.3,
                          specials=min(axis) >= 8)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))
    if kind == "regular" and dtype == np.float64:
        case.obs[n - 1][4321] = np.nan
        from interpn_amd import raw

        got = np.full(40_001, -1.0)
        with pytest.raises(AssertionError, match="Unrepresentable coordinate value"):
            raw.interpn_linear_regular_f64(case.dims, case.starts, case.steps, case.vals, case.obs, got)
        assert np.all(got[4321:] == -1.0) and np.all(got[:4321] != -1.0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("axis_regs", ["0", "1", "2"])
@pytest.mark.parametrize("ppl", ["1", "2"])
@pytest.mark.parametrize("axis", [[2, 2, 2], [64, 64, 64], [63, 2, 33], [5, 64, 7], [65, 8, 8]], ids=str)
def test_rectilinear_axes_in_registers(oracle, monkeypatch, dtype, axis_regs, ppl, axis):
    """3-D rectilinear axes of at most 64 coordinates are searched across lanes (one coordinate
    per lane, ds_bpermute probes) instead of in LDS — mode 1 with the reference's probe sequence,
    mode 2 through a 255-bucket lane table; same bits as the LDS form (mode 0), including NaN /
    +-inf coordinates, 2-point axes and the 65-point case that must fall back to LDS."""
    monkeypatch.setenv("INTERPN_HIP_AXIS_REGS", axis_regs)
    monkeypatch.setenv("INTERPN_HIP_PPL", ppl)
    case = synthetic_case("linear", "rectilinear", 3, axis, 30_011, 4000 + sum(axis), dtype, extrap=0.3,
                          specials=min(axis) >= 8)
    case.obs[0][17] = np.nan
    case.obs[1][18] = np.inf
    case.obs[2][19] = -np.inf
    case.obs[0][20] = 1e300 if dtype == np.float64 else 1e30
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("axis_regs", ["0", "1", "2"])
@pytest.mark.parametrize("method,axis", [("linear", [64, 61]), ("linear", [2, 64]), ("linear", [65, 64]), ("linear", [9, 8, 7, 6]),
                                         ("linear", [2, 64, 3, 5]), ("linear", [4, 3, 5, 6, 7]), ("linear", [3, 4, 2, 5, 3, 4]),
                                         ("nearest", [64]), ("nearest", [33, 64]), ("nearest", [64, 2, 17]),
                                         ("nearest", [5, 6, 7, 8]), ("nearest", [3, 4, 5, 2, 6, 3]), ("nearest", [65, 9])],
                         ids=str)
def test_lane_resident_axes_everywhere(oracle, monkeypatch, dtype, axis_regs, method, axis):
    """Round 2: the cross-lane axis search (lane_axes.h) also serves the 2-D brick kernel, the
    4-D..6-D brick kernels (both brick layouts) and the nearest kernel whenever every axis has at
    most 64 coordinates; modes 0 (LDS), 1 (probe sequence) and 2 (lane table) must agree bit for
    bit, dead lanes of partial waves included (batch sizes that are not multiples of 64), and a
    65-point axis must fall back to LDS."""
    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_AXIS_REGS", axis_regs)
    n = len(axis)
    for nobs in (40_001, 63, 64, 65, 1):
        case = synthetic_case(method, "rectilinear", n, axis, nobs, 4100 + sum(axis) + nobs, dtype, extrap=0.3,
                              specials=min(axis) >= 8)
        if nobs > 100:
            case.obs[0][77] = np.nan
            case.obs[n - 1][78] = np.inf
            case.obs[0][79] = -np.inf
        assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))
    cv = lambda a: np.ascontiguousarray(a, dtype=dtype)
    it = interpn_amd.Interpolator.rectilinear(method, [cv(g) for g in case.grids], cv(case.vals))
    it.eval_host([cv(o) for o in case.obs], np.zeros(1, dtype=dtype))
    name = it.kernel_name()
    it.close()
    in_lanes = max(axis) <= 64 and axis_regs != "0"
    want = {"0": (0,), "1": (1,), "2": (2, 3)}[axis_regs] if in_lanes else (0,)  # "2" = lane tables: mode 2 or 3
    args = [x.strip() for x in name[name.index("<") + 1:-1].split(",")]
    if name.startswith("interpn::k_nearest<") or name.startswith("interpn::k_linear2_brick<"):
        assert int(args[-2]) in want, name  # ..., AXR, PPL>
    elif name.startswith("interpn::k_linear_brick<"):
        assert int(args[-3]) in want, name  # ..., PPL, AXR, ABL, CELL>


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("axis_regs", ["0", "1", "2"])
def test_rectilinear_clustered_and_unsorted_axes(oracle, monkeypatch, dtype, axis_regs):
    """Lane-table corner cases: many coordinates inside one bucket (the scan length grows to the
    bucket population), coordinates a few ulps apart, and an unsorted axis (legal input: the
    reference only checks g[1] > g[0]) which has no table and takes the probe-sequence search."""
    from tests.kat import Case

    monkeypatch.setenv("INTERPN_HIP_AXIS_REGS", axis_regs)
    rng = np.random.default_rng(99)
    one = dtype(1.0)
    g0 = np.concatenate([[-1.0], 0.25 + np.arange(20) * 1e-4, [0.9, 1.0]]).astype(dtype)      # 20 nodes in one bucket
    g1 = np.concatenate([[-1.0, -0.5], [np.nextafter(one, dtype(2)) * dtype(0.125) * k for k in range(1, 4)],
                         [2.0]]).astype(dtype)
    g1 = np.sort(np.unique(g1))
    g2 = np.array([0.0, 1.0, 0.5, 3.0, 2.0, 2.5, 4.0], dtype=dtype)                          # unsorted
    for grids in ([g0, g1, np.linspace(-1, 1, 64).astype(dtype)], [g0, g2, g1]):
        nobs = 20_000
        vals = rng.uniform(-1, 1, int(np.prod([g.size for g in grids]))).astype(dtype)
        obs = [rng.uniform(float(g.min()) - 0.2, float(g.max()) + 0.2, nobs).astype(dtype) for g in grids]
        obs[0][:g0.size] = g0          # exact nodes of the clustered axis
        obs[0][100:120] = g0[1:21] + dtype(5e-5)
        case = Case("clustered", "linear", "rectilinear", grids, vals, obs, np.zeros(nobs, dtype=dtype), 0.0)
        assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("layout", ["off", "on"])
@pytest.mark.parametrize("axis", [[2, 2], [2, 9], [3, 8], [9, 2], [16, 17], [33, 15], [7, 100], [257, 129]], ids=str)
def test_linear2_brick(oracle, monkeypatch, dtype, kind, layout, axis):
    """2-D multilinear on 2 x KW2 bricks with a lane-pair gather (k_linear2_brick.hip): same bits as
    the C-order kernel; axis lengths around the brick step (7 / 15 columns), odd point counts, NaN."""
    if layout == "off":
        monkeypatch.setenv("INTERPN_HIP_BRICKS", "off")
    else:
        monkeypatch.delenv("INTERPN_HIP_BRICKS", raising=False)
    case = synthetic_case("linear", kind, 2, axis, 30_001, 1700 + sum(axis), dtype, extrap=0.3, specials=min(axis) >= 8)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))
    if kind == "regular" and dtype == np.float64:
        case.obs[0][2999] = np.inf
        from interpn_amd import raw

        got = np.full(30_001, -1.0)
        with pytest.raises(AssertionError, match="Unrepresentable coordinate value"):
            raw.interpn_linear_regular_f64(case.dims, case.starts, case.steps, case.vals, case.obs, got)
        assert np.all(got[2999:] == -1.0) and np.all(got[:2999] != -1.0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("linearize", [False, True], ids=["quad", "lin"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("layout", ["off", "44", "24", "22", "14", "11"])
@pytest.mark.parametrize("axis", [[4, 4], [9, 6], [4, 5, 4], [13, 7, 6], [5, 4, 6, 7], [8, 9, 4, 5]], ids=str)
def test_cubic_tile_layouts(oracle, monkeypatch, dtype, linearize, kind, layout, axis):
    """Multicubic N = 2..4 runs on a tiled copy of the grid (dims 0,1 in 4 x 4 tiles) with a 16-lane
    cooperative gather (k_cubic_brick.hip); every tile overlap scheme and the C-order kernel must
    give the same bits, including 4-point axes (single tile) and heavy extrapolation."""
    monkeypatch.setenv("INTERPN_HIP_BRICKS", layout)
    n = len(axis)
    case = synthetic_case("cubic", kind, n, axis, 20_003, 1300 + sum(axis), dtype, linearize=linearize, extrap=0.3,
                          specials=min(axis) >= 8)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("method", ["linear", "cubic"])
def test_generic_kernel_at_low_n(oracle, monkeypatch, method, kind, n, dtype):
    """The runtime-N kernel (normally N >= 7 linear / N >= 5 cubic) forced onto small N, where the
    templated kernels and the oracle's flattened arm give the answer: for N <= 4 no FMA site
    differs between the arms except the two the kernel takes as flags, which follow N."""
    monkeypatch.setenv("INTERPN_HIP_FORCE_GENERIC", "1")
    m = 5 if method == "linear" else 6
    case = synthetic_case(method, kind, n, [m + d for d in range(n)], 20_011, 500 + n, dtype, linearize=True,
                          extrap=0.3)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
def test_nearest_random(oracle, kind, n, dtype):
    """nearest::{regular,rectilinear}: index stage of the multilinear path +
    one gather; must pick the same node as the reference, ties (dt == 0.5) and NaN included."""
    axis = {1: [257], 2: [33, 64], 3: [17, 9, 32], 4: [7, 9, 5, 12], 5: [4, 5, 3, 6, 7], 6: [3, 4, 2, 5, 3, 4]}[n]
    case = synthetic_case("nearest", kind, n, axis, 100_003, 1500 + n, dtype, extrap=0.2)
    # exact midpoints between nodes (dt == 0.5 -> lower node)
    for d in range(n):
        g = case.grids[d]
        case.obs[d][100:100 + g.size - 1] = ((g[:-1].astype(np.float64) + g[1:].astype(np.float64)) / 2).astype(dtype)
    if kind == "rectilinear":
        case.obs[0][77] = np.nan  # NaN -> cell 0, dt NaN -> upper node, no error
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


def test_nearest_api_levels(oracle):
    """test/test_nearest_regular.py and test_nearest_rectilinear.py: raw, interpn(method="nearest"),
    class .eval and JSON round trip agree exactly; 7-D input is rejected with the reference's message."""
    import interpn_amd

    for c in kat.nearest_cases():
        if not c.name.startswith("py_"):
            continue
        want = run_oracle(oracle, c, True)
        kat.check(c, want)
        if c.kind == "regular":
            it = interpn_amd.NearestRegular.new(c.dims, c.starts, c.steps, c.vals)
        else:
            it = interpn_amd.NearestRectilinear.new(c.grids, c.vals)
        assert np.array_equal(it.eval(c.obs), want)
        assert np.array_equal(type(it).model_validate_json(it.model_dump_json()).eval(c.obs), want)
        assert np.array_equal(interpn_amd.interpn(obs=c.obs, grids=c.grids, vals=c.vals, method="nearest"), want)
    with pytest.raises(AssertionError, match=r"Dimension exceeds maximum \(6\)\."):
        interpn_amd.raw.interpn_nearest_regular_f64([2] * 7, np.zeros(7), np.ones(7), np.zeros(128),
                                                    [np.zeros(3)] * 7, np.zeros(3))


def test_no_fma_flavour(oracle):
    """interpn_hip_set_fma(0) == the reference built without the `fma` feature."""
    from interpn_amd import _lib

    lib = _lib.load()
    prev = lib.interpn_hip_set_fma(0)
    try:
        for method, kind, n, axis in [("linear", "regular", 3, [9, 8, 7]), ("linear", "rectilinear", 2, [11, 6]),
                                      ("cubic", "regular", 2, [8, 9]), ("cubic", "rectilinear", 3, [5, 6, 7]),
                                      ("linear", "regular", 7, [2, 3, 2, 3, 2, 3, 2])]:
            case = synthetic_case(method, kind, n, axis, 50_001, 400 + n, np.float64, linearize=True, extrap=0.2)
            assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, False))
    finally:
        lib.interpn_hip_set_fma(prev)


@pytest.mark.parametrize("method", ["linear", "cubic"])
@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf, 1e300])
def test_unrepresentable_aborts_at_first_bad_point(oracle, method, bad):
    """multilinear/regular.rs:418 + :277-280: the batch stops at the first failing point; the
    prefix is written, the rest of `out` untouched.  (1e300/step overflows isize.)"""
    case = synthetic_case(method, "regular", 2, [8, 9], 10_000, 5, np.float64, specials=False)
    k = 7777
    case.obs[1][k] = bad
    case.obs[0][k + 50] = np.nan  # a later failure must not win
    want = np.full(10_000, -123.0)
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
        if method == "linear":
            oracle.linear_regular(case.dims, case.starts, case.steps, case.vals, case.obs, want)
        else:
            oracle.cubic_regular(case.dims, case.starts, case.steps, case.vals, False, case.obs, want)
    assert ei.value.first_bad == k
    from interpn_amd import raw

    got = np.full(10_000, -123.0)
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value"):
        if method == "linear":
            raw.interpn_linear_regular_f64(case.dims, case.starts, case.steps, case.vals, case.obs, got)
        else:
            raw.interpn_cubic_regular_f64(case.dims, case.starts, case.steps, case.vals, False, case.obs, got)
    assert np.array_equal(got, want)
    assert np.all(got[k:] == -123.0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("nobs,k", [(1, 0), (1000, 0), (1000, 999), (5000, 4321), (8192, 8191), (8193, 8192), (8193, 10)])
def test_small_batch_zero_copy_path(oracle, dtype, nobs, k):
    """Host batches of at most 8192 points take the zero-copy path (pinned staging the kernel reads
    and writes over PCIe, one synchronisation; abi_host.hip::eval_host_small); 8193 points take
    the staged pipeline.  Same bits, and the same abort contract at the first failing point —
    prefix written, the rest of the caller's `out` untouched — on a resident handle used twice
    (the sticky status word must be clean again) and through the one-shot entry point."""
    import interpn_amd
    from interpn_amd import raw

    case = synthetic_case("linear", "regular", 3, [20, 21, 22], nobs, 600 + nobs + k, dtype, specials=False)
    cv = lambda a: np.ascontiguousarray(a, dtype=dtype)
    obs = [cv(o) for o in case.obs]
    want = run_oracle(oracle, case, True)
    it = interpn_amd.Interpolator.regular("linear", case.dims, cv(case.starts), cv(case.steps), cv(case.vals))
    assert np.array_equal(it.eval_host(obs, np.zeros(nobs, dtype=dtype)), want)
    bad = [o.copy() for o in obs]
    bad[1][k] = np.nan
    if k + 3 < nobs:
        bad[0][k + 3] = np.inf  # a later failure must not win
    got = np.full(nobs, -5.0, dtype=dtype)
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value"):
        it.eval_host(bad, got)
    assert np.array_equal(got[:k], want[:k]) and np.all(got[k:] == -5.0)
    assert np.array_equal(it.eval_host(obs, np.zeros(nobs, dtype=dtype)), want)  # status word reset
    it.close()
    got = np.full(nobs, -5.0, dtype=dtype)
    fn = raw.interpn_linear_regular_f64 if dtype == np.float64 else raw.interpn_linear_regular_f32
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value"):
        fn(case.dims, cv(case.starts), cv(case.steps), cv(case.vals), bad, got)
    assert np.array_equal(got[:k], want[:k]) and np.all(got[k:] == -5.0)


@pytest.mark.parametrize("method", ["linear", "cubic"])
def test_rectilinear_never_errors_and_propagates_nan(oracle, method):
    """multilinear/rectilinear.rs:353-370: NaN lands in cell 0 and the result is NaN; +-inf
    extrapolate to +-inf/NaN; no error is raised."""
    case = synthetic_case(method, "rectilinear", 2, [8, 9], 4096, 6, np.float64, specials=False)
    case.obs[0][10] = np.nan
    case.obs[1][11] = np.inf
    case.obs[0][12] = -np.inf
    case.obs[1][13] = 1e300
    want = run_oracle(oracle, case, True)
    got = run_hip_raw(case)
    assert np.isnan(got[10])
    assert_parity(case, got, want)


def test_eval_host_multi_chunk(oracle):
    """More points than one host-pipeline chunk (4 Mi): chunk seams must be invisible."""
    case = synthetic_case("linear", "regular", 3, [16, 16, 16], (4 << 20) + 12345, 9, np.float64)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("chunk", [1000, 777, 4999, 5000])
@pytest.mark.parametrize("k", [0, 999, 1000, 4321, 9999])
def test_host_pipeline_abort_semantics(oracle, monkeypatch, chunk, k):
    """The two-lane host pipeline (upload of chunk c+1 overlapping the download of chunk c) keeps
    the reference's contract at every chunk seam: out[0..k) written, out[k..] untouched, also when
    later chunks (of either lane) fail as well."""
    from interpn_amd import raw

    monkeypatch.setenv("INTERPN_HIP_HOST_CHUNK", str(chunk))
    case = synthetic_case("linear", "regular", 2, [8, 9], 10_000, 5, np.float64, specials=False)
    case.obs[1][k] = np.nan
    for later in (k + 1, k + chunk, k + 2 * chunk + 3, 9_998):
        if k < later < 10_000:
            case.obs[0][later] = np.inf
    want = np.full(10_000, -123.0)
    with pytest.raises(AssertionError) as eo:
        oracle.linear_regular(case.dims, case.starts, case.steps, case.vals, case.obs, want)
    assert eo.value.first_bad == k
    got = np.full(10_000, -123.0)
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value"):
        raw.interpn_linear_regular_f64(case.dims, case.starts, case.steps, case.vals, case.obs, got)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("chunk", [1, 333, 4096])
def test_host_pipeline_many_chunks(oracle, monkeypatch, chunk):
    monkeypatch.setenv("INTERPN_HIP_HOST_CHUNK", str(chunk))
    nobs = 50 if chunk == 1 else 20_011
    case = synthetic_case("cubic", "rectilinear", 3, [6, 7, 5], nobs, 13, np.float64, specials=False)
    assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("nhandles", [1, 2, 3, 5])
@pytest.mark.parametrize("method,kind", [("linear", "regular"), ("linear", "rectilinear"), ("cubic", "regular")])
def test_eval_host_sharded(oracle, method, kind, nhandles):
    """Single-process multi-GPU entry (`interpn_hip_eval_host_sharded`):
    contiguous ranges of the observation index, one handle each (all on this box's one GPU),
    bit-identical to the one-handle result; uneven split and ranges shorter than a row."""
    import interpn_amd

    for nobs in (10_007, 3):
        case = synthetic_case(method, kind, 3, [9, 8, 11], nobs, 21, np.float64, specials=nobs > 64)
        want = run_oracle(oracle, case, True)
        hs = []
        for _ in range(nhandles):
            if kind == "regular":
                hs.append(interpn_amd.Interpolator.regular(method, case.dims, case.starts, case.steps, case.vals))
            else:
                hs.append(interpn_amd.Interpolator.rectilinear(method, case.grids, case.vals))
        try:
            got = interpn_amd.eval_host_sharded(hs, case.obs, np.zeros(nobs))
        finally:
            for h in hs:
                h.close()
        assert_parity(case, got, want)


def test_eval_host_sharded_eight_handles_cfg5_split(oracle):
    """cfg5's split at 1/100 scale through the single-process entry point: a 128^3 grid on one
    handle, replicated device to device into seven more (`interpn_hip_replicate`; handle i on device
    i % device_count: all eight on the one GPU of a 1-GPU box, one per GPU — `hipMemcpyPeer` over
    xGMI, per-device pools, host threads pinned next to their GPU — on an 8-GPU node), 8e6 points
    cut into eight contiguous ranges, one host thread per handle.
    Bit-identical to the oracle on a sample; a NaN in the LAST range comes back as its global index
    with everything in front of it written (multilinear/regular.rs:277-280)."""
    import interpn_amd

    n, nobs = 128, 8_000_000
    case = synthetic_case("linear", "regular", 3, [n, n, n], nobs, 55, np.float64, specials=False)
    first = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
    from interpn_amd.sharded import replicate_across

    hs = replicate_across(first, 8)
    assert [h.device() for h in hs] == [i % _ndev() for i in range(8)]
    try:
        got = interpn_amd.eval_host_sharded(hs, case.obs, np.zeros(nobs))
        rng = np.random.default_rng(7)
        idx = np.unique(np.concatenate([rng.integers(0, nobs, 200_000), np.arange(nobs - 1000, nobs),
                                        np.arange(nobs // 8 - 500, nobs // 8 + 500)]))
        sub = [np.ascontiguousarray(o[idx]) for o in case.obs]
        want = np.zeros(idx.size)
        oracle.linear_regular(case.dims, case.starts, case.steps, case.vals, sub, want)
        assert np.array_equal(got[idx], want)
        k = nobs - 12_345  # in the eighth range
        case.obs[2][k] = np.nan
        out = np.full(nobs, -7.0)
        with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
            interpn_amd.eval_host_sharded(hs, case.obs, out)
        assert ei.value.first_bad_index == k
        assert np.array_equal(out[:k], got[:k])
    finally:
        for h in hs:
            h.close()


def test_eval_device_sharded(oracle):
    """Device-resident single-process form (`interpn_hip_eval_device_sharded`): every shard's
    coordinates and results live on its handle's device; all shards are enqueued before the first
    is waited for.  Three handles, handle i on device i % device_count (one GPU: all three on it), uneven shards (one of them empty-ish), 4-D
    multicubic so that one shard is large enough to be sorted first: the oracle's bits; a failing
    point in the last shard is reported with its index in shard order; a clean re-run afterwards."""
    import torch

    import interpn_amd

    from interpn_amd.sharded import replicate_across

    for method, kind, nd, axis in (("linear", "rectilinear", 3, [70, 9, 33]), ("cubic", "regular", 4, [6, 7, 5, 6])):
        sizes = [600_000, 3, 150_001]
        total = sum(sizes)
        case = synthetic_case(method, kind, nd, axis, total, 77, np.float64, extrap=0.1, specials=False)
        want = run_oracle(oracle, case, True)
        first = _make_interp(interpn_amd, case)
        hs = replicate_across(first, 3)
        assert [h.device() for h in hs] == [i % _ndev() for i in range(3)]
        try:
            shards, lo = [], 0
            for h, c in zip(hs, sizes):
                dev = torch.device("cuda", h.device())
                shards.append([torch.from_numpy(np.ascontiguousarray(o[lo:lo + c])).to(dev) for o in case.obs])
                lo += c
            outs = interpn_amd.eval_device_sharded(hs, shards)
            got = np.concatenate([o.cpu().numpy() for o in outs])
            assert np.array_equal(got, want)
            if kind == "regular":
                shards[2][1][77] = float("nan")
                shards[2][0][9_000] = float("inf")
                with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
                    interpn_amd.eval_device_sharded(hs, shards)
                assert ei.value.first_bad_index == sizes[0] + sizes[1] + 77
                shards[2][1][77] = float(case.obs[1][sizes[0] + sizes[1] + 77])
                shards[2][0][9_000] = float(case.obs[0][sizes[0] + sizes[1] + 9_000])
                outs = interpn_amd.eval_device_sharded(hs, shards, outs)
                assert np.array_equal(np.concatenate([o.cpu().numpy() for o in outs]), want)
            with pytest.raises(ValueError):
                interpn_amd.eval_device_sharded(hs[:2], shards)
        finally:
            for h in hs:
                h.close()


def test_eval_host_sharded_first_bad_index(oracle):
    """The first failing point over all ranges is reported with its global index; everything in
    front of it holds the reference's results (multilinear/regular.rs:277-280)."""
    import interpn_amd

    case = synthetic_case("linear", "regular", 2, [8, 9], 9_000, 5, np.float64, specials=False)
    k = 4_321  # inside the 2nd of 3 ranges
    case.obs[1][k] = np.nan
    case.obs[0][8_000] = np.inf  # a failure in a later range must not win
    want = np.full(9_000, -123.0)
    with pytest.raises(AssertionError) as eo:
        oracle.linear_regular(case.dims, case.starts, case.steps, case.vals, case.obs, want)
    assert eo.value.first_bad == k
    hs = [interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals) for _ in range(3)]
    got = np.full(9_000, -123.0)
    try:
        with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
            interpn_amd.eval_host_sharded(hs, case.obs, got)
        assert ei.value.first_bad_index == k
        assert np.array_equal(got[:k], want[:k])
        with pytest.raises(ValueError):
            interpn_amd.eval_host_sharded([hs[0], hs[0]], case.obs, got)  # one workspace per handle
        with pytest.raises(AssertionError, match="Dimension mismatch"):
            interpn_amd.eval_host_sharded(hs, case.obs[:1], got)
    finally:
        for h in hs:
            h.close()


def test_classes_and_helper(oracle):
    """`.new(...).eval(obs)` and `interpn()` — the three API levels of
    test/test_multilinear_regular.py:46-93 give the same (oracle-identical) result."""
    import interpn_amd

    for dtype in (np.float64, np.float32):
        for c in kat.py_on_grid_cases():
            if c.vals.dtype != dtype:
                continue
            want = run_oracle(oracle, c, True)
            if c.kind == "regular":
                cls = interpn_amd.MultilinearRegular if c.method == "linear" else interpn_amd.MulticubicRegular
                args = (c.dims, c.starts, c.steps, c.vals)
            else:
                cls = interpn_amd.MultilinearRectilinear if c.method == "linear" else interpn_amd.MulticubicRectilinear
                args = (c.grids, c.vals)
            it = cls.new(*args) if c.method == "linear" else cls.new(*args, linearize_extrapolation=False)
            got = it.eval(c.obs)
            assert got.dtype == dtype
            assert np.array_equal(got, want)
            rt = cls.model_validate_json(it.model_dump_json())
            assert np.array_equal(rt.eval(c.obs), want)
            got2 = interpn_amd.interpn(obs=c.obs, grids=c.grids, vals=c.vals, method=c.method,
                                       linearize_extrapolation=False)
            kat.check(c, got2)
            # check_bounds: nodes are inside, a far point is outside
            assert not any(it.check_bounds(c.obs, dtype(1e-6)))
            far = [np.array([-5.0]).astype(dtype), np.array([-25.0]).astype(dtype)]
            assert any(it.check_bounds(far, dtype(1e-6)))


def test_interpn_helper_contract(oracle):
    """src/interpn/__init__.py:48-194: N-D observation arrays are ravelled and the result takes
    their shape; `out=` is written in place; exact-equal spacing selects the regular path, anything
    else the rectilinear one; check_bounds raises ValueError; f32 follows `vals`."""
    import interpn_amd

    rng = np.random.default_rng(5)
    x, y = np.linspace(-1, 1, 9), np.linspace(0, 3, 7)
    vals = rng.uniform(-1, 1, (9, 7))
    obs = [rng.uniform(-1.2, 1.2, (13, 11)), rng.uniform(-0.3, 3.3, (13, 11))]
    flat = [o.ravel() for o in obs]
    for method, fn_reg, fn_rect, extra in (
            ("linear", oracle.linear_regular, oracle.linear_rectilinear, ()),
            ("cubic", oracle.cubic_regular, oracle.cubic_rectilinear, (True,)),
            ("nearest", oracle.nearest_regular, oracle.nearest_rectilinear, ())):
        want = np.zeros(flat[0].size)
        starts, steps = np.array([x[0], y[0]]), np.array([x[1] - x[0], y[1] - y[0]])
        is_regular = interpn_amd._check_regular([x, y])
        if is_regular:
            fn_reg([9, 7], starts, steps, vals.ravel(), *extra, flat, want)
        else:
            fn_rect([x, y], vals.ravel(), *extra, flat, want)
        got = interpn_amd.interpn(obs=obs, grids=[x, y], vals=vals, method=method)
        assert got.shape == (13, 11) and np.array_equal(got.ravel(), want)
        buf = np.zeros((13, 11))
        ret = interpn_amd.interpn(obs=obs, grids=[x, y], vals=vals, method=method, out=buf)
        assert np.array_equal(buf.ravel(), want) and ret.shape == (13, 11)
        # assume_regular skips the spacing check and uses grid[1]-grid[0] (:107)
        want2 = np.zeros(flat[0].size)
        fn_reg([9, 7], starts, steps, vals.ravel(), *extra, flat, want2)
        got2 = interpn_amd.interpn(obs=obs, grids=[x, y], vals=vals, method=method, assume_regular=True)
        assert np.array_equal(got2.ravel(), want2)
        # a perturbed axis goes down the rectilinear path
        xr = x.copy(); xr[3] += 1e-3
        want3 = np.zeros(flat[0].size)
        fn_rect([xr, y], vals.ravel(), *extra, flat, want3)
        got3 = interpn_amd.interpn(obs=obs, grids=[xr, y], vals=vals, method=method)
        assert np.array_equal(got3.ravel(), want3)
    with pytest.raises(ValueError, match="violate interpolator bounds"):
        interpn_amd.interpn(obs=obs, grids=[x, y], vals=vals, check_bounds=True)
    inside = [np.clip(obs[0], -1, 1), np.clip(obs[1], 0, 3)]
    interpn_amd.interpn(obs=inside, grids=[x, y], vals=vals, check_bounds=True)
    got32 = interpn_amd.interpn(obs=[o.astype(np.float32) for o in obs], grids=[x.astype(np.float32), y.astype(np.float32)],
                                vals=vals.astype(np.float32))
    assert got32.dtype == np.float32
    with pytest.raises(AssertionError, match="defined only for float32 and float64"):
        interpn_amd.interpn(obs=obs, grids=[x, y], vals=vals.astype(np.int32))


@pytest.mark.parametrize("npts", [1, 2, 3, 255, 256, 257, 100_001])
def test_device_eval_alignment_and_tails(oracle, npts):
    """3-D multilinear moves two points per lane as 16-B vectors when every stream is 16-B aligned
    and one point per lane otherwise (k_linear_brick.hip, PPL): device tensors that start on an odd
    element, odd point counts and single-point batches give the oracle's bits either way."""
    import torch

    import interpn_amd

    rng = np.random.default_rng(21)
    dims, starts, steps = [9, 8, 10], np.array([-1.0, 0.0, 2.0]), np.array([0.25, 0.5, 0.1])
    vals = rng.uniform(-1, 1, 9 * 8 * 10)
    base = [rng.uniform(starts[d] - 0.2, starts[d] + steps[d] * (dims[d] - 1) + 0.2, npts + 1) for d in range(3)]
    it = interpn_amd.Interpolator.regular("linear", dims, starts, steps, vals)
    for shift in (0, 1):  # shift = 1: obs/out start on an odd element => only 8-B aligned
        obs_np = [b[shift:shift + npts] for b in base]
        want = np.zeros(npts)
        oracle.linear_regular(dims, starts, steps, vals, [np.ascontiguousarray(o) for o in obs_np], want)
        dev = [torch.from_numpy(b).cuda()[shift:shift + npts] for b in base]
        out_full = torch.full((npts + 2,), -5.0, dtype=torch.float64, device="cuda")
        out = out_full[shift:shift + npts]
        it.eval_tensors(dev, out)
        it.finish()
        assert np.array_equal(out.cpu().numpy(), want)
        # nothing outside the requested range was written
        assert float(out_full[shift + npts]) == -5.0 and (shift == 0 or float(out_full[0]) == -5.0)


@pytest.mark.parametrize("method,axis", [("linear", [40, 50]), ("nearest", [30, 20, 10]), ("nearest", [500])])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("npts", [1, 2, 3, 127, 128, 129, 100_001])
def test_two_points_per_lane_2d_and_nearest(oracle, method, kind, axis, npts):
    """Round 2: the 2-D brick kernel and the nearest kernel move two points per lane as 16-byte
    vectors when every stream is 16-byte aligned, and fall back to the scalar form otherwise: both
    forms (tensor views at element offset 0 and 1), odd batch sizes and single points give the
    oracle's bits; `ppl` = 1 forces the scalar form."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    case = synthetic_case(method, kind, len(axis), axis, npts, 71 + npts + len(axis), np.float64, extrap=0.2, specials=False)
    want = run_oracle(oracle, case, True)
    if kind == "regular":
        it = interpn_amd.Interpolator.regular(method, case.dims, case.starts, case.steps, case.vals)
    else:
        it = interpn_amd.Interpolator.rectilinear(method, case.grids, case.vals)
    for off, forced in ((0, 0), (1, 0), (0, 1)):
        it.set_option("ppl", forced)
        bufs = [torch.zeros(npts + 2, dtype=torch.float64, device=dev) for _ in range(len(axis) + 1)]
        obs = []
        for b, o in zip(bufs, case.obs):
            b[off:off + npts] = torch.from_numpy(o).to(dev)
            obs.append(b[off:off + npts])
        out = bufs[-1][off:off + npts]
        it.eval_tensors(obs, out)
        it.finish()
        got = out.cpu().numpy()
        assert np.array_equal(got, want) or np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
        name = it.kernel_name()
        if name.startswith(("interpn::k_nearest<", "interpn::k_linear2_brick<")):
            assert name.endswith(", 2>" if (off == 0 and not forced) else ", 1>"), (name, off, forced)
    it.close()


def test_eval_device_is_graph_capturable(oracle):
    """`interpn_hip_eval_device` enqueues exactly one kernel and performs no allocation, copy or
    synchronisation, so it can be captured into a hipGraph and replayed on new data."""
    import torch

    import interpn_amd

    rng = np.random.default_rng(31)
    dims, starts, steps = [16, 12, 20], np.full(3, -1.0), np.array([2 / 15, 2 / 11, 2 / 19])
    vals = rng.uniform(-1, 1, 16 * 12 * 20)
    it = interpn_amd.Interpolator.regular("linear", dims, starts, steps, vals)
    P = 200_000
    obs = [torch.zeros(P, dtype=torch.float64, device="cuda") for _ in range(3)]
    out = torch.zeros(P, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        it.eval_tensors(obs, out)  # warm-up outside capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        it.eval_tensors(obs, out)
    for rep in range(3):
        host = [rng.uniform(-1.1, 1.1, P) for _ in range(3)]
        for d in range(3):
            obs[d].copy_(torch.from_numpy(host[d]))
        graph.replay()
        torch.cuda.synchronize()
        want = np.zeros(P)
        oracle.linear_regular(dims, starts, steps, vals, host, want)
        assert np.array_equal(out.cpu().numpy(), want), rep
    it.finish()


def test_class_eval_on_torch_tensors(oracle):
    """`.eval` of the classes accepts torch CUDA tensors: points never leave the device."""
    import torch

    import interpn_amd

    rng = np.random.default_rng(8)
    dims, starts, steps = [12, 9, 10], np.array([-1.0, 0.0, 2.0]), np.array([0.2, 0.5, 0.1])
    vals = rng.uniform(-1, 1, 12 * 9 * 10)
    obs = [rng.uniform(starts[d] - 0.1, starts[d] + steps[d] * (dims[d] - 1) + 0.1, 50_000) for d in range(3)]
    want = np.zeros(50_000)
    oracle.cubic_regular(dims, starts, steps, vals, True, obs, want)
    it = interpn_amd.MulticubicRegular.new(dims, starts, steps, vals)
    got = it.eval([torch.from_numpy(o).cuda() for o in obs])
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), want)
    bad = [torch.from_numpy(o).cuda() for o in obs]
    bad[1][123] = float("nan")
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
        it.eval(bad)
    assert ei.value.first_bad_index == 123


def test_check_bounds_matches_oracle(oracle):
    from interpn_amd import raw

    rng = np.random.default_rng(3)
    dims, starts, steps = [10, 20, 30], np.array([0.0, -1.0, 2.0]), np.array([0.1, 0.2, 0.3])
    obs = [rng.uniform(0.0, 0.9, 100_000), rng.uniform(-1.0, 2.8, 100_000), rng.uniform(2.0, 10.7, 100_000)]
    obs[1][77_777] = 2.8 + 1e-3
    got = np.zeros(3, dtype=bool)
    want = np.zeros(3, dtype=bool)
    raw.check_bounds_regular_f64(dims, starts, steps, obs, 1e-8, got)
    oracle.check_bounds_regular(dims, starts, steps, obs, 1e-8, want)
    assert np.array_equal(got, want) and list(want) == [False, True, False]
    grids = [starts[d] + steps[d] * np.arange(dims[d]) for d in range(3)]
    raw.check_bounds_rectilinear_f64(grids, obs, 1e-8, got)
    oracle.check_bounds_rectilinear(grids, obs, 1e-8, want)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_check_bounds_on_device_tensors(oracle, dtype):
    """`check_bounds` with the points already on the device (interpn_hip_check_bounds_device):
    limits come from the resident grid; same flags as the oracle on the same coordinates, for
    regular and rectilinear grids, including a violation smaller / larger than atol."""
    import torch

    import interpn_amd

    rng = np.random.default_rng(3)
    dims = [10, 20, 30]
    starts, steps = np.array([0.0, -1.0, 2.0], dtype=dtype), np.array([0.1, 0.2, 0.3], dtype=dtype)
    grids = [(starts[d] + steps[d] * np.arange(dims[d], dtype=dtype)).astype(dtype) for d in range(3)]
    vals = rng.uniform(-1, 1, int(np.prod(dims))).astype(dtype)
    reg = interpn_amd.MultilinearRegular.new(dims, starts, steps, vals)
    rect = interpn_amd.MulticubicRectilinear.new(grids, vals)
    atol = 1e-4
    for violate in (None, (1, 5e-3), (2, -5e-3), (0, 1e-6)):
        obs = [rng.uniform(float(g[0]), float(g[-1]), 200_003).astype(dtype) for g in grids]
        if violate is not None:
            d, eps = violate
            obs[d][12_345] = (grids[d][-1] + dtype(eps)) if eps > 0 else (grids[d][0] + dtype(eps))
        obs_t = [torch.from_numpy(o).cuda() for o in obs]
        want = np.zeros(3, dtype=bool)
        oracle.check_bounds_regular(dims, starts, steps, obs, dtype(atol), want)
        assert np.array_equal(reg.check_bounds(obs_t, atol), want)
        assert np.array_equal(reg.check_bounds(obs, atol), want)  # host form, same answer
        oracle.check_bounds_rectilinear(grids, obs, dtype(atol), want)
        assert np.array_equal(rect.check_bounds(obs_t, atol), want)
        if violate is not None and abs(violate[1]) > atol:
            assert want[violate[0]]


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_cfg1_plumbing_case(oracle, dtype):
    """BASELINE configs[0] as stated: 2-D multilinear::regular, 4x4 grid, 1e3 obs
    (src/multilinear/regular.rs:51-117, the N = 2 arm) — through every entry point a caller of
    that case can take: raw one-shot, `interpn()`, the class on host arrays, the resident handle on
    host arrays and on device tensors.  ~5 % of the points extrapolate; exact nodes, domain ends
    and +-0 are injected; a linear field is reproduced to 1e-12 like the reference's own test
    (`regular.rs:438-477`); NaN aborts at the first bad point with the prefix written."""
    import torch

    import interpn_amd
    from interpn_amd import raw

    case = synthetic_case("linear", "regular", 2, [4, 4], 1000, 4242, dtype)
    dims = [4, 4]
    starts = np.array([g[0] for g in case.grids], dtype=dtype)
    steps = np.array([g[1] - g[0] for g in case.grids], dtype=dtype)
    want = run_oracle(oracle, case, True)
    # raw one-shot
    assert_parity(case, run_hip_raw(case), want)
    # resident handle, host arrays and device tensors
    it = interpn_amd.Interpolator.regular("linear", dims, starts, steps, case.vals)
    got = np.zeros(1000, dtype=dtype)
    it.eval_host(case.obs, got)
    assert_parity(case, got, want)
    dev = torch.device("cuda:0")
    out = it.eval_tensors([torch.from_numpy(o).to(dev) for o in case.obs])
    it.finish()
    assert_parity(case, out.cpu().numpy(), want)
    # class + helper (what the reference's Python tests call)
    cls = interpn_amd.MultilinearRegular.new(dims, starts, steps, case.vals)
    assert_parity(case, cls.eval(case.obs), want)
    helper = interpn_amd.interpn(obs=case.obs, grids=[np.asarray(g) for g in case.grids], vals=case.vals.reshape(4, 4),
                                 method="linear", assume_regular=True)
    assert_parity(case, np.asarray(helper).ravel(), want)
    # the reference's own assertion for this arm: a field linear in the coordinates is reproduced
    if dtype == np.float64:
        mesh = np.stack(np.meshgrid(*case.grids, indexing="ij"), axis=-1).reshape(-1, 2)
        lin = np.ascontiguousarray(mesh @ np.array([1.0, 1.0]))
        got = np.zeros(1000)
        raw.interpn_linear_regular_f64(dims, starts, steps, lin, case.obs, got)
        assert np.max(np.abs(got - (case.obs[0] + case.obs[1]))) < 1e-12
    # abort at the first bad point, prefix written, rest untouched (regular.rs:277-280, :418)
    bad = [o.copy() for o in case.obs]
    bad[1][617] = np.nan
    got = np.full(1000, -7.0, dtype=dtype)
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value"):
        getattr(raw, "interpn_linear_regular_" + ("f64" if dtype == np.float64 else "f32"))(
            dims, starts, steps, case.vals, bad, got)
    assert np.array_equal(got[:617], want[:617]) and np.all(got[617:] == -7.0)
    it.close()


def test_device_tensors_full_size_properties(oracle):
    """BASELINE config 2 at full size (3-D multilinear-regular, 64^3 grid, 1e8 obs) on device
    tensors: (a) a sampled subset is bit-identical to the oracle; (b) evaluating a sub-range
    separately reproduces the corresponding slice bit for bit (no dependence on launch
    geometry); (c) a field that is linear in every coordinate is reproduced to 1e-12."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n, P = 64, 100_000_000
    g = np.linspace(-1.0, 1.0, n)
    dims, starts, steps = [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0])
    rng = np.random.default_rng(11)
    vals = rng.uniform(-1, 1, n**3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(3)]
    it = interpn_amd.Interpolator.regular("linear", dims, starts, steps, vals)
    out = it.eval_tensors(obs)
    it.finish()
    # the headline batch takes the sweep kernel on the one-line table (linear_sweep.h); the brick kernel
    # on the handle's (1,2) table must give the same bits over the whole batch
    assert it.last_path == "sweep" and it.kernel_name().startswith("interpn::k_linear_sweep<double, false, true, 1, 1, 12, 768, 0, false"), it.kernel_name()
    it.set_option("sweep", 0)
    brick = it.eval_tensors(obs)
    it.finish()
    assert it.kernel_name() == "interpn::k_linear_brick<double, 3, false, true, 1, 2, 2, 0, 0, 0>", it.kernel_name()
    assert torch.equal(brick, out)
    del brick
    it.set_option("sweep", -1)
    # (a) sampled subset vs oracle
    idx = torch.randint(0, P, (500_000,), device=dev, generator=gen)
    sub = [o[idx].cpu().numpy() for o in obs]
    want = np.zeros(idx.numel())
    oracle.linear_regular(dims, starts, steps, vals, sub, want)
    assert np.array_equal(out[idx].cpu().numpy(), want)
    # (b) sub-range == slice
    lo, hi = 12_345_678, 23_456_789
    part = it.eval_tensors([o[lo:hi].contiguous() for o in obs])
    it.finish()
    assert torch.equal(part, out[lo:hi])
    # (c) linear field
    mesh = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    lin = mesh @ np.array([0.5, -1.25, 2.0]) + 0.75
    it2 = interpn_amd.Interpolator.regular("linear", dims, starts, steps, np.ascontiguousarray(lin))
    out2 = it2.eval_tensors(obs)
    it2.finish()
    exact = 0.5 * obs[0] - 1.25 * obs[1] + 2.0 * obs[2] + 0.75
    assert float((out2 - exact).abs().max()) < 1e-12


def _full_size_checks(torch, it, obs, oracle_fn, sample, sub_range):
    """Shared body of the BASELINE-size tests: one launch over the whole batch, then (a) a random
    sample bit-compared with the oracle, (b) a sub-range evaluated on its own == the slice."""
    dev = obs[0].device
    P = obs[0].numel()
    out = it.eval_tensors(obs)
    it.finish()
    full_name = it.kernel_name()  # the instantiation the full-size launch ran
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321)
    idx = torch.randint(0, P, (sample,), device=dev, generator=gen)
    sub = [o[idx].cpu().numpy() for o in obs]
    want = np.zeros(sample)
    oracle_fn(sub, want)
    got = out[idx].cpu().numpy()
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    assert np.all(same), (int((~same).sum()), float(np.nanmax(np.abs(got - want))))
    lo, hi = sub_range
    part = it.eval_tensors([o[lo:hi] for o in obs])  # views: an odd `lo` also exercises the 8-byte-aligned path
    it.finish()
    assert torch.equal(part, out[lo:hi])
    return out, full_name


def test_cfg3_full_size_rectilinear(oracle):
    """BASELINE config 3 at full size: 3-D multilinear-rectilinear, non-uniform 64^3 grid, 1e8
    unordered obs (~5 % outside the grid: rectilinear extrapolates, it never fails).
    multilinear/rectilinear.rs:244-370."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n, P = 64, 100_000_000
    rng = np.random.default_rng(21)
    g = np.linspace(-1.0, 1.0, n)
    step = g[1] - g[0]
    grids = []
    for _ in range(3):
        j = (rng.random(n) - 0.5) * 0.5 * step
        j[0] = j[-1] = 0.0
        grids.append(g + j)
    vals = rng.uniform(-1, 1, n**3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(3)]
    it = interpn_amd.Interpolator.rectilinear("linear", grids, vals)
    _, name = _full_size_checks(torch, it, obs, lambda sub, want: oracle.linear_rectilinear(grids, vals, sub, want),
                                500_000, (31_234_567, 47_000_001))  # (_ = the full batch's results)
    # a batch of this size takes the sweep kernel (axes in lanes, lane tables: AXR = 2); the odd sub-range
    # of _full_size_checks went through the brick kernel and had to give the same bits
    assert name.startswith("interpn::k_linear_sweep<double, true, true, 1, 1, 12, 768, 2, false"), name
    it.set_option("sweep", 0)
    again = it.eval_tensors([o[:20_000_000] for o in obs])
    it.finish()
    assert it.kernel_name().startswith("interpn::k_linear_brick<double, 3, true, true,") and it.kernel_name().endswith(", 2, 2, 0, 0>"), it.kernel_name()
    assert torch.equal(again, _[:20_000_000])
    it.close()


@pytest.mark.parametrize("linearize", [False, True], ids=["quad", "lin"])
def test_cfg4_full_size_cubic_4d(oracle, linearize):
    """BASELINE config 4 at its grid size: 4-D multicubic-regular, 32^4 f64 grid (the 128 MiB
    fully overlapped tile table and its plane strides), 1e7 obs of which ~10 % extrapolate so
    that both `linearize_extrapolation` flags differ.  multicubic/regular.rs:325-623."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n, P = 32, 10_000_000
    g = np.linspace(-1.0, 1.0, n)
    dims, starts, steps = [n] * 4, np.full(4, -1.0), np.full(4, g[1] - g[0])
    vals = np.random.default_rng(22).uniform(-1, 1, n**4)
    gen = torch.Generator(device=dev)
    gen.manual_seed(78)
    obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(4)]
    it = interpn_amd.Interpolator.regular("cubic", dims, starts, steps, vals, linearize_extrapolation=linearize)
    _, name = _full_size_checks(torch, it, obs,
                                lambda sub, want: oracle.cubic_regular(dims, starts, steps, vals, linearize, sub, want),
                                200_000, (1_234_567, 4_700_001))
    tbytes, si, sj = it.table_layout()
    # large batches are sorted by cell and evaluated out of an LDS-resident table column
    assert name.startswith(("interpn::k_cubic_column<double,", "interpn::k_cubic_brick<double, 4, false, true,")), name
    assert tbytes >= 8 * n**4  # a re-laid copy is in use (its layout is the heuristic's choice)
    it.close()


def test_cfg5_shard_full_size(oracle):
    """One shard of BASELINE config 5: 3-D multilinear-regular on the 128^3 grid (86 MiB fully
    overlapped brick table, Infinity-Cache resident), 1e8 obs.  Also the linear-field property at
    this size (multilinear reproduces a field that is linear in every coordinate to 1e-12)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n, P = 128, 100_000_000
    g = np.linspace(-1.0, 1.0, n)
    dims, starts, steps = [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0])
    vals = np.random.default_rng(23).uniform(-1, 1, n**3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(79)
    obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(3)]
    it = interpn_amd.Interpolator.regular("linear", dims, starts, steps, vals)
    _, name = _full_size_checks(torch, it, obs,
                                lambda sub, want: oracle.linear_regular(dims, starts, steps, vals, sub, want),
                                500_000, (61_234_567, 77_000_001))
    assert name.startswith("interpn::k_linear_sweep<double, false, true, 1, 1, 12, 768, 0, false"), name
    it.set_option("sweep", 0)
    again = it.eval_tensors([o[:20_000_000] for o in obs])
    it.finish()
    assert it.kernel_name() == "interpn::k_linear_brick<double, 3, false, true, 1, 1, 2, 0, 0, 0>", it.kernel_name()
    assert torch.equal(again, _[:20_000_000])
    it.close()
    mesh = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    lin = np.ascontiguousarray(mesh @ np.array([0.5, -1.25, 2.0]) + 0.75)
    it2 = interpn_amd.Interpolator.regular("linear", dims, starts, steps, lin)
    out2 = it2.eval_tensors(obs)
    it2.finish()
    exact = 0.5 * obs[0] - 1.25 * obs[1] + 2.0 * obs[2] + 0.75
    assert float((out2 - exact).abs().max()) < 1e-12
    it2.close()


def test_handle_options_and_kernel_name(oracle):
    """Tuning knobs are per-handle state (latched from the environment at creation, changed with
    interpn_hip_set_option), never read on the launch path; the kernel name is reported by the
    handle."""
    import interpn_amd

    case = synthetic_case("linear", "regular", 3, [20, 21, 22], 50_001, 5)
    want = run_oracle(oracle, case, True)
    it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
    assert it.kernel_name() == ""
    assert it.get_option("ppl") == 0 and it.get_option("force_generic") == 0
    out = it.eval_host(case.obs, np.zeros_like(want))
    assert np.array_equal(out, want)
    first = it.kernel_name()
    assert first.startswith("interpn::k_linear_brick<double, 3, false, true,"), first
    it.set_option("force_generic", 1)
    out = it.eval_host(case.obs, np.zeros_like(want))
    assert np.array_equal(out, want)
    assert it.kernel_name().startswith("interpn::k_generic<double,"), it.kernel_name()
    it.set_option("force_generic", 0)
    it.set_option("ppl", 1)
    out = it.eval_host(case.obs, np.zeros_like(want))
    assert np.array_equal(out, want)
    assert it.kernel_name().endswith(", 1, 0, 0, 0>"), it.kernel_name()  # PPL = 1, AXR = 0, ABL = 0, CELL = 0
    with pytest.raises(ValueError):
        it.set_option("no_such_option", 1)
    with pytest.raises(ValueError):
        it.set_option("ppl", 7)
    it.close()


def test_thresholds_are_derived_from_the_device(oracle):
    """The tuning thresholds (which table layout, when to sort, how much LDS the axes / the column
    may take) are functions of the device's L2, LDS and CU count (interpn_host.h::Thresholds),
    queried once per device.  On MI355X they evaluate to the constants the measurements behind
    DESIGN.md were taken with."""
    import torch

    import interpn_amd

    case = synthetic_case("linear", "regular", 3, [5, 6, 7], 100, 1234, np.float64)
    it = _make_interp(interpn_amd, case)
    props = torch.cuda.get_device_properties(0)
    assert it.get_option("dev_num_cus") == props.multi_processor_count
    l2, cus = it.get_option("dev_l2_bytes"), it.get_option("dev_num_cus")
    lds_cu, lds_wg = it.get_option("dev_lds_per_cu"), it.get_option("dev_lds_per_wg")
    assert it.get_option("thr_table_l2_sized") == l2 * 3 // 2 and it.get_option("thr_table_l2_share") == l2 * 3 // 4
    assert it.get_option("thr_binned_table_min") == 2 * l2 and it.get_option("thr_binned_points_min") == 2048 * cus
    assert it.get_option("thr_bin_table_share") == l2 // 8
    assert it.get_option("thr_axis_lds") == lds_cu // 8 and it.get_option("thr_axis_lds_wide") == lds_wg - 4096
    assert it.get_option("thr_column_lds") == lds_cu
    if "MI355" in props.name or (cus == 256 and l2 == 4 << 20):
        assert (l2, cus, lds_cu, lds_wg, it.get_option("dev_num_xcds")) == (4 << 20, 256, 160 << 10, 64 << 10, 8)
        assert it.get_option("thr_table_l2_sized") == 6 << 20 and it.get_option("thr_table_l2_share") == 3 << 20
        assert it.get_option("thr_binned_table_min") == 8 << 20 and it.get_option("thr_binned_points_min") == 1 << 19
        assert it.get_option("thr_axis_lds") == 20 << 10 and it.get_option("thr_axis_lds_wide") == 60 << 10
        assert it.get_option("thr_column_lds") == 160 << 10
    it.close()


def test_column_plan_fits_the_lds(oracle):
    """How the column evaluation lays a grid's (k, l) column of tiles out in LDS (k_cubic_column.hip::
    cubic_column_plan, read back through the col_* options): never more than a CU's LDS; cfg4's 32 x 32
    f64 column (144 KiB with 16 bytes of padding per tile) is resident whole — the local order of a
    12 288-point part sits in the padding (8 entries per tile) and 8 KiB behind the column —; f32 tiles are
    80 bytes apart; 48^4 takes three K-range phases; bare tiles (column_pad 0) and 8-bit local sort keys."""
    import interpn_amd

    rng = np.random.default_rng(5)
    for n, dtype in ((32, np.float64), (32, np.float32), (48, np.float64), (8, np.float64)):
        g = np.linspace(-1.0, 1.0, n)
        it = interpn_amd.Interpolator.regular("cubic", [n] * 4, np.full(4, -1.0, dtype=dtype), np.full(4, g[1] - g[0], dtype=dtype),
                                              rng.uniform(-1, 1, n**4).astype(dtype), False, 0, dtype)
        assert it.get_option("col_applies") == 1
        lds_cu = it.get_option("dev_lds_per_cu")
        elem = 8 if dtype == np.float64 else 4
        for pad in (-1, 1, 0):
            it.set_option("column_pad", pad)
            pitch, nphase, cpp = it.get_option("col_pitch"), it.get_option("col_nphase"), it.get_option("col_cpp")
            assert it.get_option("col_lds_bytes") + 4352 * it.get_option("col_groups") <= lds_cu
            assert pitch == (16 * elem if pad == 0 else 16 * elem + 16)
            assert nphase * cpp >= n - 1 and (nphase - 1) * cpp < n - 1
            assert it.get_option("col_perm_pad") == (0 if pad == 0 else 8 * n * (n if nphase == 1 else min(n, cpp + 3)))
            assert it.get_option("col_q3") * (n - 1) <= 256  # keys ride in the index words of regular grids
        it.set_option("column_pad", -1)
        if lds_cu == 160 << 10:
            assert it.get_option("col_nphase") == (3 if n == 48 else 1)
            assert it.get_option("col_part_points") == 12288
        it.set_option("column_keys", 0)
        assert 1 <= it.get_option("col_q3") and it.get_option("col_q3") * (n - 1) <= 1024
        it.close()


@pytest.mark.parametrize("method,kind,axis", [("linear", "regular", [20, 21, 22]), ("linear", "rectilinear", [33, 9, 40]),
                                              ("cubic", "rectilinear", [9, 8, 7]), ("linear", "regular", [9, 8, 7, 6]),
                                              ("nearest", "rectilinear", [300, 40])])
def test_replicate_device_to_device(oracle, method, kind, axis):
    """interpn_hip_replicate clones a handle onto a device with a device-to-device copy of the grid
    (and of the rectilinear axis image) and rebuilds the re-laid table there.  The target is the LAST
    visible device (another GPU wherever there is one: the copy then crosses xGMI); on a 1-GPU box
    it is the same device: the clone must own its memory (the source is destroyed first) and
    give the oracle's bits; the pair then serves interpn_hip_eval_host_sharded."""
    import interpn_amd
    from interpn_amd.handle import eval_host_sharded

    case = synthetic_case(method, kind, len(axis), axis, 30_011, 17 + sum(axis), np.float64, linearize=True, extrap=0.1)
    want = run_oracle(oracle, case, True)
    if kind == "regular":
        it = interpn_amd.Interpolator.regular(method, case.dims, case.starts, case.steps, case.vals, True)
    else:
        it = interpn_amd.Interpolator.rectilinear(method, case.grids, case.vals, True)
    clone = it.replicate(_ndev() - 1)
    assert clone.device() == _ndev() - 1 and it.device() == 0
    both = eval_host_sharded([it, clone], case.obs, np.zeros_like(want))
    assert_parity(case, both, want)
    it.close()
    out = clone.eval_host(case.obs, np.zeros_like(want))
    assert_parity(case, out, want)
    clone.close()


def test_destroy_does_not_wait_for_other_streams(oracle):
    """interpn_hip_destroy waits for the handle's own work only: a long-running
    kernel on an unrelated stream must still be running when destroy returns."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    case = synthetic_case("linear", "regular", 3, [16, 16, 16], 10_000, 6)
    want = run_oracle(oracle, case, True)
    obs = [torch.from_numpy(o).to(dev) for o in case.obs]
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    # The runtime maps streams onto a few hardware queues; when the side stream happens to share
    # the default stream's queue, the evaluation itself queues up behind the unrelated work and no
    # destroy could return early.  A destroy that synchronised the device would fail EVERY attempt;
    # one attempt on a side stream with its own queue is enough to show it does not.
    seen_running = False
    for _attempt in range(6):
        it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
        side = torch.cuda.Stream(device=dev)
        done = torch.cuda.Event()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(60):  # ~1 GiB written 60 times: tens of milliseconds of unrelated work
                big.add_(1.0)
            done.record(side)
        out = it.eval_tensors(obs)  # on torch's current (default) stream
        it.close()
        still_running = not done.query()
        assert np.array_equal(out.cpu().numpy(), want)
        side.synchronize()
        seen_running = seen_running or still_running
        if seen_running:
            break
    assert seen_running, "destroy waited for an unrelated stream"


def test_differential_fuzz_short(oracle):
    """A fixed-seed slice of tools/fuzz_parity.py (random method / kind / N / axis sizes / dtype /
    layout and scheduling knobs / special coordinates): every case bit-identical to the oracle.
    Long runs by hand: 900 000 cases over several seeds (the longest 40 minutes), 0 differences."""
    from tools.fuzz_parity import run

    cases, failures = run(budget=12.0, seed=20261003, max_cases=1500)
    assert cases > 100
    assert failures == 0


def test_c_consumer_runs(tmp_path):
    """examples/c_abi_demo.c — a pure-C program linked against libinterpn_hip.so (no Python, no
    torch in the process): one-shot and handle calls, abort semantics, the reference's error
    strings, a cubic rectilinear call."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "interpn_amd")
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "c_abi_demo.c"), "-L", libdir, "-linterpn_hip",
                           f"-Wl,-rpath,{libdir}", "-lm", "-o", exe])
    for pool_mb in (None, "0"):  # default pool of freed device blocks, and pool disabled
        env = dict(os.environ)
        if pool_mb is not None:
            env["INTERPN_HIP_POOL_MB"] = pool_mb
        res = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
        assert res.returncode == 0, res.stdout + res.stderr
        assert "ALL PASSED" in res.stdout and "FAIL " not in res.stdout


def test_one_shot_calls_do_not_leak_device_memory(oracle):
    """The one-shot entry points rebuild the interpolator per call (as the reference does) and
    recycle device blocks through a per-device pool capped at INTERPN_HIP_POOL_MB (1 GiB): a long
    run of calls with changing grid and batch sizes must not grow the device footprint past it."""
    import torch

    from interpn_amd import raw

    rng = np.random.default_rng(11)
    torch.cuda.synchronize()
    free0, _total = torch.cuda.mem_get_info()
    for k in range(400):
        n = int(rng.integers(2, 40))
        nobs = int(rng.integers(1, 50_000))
        g = np.linspace(-1.0, 1.0, n)
        vals = rng.uniform(-1, 1, n ** 3)
        obs = [rng.uniform(-1.1, 1.1, nobs) for _ in range(3)]
        out = np.zeros(nobs)
        raw.interpn_linear_regular_f64([n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals, obs, out)
        if k % 100 == 0:
            want = np.zeros(nobs)
            oracle.linear_regular([n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0]), vals, obs, want)
            assert np.array_equal(out, want)
    torch.cuda.synchronize()
    free1, _total = torch.cuda.mem_get_info()
    assert free0 - free1 < (1 << 30) + (256 << 20), f"device footprint grew by {(free0 - free1) >> 20} MiB"


def test_concurrent_host_calls_from_threads(oracle):
    """The ABI is re-entrant (threading): one-shot calls and evaluations on a
    shared resident handle from eight host threads at once, every result bit-identical."""
    import threading

    import interpn_amd
    from interpn_amd import raw

    rng = np.random.default_rng(5)
    n = 17
    g = np.linspace(-1.0, 1.0, n)
    dims, starts, steps = [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0])
    vals = rng.uniform(-1, 1, n ** 3)
    shared = interpn_amd.MultilinearRegular.new(dims, starts, steps, vals)
    errors = []

    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            for k in range(25):
                nobs = int(r.integers(1, 30_000))
                obs = [r.uniform(-1.2, 1.2, nobs) for _ in range(3)]
                want = np.zeros(nobs)
                oracle.linear_regular(dims, starts, steps, vals, obs, want)
                got = np.zeros(nobs)
                if k % 2:
                    raw.interpn_linear_regular_f64(dims, starts, steps, vals, obs, got)
                else:
                    cub = np.zeros(nobs)
                    raw.interpn_cubic_regular_f64(dims, starts, steps, vals, True, obs, cub)
                    wantc = np.zeros(nobs)
                    oracle.cubic_regular(dims, starts, steps, vals, True, obs, wantc)
                    if not np.array_equal(cub, wantc):
                        errors.append(f"cubic one-shot differs (seed {seed}, call {k})")
                    raw.interpn_linear_regular_f64(dims, starts, steps, vals, obs, got)
                if not np.array_equal(got, want):
                    errors.append(f"linear one-shot differs (seed {seed}, call {k})")
                if not np.array_equal(shared.eval(obs), want):  # one handle, many threads
                    errors.append(f"shared-handle eval differs (seed {seed}, call {k})")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(100 + i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    del shared
    assert not errors, errors[:3]


@pytest.mark.parametrize("method", ["linear", "cubic", "nearest"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
def test_interpn_helper_on_device_tensors(oracle, method, kind):
    """`interpn()` with the observation points as torch CUDA tensors: same dispatch rules as the
    host form, same bits, the result comes back as a CUDA tensor of the observation shape; the
    optional bounds check runs on the device too."""
    import torch

    import interpn_amd

    rng = np.random.default_rng(8)
    grids = [np.linspace(-1.0, 1.0, 9), np.linspace(0.0, 3.0, 7)]
    if kind == "rectilinear":
        grids[1] = np.array([0.0, 0.4, 1.0, 1.1, 2.0, 2.7, 3.0])
    vals = rng.uniform(-1, 1, (9, 7))
    xs = rng.uniform(-1.0, 1.0, (50, 40))
    ys = rng.uniform(0.0, 3.0, (50, 40))
    host = interpn_amd.interpn([xs, ys], grids, vals, method=method)
    dev = interpn_amd.interpn([torch.from_numpy(xs).cuda(), torch.from_numpy(ys).cuda()], grids, vals, method=method,
                              check_bounds=True)
    assert dev.is_cuda and tuple(dev.shape) == xs.shape
    assert np.array_equal(dev.cpu().numpy(), host)
    ys_bad = torch.from_numpy(ys).cuda()
    ys_bad[3, 3] = 3.5
    with pytest.raises(ValueError, match="violate interpolator bounds"):
        interpn_amd.interpn([torch.from_numpy(xs).cuda(), ys_bad], grids, vals, method=method, check_bounds=True)


@pytest.mark.parametrize("method", ["linear", "cubic", "nearest"])
@pytest.mark.parametrize("axis", [[3000], [6000], [1500, 900], [2800, 2500]], ids=str)
def test_rectilinear_long_axes(oracle, monkeypatch, method, axis):
    """1-D / 2-D rectilinear axes of a few thousand coordinates: the axis image (coordinates +
    bucket tables, 12 bytes per coordinate) is searched in LDS up to 60 KiB in the kernels that
    have no other LDS use, and through L1/L2 beyond; bricks on and off."""
    for bricks in (None, "off"):
        if bricks:
            monkeypatch.setenv("INTERPN_HIP_BRICKS", bricks)
        else:
            monkeypatch.delenv("INTERPN_HIP_BRICKS", raising=False)
        case = synthetic_case(method, "rectilinear", len(axis), axis, 60_007, 5000 + sum(axis), np.float64,
                              linearize=True, extrap=0.1)
        assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


def _make_interp(interpn_amd, case):
    if case.kind == "regular":
        return interpn_amd.Interpolator.regular(case.method, case.dims, case.starts, case.steps, case.vals,
                                                linearize_extrapolation=case.linearize)
    return interpn_amd.Interpolator.rectilinear(case.method, case.grids, case.vals, linearize_extrapolation=case.linearize)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("axis,layout", [([9, 7], "11"), ([6, 5, 7], "44"), ([5, 6, 4, 7], "11"), ([4, 4, 4, 4], "24"),
                                         ([40, 37, 5, 4], "11")], ids=str)
def test_binned_multicubic_evaluation(oracle, monkeypatch, dtype, kind, axis, layout):
    """Binned evaluation (k_bin_points.hip): the points of a device-resident batch are
    counting-sorted by the tile position of their footprint, the tiled multicubic kernel reads the
    sorted copy and scatters its results to out[original index].  Forced here for N = 2..4 at
    small sizes: batches of one point, less than a chunk (4096), several chunks with a ragged
    tail, grids with more tile positions than bins (key shifts), minimum-size grids (one bin),
    sorted order dealt out to the XCDs or not, several tile layouts (forced: grids this small
    would stay on the C-order kernels, which have no binned form) — always the oracle's bits,
    NaN / inf coordinates and out-of-range points included.  src/multicubic/regular.rs:297-313 (a point's result depends
    on its own coordinates only)."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", layout)
    dev = torch.device("cuda:0")
    n = len(axis)
    want_t = torch.float64 if dtype == np.float64 else torch.float32
    for nobs, deal in ((1, 1), (255, 1), (4097, 0), (16_384, 1), (32_700, 1), (32_700, 0), (70_001, 1), (600_011, 1)):
        case = synthetic_case("cubic", kind, n, axis, nobs, 9100 + sum(axis) + nobs, dtype, linearize=bool(nobs % 2),
                              extrap=0.3, specials=min(axis) >= 8)
        if kind == "rectilinear" and nobs > 100:  # rectilinear grids never fail per point: NaN propagates
            case.obs[0][17] = np.nan
            case.obs[n - 1][18] = np.inf
        want = run_oracle(oracle, case, True)
        it = _make_interp(interpn_amd, case)
        it.set_option("binned", 1)
        it.set_option("deal", deal)
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        out_full = torch.full((nobs + 2,), -5.0, dtype=want_t, device=dev)
        out = out_full[1:1 + nobs]
        it.eval_tensors(obs, out)
        it.finish()
        assert it.get_option("last_binned") == 1
        got = out.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)]), (nobs, deal)
        assert float(out_full[0]) == -5.0 and float(out_full[-1]) == -5.0  # nothing outside the batch was written
        # the same handle with the points evaluated in place gives the same bits
        it.set_option("binned", 0)
        got0 = it.eval_tensors(obs).cpu().numpy()
        it.finish()
        assert it.get_option("last_binned") == 0
        assert np.array_equal(np.isnan(got0), np.isnan(got)) and np.array_equal(got0[~np.isnan(got0)], got[~np.isnan(got)])
        it.close()


def test_binned_evaluation_first_bad_index_and_concurrency(oracle, monkeypatch):
    """Binned evaluation keeps the device-pointer contract: the sticky status reports the SMALLEST
    failing original index although the points were evaluated in table order; two streams
    evaluating through one handle take turns in its scratch (the second waits on the first's
    event); under graph capture the call falls back to the one-kernel form."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    dev = torch.device("cuda:0")
    case = synthetic_case("cubic", "regular", 4, [6, 7, 5, 6], 200_000, 9300, np.float64, extrap=0.2, specials=False)
    want = run_oracle(oracle, case, True)
    it = _make_interp(interpn_amd, case)
    it.set_option("binned", 1)
    obs = [torch.from_numpy(o).to(dev) for o in case.obs]
    bad = [o.clone() for o in obs]
    bad[2][150_000] = float("nan")
    bad[0][60_123] = float("inf")
    bad[3][199_999] = float("nan")
    out = it.eval_tensors(bad)
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as err:
        it.finish()
    assert err.value.first_bad_index == 60_123
    assert it.get_option("last_binned") == 1
    assert np.array_equal(out[:60_123].cpu().numpy(), want[:60_123])
    # two streams, one handle
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for rep in range(4):
        for s in (s1, s2):
            with torch.cuda.stream(s):
                outs.append(it.eval_tensors(obs))
    torch.cuda.synchronize()
    it.finish()
    for o in outs:
        assert np.array_equal(o.cpu().numpy(), want)
    # capture: no allocation, no extra launches -> the direct kernel
    res = torch.zeros(200_000, dtype=torch.float64, device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        it.eval_tensors(obs, res)
    assert it.get_option("last_binned") == 0
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(res.cpu().numpy(), want)
    it.finish()
    it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("axis,layout", [([12, 11], "11"), ([10, 12, 9], "44"), ([9, 10, 8, 9], "11"), ([9, 10, 8, 9], "22")], ids=str)
@pytest.mark.parametrize("region", ["interior", "dim0_interior", "mixed"])
def test_cubic_wave_uniform_interior_nodes(oracle, monkeypatch, dtype, axis, layout, region):
    """The tiled multicubic kernel evaluates dims 0 and 1 with a select-free node when every lane
    of a wave sits in an interior cell along that dimension (cubic_regular_node_interior: the
    reference's Saturation::None arm, multicubic/regular.rs:495-505).  Batches that are interior
    everywhere, interior along dim 0 only, and mixed must give the oracle's bits; both
    `linearize_extrapolation` flags."""
    monkeypatch.setenv("INTERPN_HIP_BRICKS", layout)
    n = len(axis)
    for linearize in (False, True):
        case = synthetic_case("cubic", "regular", n, axis, 20_003, 9500 + sum(axis), dtype, linearize=linearize, extrap=0.2,
                              specials=False)
        rng = np.random.default_rng(5)
        for d in range(n):
            g = case.grids[d].astype(np.float64)
            if region == "interior" or (region == "dim0_interior" and d == 0):
                case.obs[d] = rng.uniform(g[1] + 1e-3, g[-2] - 1e-3, 20_003).astype(dtype)
        assert_parity(case, run_hip_raw(case), run_oracle(oracle, case, True))


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("axis", ["n2", "n3", "jitter_100", "jitter_9000", "log_300", "two_scales_6000", "dense_pairs"])
def test_linear_1d_rectilinear_records(oracle, monkeypatch, dtype, fma, axis):
    """1-D multilinear on a rectilinear axis from one record per search bucket
    (k_linear1_records.hip): chosen by itself for axes too long for the LDS search (9000 points),
    forced here for short ones (INTERPN_HIP_BRICKS=on); axes on which no bucket count separates
    the coordinates within the table budget keep the general kernel.  Exact nodes, both ends,
    NaN, +-inf and far-away coordinates included; multilinear/rectilinear.rs:353-370 (the cell),
    :310-313 and :339-344 (the arithmetic)."""
    import interpn_amd
    from interpn_amd import _lib

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "on")
    rng = np.random.default_rng(77)
    if axis == "n2":
        g = np.array([-0.5, 2.0])
    elif axis == "n3":
        g = np.array([-1.0, -0.75, 3.0])
    elif axis.startswith("jitter"):
        n = int(axis.split("_")[1])
        g = np.linspace(-1.0, 1.0, n)
        g[1:-1] += (rng.uniform(size=n - 2) - 0.5) * 0.5 * (g[1] - g[0])
    elif axis == "log_300":
        g = -1.0 + 2.0 * (np.logspace(0, 3, 300) - 1.0) / 999.0
    elif axis == "two_scales_6000":
        g = np.concatenate([np.linspace(-1.0, -0.9, 3000, endpoint=False), np.linspace(-0.9, 1.0, 3000)])
    else:  # pairs of coordinates 1e-9 apart: no affordable bucket count separates them
        base = np.linspace(-1.0, 1.0, 40)
        g = np.sort(np.concatenate([base, base[:-1] + 1e-9]))
    g = g.astype(dtype)
    g = np.unique(g)  # f32 rounding may merge neighbours
    assert np.all(np.diff(g) > 0)
    vals = rng.uniform(-1, 1, g.size).astype(dtype)
    nobs = 50_003
    lo, hi = float(g[0]), float(g[-1])
    x = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), nobs).astype(dtype)
    k = min(g.size, 300)
    x[:k] = g[:k]
    x[k:k + 4] = [g[-1], np.nextafter(g[-1], np.inf, dtype=dtype), np.nextafter(g[0], -np.inf, dtype=dtype), 0.0]
    x[k + 4:k + 9] = [np.nan, np.inf, -np.inf, 1e30, -1e30]
    mids = ((g[:-1].astype(np.float64) + g[1:]) / 2).astype(dtype)[:1000]
    x[1000:1000 + mids.size] = mids
    want = np.zeros(nobs, dtype=dtype)
    lib = _lib.load()
    prev = lib.interpn_hip_set_fma(int(fma))
    try:
        oracle.linear_rectilinear([g], vals, [x], want, fma=fma)
        it = interpn_amd.Interpolator.rectilinear("linear", [g], vals, dtype=dtype)
        got = it.eval_host([x], np.zeros(nobs, dtype=dtype))
        name = it.kernel_name()
        it.close()
    finally:
        lib.interpn_hip_set_fma(prev)
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(got[~np.isnan(got)], want[~np.isnan(want)])
    if axis == "dense_pairs" and dtype == np.float64:
        assert name.startswith("interpn::k_linear_rectilinear<"), name
    elif axis != "dense_pairs":
        assert name.startswith("interpn::k_linear1_records<"), name


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
def test_cubic_4d_second_table_for_binned_batches(oracle, kind, dtype):
    """4-D multicubic grids whose in-place layout is not the fully overlapped one (20^4: an
    L2-friendly layout serves small batches) also keep the fully overlapped tile table; batches of
    524 288 points and more are counting-sorted and evaluated on it (LDS-DMA gather), smaller ones
    run in place on the first table — both must give the oracle's bits.
    multicubic/regular.rs:325-623, rectilinear.rs:265-545."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    axis = [20, 20, 20, 20] if dtype == np.float64 else [24, 24, 24, 24]
    case = synthetic_case("cubic", kind, 4, axis, 800_003, 9700, dtype, linearize=True, extrap=0.1, specials=True)
    want = run_oracle(oracle, case, True)
    it = _make_interp(interpn_amd, case)
    obs = [torch.from_numpy(o).to(dev) for o in case.obs]
    got = it.eval_tensors(obs).cpu().numpy()
    it.finish()
    assert np.array_equal(got, want)
    _, si, sj = it.table_layout()
    if kind == "regular" or dtype == np.float32:
        assert (si, sj) != (1, 1)                       # the in-place table is another layout ...
        assert it.get_option("last_binned") == 1        # ... and this batch ran sorted on the second one
        # the tiled kernel on the fully overlapped tiles, or (from one row of its lanes per bin on) the column kernel, which is filled from them
        assert it.kernel_name().endswith(", 1, 1>") or it.kernel_name().startswith("interpn::k_cubic_column<"), it.kernel_name()
    small = [o[:100_000] for o in obs]
    got_small = it.eval_tensors(small).cpu().numpy()
    it.finish()
    assert it.get_option("last_binned") == 0
    assert np.array_equal(got_small, want[:100_000])
    it.close()


def test_binned_evaluation_across_slices(oracle, monkeypatch):
    """Binned evaluation sorts and evaluates at most 2^25 points at a time (bounded scratch): a
    batch of 2^25 + 12 345 points spans two slices — the second one a ragged handful — and must
    equal the in-place evaluation bit for bit; 2e5 sampled points are checked against the oracle,
    and a NaN planted in the SECOND slice is reported with its batch-wide index."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    dev = torch.device("cuda:0")
    P = (1 << 25) + 12_345
    case = synthetic_case("cubic", "regular", 4, [6, 5, 7, 6], 200_000, 9800, np.float32, linearize=False, extrap=0.1, specials=False)
    it = _make_interp(interpn_amd, case)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    obs = [torch.rand(P, dtype=torch.float32, device=dev, generator=gen) * 2.2 - 1.1 for _ in range(4)]
    it.set_option("binned", 0)
    ref = it.eval_tensors(obs)
    it.finish()
    it.set_option("binned", 1)
    out = it.eval_tensors(obs)
    it.finish()
    assert it.get_option("last_binned") == 1
    assert bool(torch.equal(out, ref))
    idx = torch.randint(0, P, (200_000,), device=dev, generator=gen)
    idx[:1000] = torch.arange(P - 1000, P, device=dev)  # the tail slice
    sub = [o[idx].cpu().numpy() for o in obs]
    want = np.zeros(200_000, dtype=np.float32)
    oracle.cubic_regular(case.dims, case.starts, case.steps, case.vals, False, sub, want)
    assert np.array_equal(out[idx].cpu().numpy(), want)
    bad_at = (1 << 25) + 777
    obs[1][bad_at] = float("nan")
    it.eval_tensors(obs, out)
    with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as err:
        it.finish()
    assert err.value.first_bad_index == bad_at
    it.close()


def test_fma_flavour_is_a_per_handle_property(oracle):
    """The reference's `fma` cargo feature (Cargo.toml:34-38) is chosen per interpolator
    (INTERPN_HIP_FLAVOUR_* in `method`), not through process-global state: two handles of
    different flavour, evaluated concurrently from two threads on their own streams, each give
    their own flavour's bits; the process default (`interpn_hip_set_fma`) is never touched and
    flipping it does not change an existing handle; option "fma" reads and switches a handle."""
    import threading

    import torch

    import interpn_amd
    from interpn_amd import _lib

    dev = torch.device("cuda:0")
    results = {}
    for method, kind, axis in (("linear", "regular", [17, 9, 32]), ("cubic", "rectilinear", [9, 6, 16])):
        case = synthetic_case(method, kind, 3, axis, 300_007, 777, np.float64, extrap=0.2)
        want = {True: run_oracle(oracle, case, True), False: run_oracle(oracle, case, False)}
        assert not np.array_equal(want[True], want[False])  # the flavours differ on this workload
        mk = (lambda f: interpn_amd.Interpolator.regular(method, case.dims, case.starts, case.steps, case.vals, False, 0,
                                                         np.float64, fma=f)) if kind == "regular" else \
             (lambda f: interpn_amd.Interpolator.rectilinear(method, case.grids, case.vals, False, 0, np.float64, fma=f))
        its = {True: mk(True), False: mk(False)}
        assert its[True].get_option("fma") == 1 and its[False].get_option("fma") == 0
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        errors = []

        def work(flag):
            try:
                s = torch.cuda.Stream()
                for _ in range(25):
                    with torch.cuda.stream(s):
                        out = its[flag].eval_tensors(obs)
                    its[flag].finish()
                    if not np.array_equal(out.cpu().numpy(), want[flag]):
                        errors.append((method, flag))
                        return
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        ts = [threading.Thread(target=work, args=(f,)) for f in (True, False)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not errors, errors
        # the deprecated process default does not reach existing handles ...
        lib = _lib.load()
        prev = lib.interpn_hip_set_fma(0)
        try:
            out = its[True].eval_tensors(obs)
            its[True].finish()
            assert np.array_equal(out.cpu().numpy(), want[True])
            # ... only handles created afterwards without a flavour flag
            d = mk(None)
            assert d.get_option("fma") == 0
            d.close()
        finally:
            lib.interpn_hip_set_fma(prev)
        # option "fma" switches a handle between launches
        its[True].set_option("fma", 0)
        out = its[True].eval_tensors(obs)
        its[True].finish()
        assert np.array_equal(out.cpu().numpy(), want[False])
        for it in its.values():
            it.close()
        results[method] = True
    # both flags at once is an error
    import ctypes

    lib = _lib.load()
    h = ctypes.c_void_p()
    d, st, sp, v = (ctypes.c_size_t * 1)(4), (ctypes.c_double * 1)(0.0), (ctypes.c_double * 1)(1.0), (ctypes.c_double * 4)(0, 1, 2, 3)
    rc = lib.interpn_hip_create_regular_f64(_lib.FLAVOUR_FMA | _lib.FLAVOUR_NO_FMA, d, 1, st, 1, sp, 1, v, 4, 0, 0, 0, ctypes.byref(h))
    assert rc == _lib.ERR_INVALID_ARGUMENT and not h.value
    assert results == {"linear": True, "cubic": True}


def test_eval_device_reports_its_path_and_reserve_stops_allocation(oracle, monkeypatch):
    """`interpn_hip_eval_device_ex` says which path an evaluation took and why; after
    `interpn_hip_reserve(h, n, k)` evaluations of at most n points on at most k streams allocate
    nothing (also with INTERPN_HIP_EVAL_NO_ALLOC, which never allocates); two threads on two
    streams use two scratch blocks concurrently instead of one falling back to the in-place
    kernel; under graph capture the call evaluates in place and says so."""
    import threading

    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    dev = torch.device("cuda:0")
    P = 300_000
    case = synthetic_case("cubic", "regular", 4, [6, 7, 5, 6], P, 9400, np.float64, extrap=0.2, specials=False)
    want = run_oracle(oracle, case, True)
    it = _make_interp(interpn_amd, case)
    it.set_option("binned", 1)
    obs = [torch.from_numpy(o).to(dev) for o in case.obs]
    # nothing reserved, allocation forbidden: in place, and the reason is reported
    out = it.eval_tensors(obs, no_alloc=True)
    it.finish()
    assert it.last_path == "in_place" and "scratch" in it.last_path_reason
    assert it.get_option("scratch_allocs") == 0 and it.get_option("scratch_bytes") == 0
    assert np.array_equal(out.cpu().numpy(), want)
    it.reserve(P, 2)
    assert it.get_option("scratch_allocs") == 2
    bytes2 = it.get_option("scratch_bytes")
    assert bytes2 >= 2 * P * (4 * 8 + 4)
    out = it.eval_tensors(obs, no_alloc=True)
    it.finish()
    assert it.last_path == "binned" and it.last_path_reason == ""
    assert np.array_equal(out.cpu().numpy(), want)
    # two threads, two streams, one handle: both sorted, no allocation, right answers
    errors = []

    def work():
        try:
            s = torch.cuda.Stream()
            for _ in range(10):
                with torch.cuda.stream(s):
                    o = it.eval_tensors(obs, no_alloc=True)
                s.synchronize()
                if not np.array_equal(o.cpu().numpy(), want):
                    errors.append("mismatch")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    before = it.get_option("evals_binned")
    ts = [threading.Thread(target=work) for _ in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    it.finish()
    assert not errors, errors
    assert it.get_option("evals_binned") == before + 20  # none of them fell back
    assert it.get_option("scratch_allocs") == 2 and it.get_option("scratch_bytes") == bytes2
    # more streams than blocks: the extra ones wait (on the device) for the least recently used
    # block instead of allocating or falling back
    outs = []
    for s in [torch.cuda.Stream() for _ in range(5)]:
        with torch.cuda.stream(s):
            outs.append(it.eval_tensors(obs, no_alloc=True))
        assert it.last_path == "binned"
    torch.cuda.synchronize()
    it.finish()
    assert all(np.array_equal(o.cpu().numpy(), want) for o in outs)
    assert it.get_option("scratch_allocs") == 2 and it.get_option("scratch_bytes") == bytes2
    # a larger batch than reserved: allocation allowed -> grows; forbidden -> in place
    big = [torch.cat([o, o]) for o in obs]
    o2 = it.eval_tensors(big, no_alloc=True)
    it.finish()
    assert it.last_path == "in_place"
    assert np.array_equal(o2[:P].cpu().numpy(), want)
    o2 = it.eval_tensors(big)
    it.finish()
    assert it.last_path == "binned" and it.get_option("scratch_allocs") == 3
    assert np.array_equal(o2[P:].cpu().numpy(), want)
    # small batches of a handle in automatic mode, and captures, say why
    it.set_option("binned", 0)
    it.eval_tensors(obs)
    it.finish()
    assert it.last_path == "in_place" and "binned = 0" in it.last_path_reason
    it.set_option("binned", 1)
    res = torch.zeros(P, dtype=torch.float64, device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        it.eval_tensors(obs, res)
    assert it.last_path == "in_place" and "capture" in it.last_path_reason
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(res.cpu().numpy(), want)
    it.finish()
    it.close()
    # a handle that never sorts and never sweeps (2-D): nothing to reserve, path always in place, no reason
    lin = synthetic_case("linear", "regular", 2, [9, 8], 1000, 1, np.float64)
    it = _make_interp(interpn_amd, lin)
    it.reserve(10**8, 4)
    assert it.get_option("scratch_bytes") == 0
    it.eval_tensors([torch.from_numpy(o).to(dev) for o in lin.obs])
    it.finish()
    assert it.last_path == "in_place" and it.last_path_reason == ""
    it.close()
    # a 3-D multilinear handle: the sweep kernel's work words for batches that size (1.25 KiB per stream), nothing for small ones
    lin = synthetic_case("linear", "regular", 3, [9, 8, 7], 1000, 1, np.float64)
    it = _make_interp(interpn_amd, lin)
    it.reserve(10**6, 4)
    assert it.get_option("scratch_bytes") == 0
    it.reserve(10**8, 4)
    assert it.get_option("scratch_bytes") == 4 * 1280
    it.eval_tensors([torch.from_numpy(o).to(dev) for o in lin.obs])
    it.finish()
    assert it.last_path == "in_place"
    it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("axis", [[6, 7, 5, 6], [4, 4, 4, 4], [9, 5, 12, 7], [33, 32, 6, 5], [41, 40, 5, 6]], ids=str)
def test_column_evaluation_of_sorted_4d_multicubic(oracle, monkeypatch, dtype, axis, kind):
    """Column evaluation (cubic_column.h): large 4-D multicubic batches on a regular grid are sorted
    by the saturation-class pair of dims 0, 1 and a workgroup evaluates its bin's points out of an
    LDS-resident column of table tiles (cfg4's form).  Forced here at small sizes: batches from one
    point to several workgroup parts with ragged tails, ~30 % of the points extrapolating (every
    node form: interior, saturated low / high, linearised, mixed waves), NaN / inf / huge
    coordinates (first failing index = the ORIGINAL index), both `linearize_extrapolation` values,
    part sizes down to one row, persistent workgroups of 768 threads as two wave groups or one, of
    384 threads (one group) and of 256 (two groups of two waves), the column resident whole or one
    K-range at a time (`column_cpp`: 1..3 classes of dim 2 per phase, i.e. up to n2 - 1 phases per
    part, some of them empty), and — `bin_scramble` — every fifth point deliberately sorted into
    the wrong bin, so that the kernel's out-of-cell path (the same tree from the table in global
    memory) is exercised: always the oracle's bits.  [41, 40, 5, 6]: 1560 class-pair bins (more than
    one per thread of the sort's kernels).  Rectilinear grids (round 4): the sort and the kernel
    classify with the reference's own cell search, so no point is ever mis-binned by itself — the
    scramble option still forces the out-of-cell path; NaN / inf coordinates never fail there and
    propagate.  The sort's scatter runs in both forms (records staged in LDS in bin order and copied
    out linearly / stored directly; option `scatter_staged`).  Round 4, second session: dim 0 from per-part
    Hermite coefficients (`column_coef`, the default: parts whose dim-0 class is interior, or saturated
    without linearised extrapolation, rewrite their tile lines as y0, c1, c2, c3 once and evaluate 64 of a
    point's 85 nodes by Horner's three steps; points whose own arm is not the part's — mis-binned ones,
    extrapolating ones under `linearize` — come from the table in global memory) against every node from
    the table values (`column_coef` 0), LDS tiles padded or bare (`column_pad`: the local order then lives
    in the padding or behind the column), the bins' tail cut (`column_tail`), the local sort's keys taken
    from the records (10 bits) or from the upper eight bits of the index words, where the sort left them
    (`column_keys`, regular grids).
    src/multicubic/regular.rs:325-623, rectilinear.rs:265-545, mod.rs:72-117."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    dev = torch.device("cuda:0")
    want_t = torch.float64 if dtype == np.float64 else torch.float32
    for case_no, (nobs, threads, part, scramble, cpp) in enumerate(((1, 768, 0, 0, 0), (700, 384, 0, 0, 2), (5_000, 768, 0, 1, 1),
                                                                     (40_001, 256, 2048, 0, 3), (40_001, 768, 1, 1, 0), (250_013, 384, 0, 0, 1),
                                                                     (250_013, 768, 0, 0, 0), (250_014, 768, 3000, 1, 2))):
        lin = bool((nobs + threads) % 2)
        case = synthetic_case("cubic", kind, 4, axis, nobs, 9900 + sum(axis) + nobs, dtype, linearize=lin, extrap=0.3,
                              specials=min(axis) >= 8)
        want = run_oracle(oracle, case, True)
        it = _make_interp(interpn_amd, case)
        for k, v in (("binned", 1), ("column", 1), ("column_threads", threads), ("column_part", part), ("bin_scramble", scramble),
                     ("column_cpp", cpp), ("column_groups", 1 + nobs % 2), ("scatter_staged", (nobs // 7) % 2),
                     ("column_coef", 0 if case_no in (1, 4) else 1), ("column_pad", (-1, 0, 1)[case_no % 3]),
                     ("column_tail", (0x84, 0, 0x22)[(case_no + 1) % 3]), ("column_keys", (case_no // 2) % 2)):
            it.set_option(k, v)
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        out_full = torch.full((nobs + 2,), -5.0, dtype=want_t, device=dev)
        out = out_full[1:1 + nobs]
        it.eval_tensors(obs, out)
        it.finish()
        assert it.last_path == "binned" and it.kernel_name().startswith("interpn::k_cubic_column<"), it.kernel_name()
        got = out.cpu().numpy()
        same = (got == want) | (np.isnan(got) & np.isnan(want))
        assert np.all(same), (case_no, nobs, threads, part, scramble, cpp, int((~same).sum()))
        assert float(out_full[0]) == -5.0 and float(out_full[-1]) == -5.0  # nothing outside the batch was written
        if kind == "rectilinear" and nobs > 1000:
            # rectilinear grids never fail per point: NaN / inf go through the search (cell 0 / last) and propagate
            odd = [o.clone() for o in obs]
            odd[2][nobs // 2] = float("nan")
            odd[0][nobs // 3] = float("inf")
            sub = [o.cpu().numpy() for o in odd]
            want_odd = np.zeros(nobs, dtype=dtype)
            oracle.cubic_rectilinear(case.grids, case.vals, lin, sub, want_odd)
            res = it.eval_tensors(odd).cpu().numpy()
            it.finish()
            same = (res == want_odd) | (np.isnan(res) & np.isnan(want_odd))
            assert np.all(same), int((~same).sum())
        if kind == "regular" and nobs > 1000:
            # failing coordinates: the smallest ORIGINAL index is reported; the prefix is right
            bad = [o.clone() for o in obs]
            bad[2][nobs // 2] = float("nan")
            bad[0][nobs // 3] = float("inf")
            bad[3][nobs - 1] = 1e300 if dtype == np.float64 else 3e38
            res = it.eval_tensors(bad)
            with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as err:
                it.finish()
            assert err.value.first_bad_index == nobs // 3
            assert np.array_equal(res[:nobs // 3].cpu().numpy(), want[:nobs // 3])
        it.close()


def test_column_evaluation_is_chosen_for_large_batches_only(oracle, monkeypatch):
    """Automatic mode: the column form needs about one row of its workgroup's lanes (768 points) per
    bin to pay for its column fills (three, or one per K-range phase, where the column is not resident
    whole); below that the tiled kernel runs on the sorted points (and below 2^19 points the batch
    is not sorted at all).  Every path gives the same bits."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    dev = torch.device("cuda:0")
    case = synthetic_case("cubic", "regular", 4, [6, 7, 5, 6], 600_000, 9901, np.float64, extrap=0.1, specials=False)
    it = _make_interp(interpn_amd, case)  # (6-1)(7-1) = 30 bins: 600 000 points = 20 000 per bin
    obs = [torch.from_numpy(o).to(dev) for o in case.obs]
    it.set_option("binned", 1)
    a = it.eval_tensors(obs)
    it.finish()
    assert it.kernel_name().startswith("interpn::k_cubic_column<")
    small = [o[:15_000] for o in obs]   # 500 per bin: sorted, tiled kernel
    b = it.eval_tensors(small)
    it.finish()
    assert it.last_path == "binned" and it.kernel_name().startswith("interpn::k_cubic_brick<")
    mid = [o[:50_000] for o in obs]     # 1 667 per bin: the column kernel (one phase)
    m = it.eval_tensors(mid)
    it.finish()
    assert it.last_path == "binned" and it.kernel_name().startswith("interpn::k_cubic_column<")
    it.set_option("binned", 0)
    c = it.eval_tensors(obs)
    it.finish()
    assert it.last_path == "in_place"
    assert torch.equal(a, c) and torch.equal(b, c[:15_000]) and torch.equal(m, c[:50_000])
    it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("method,axis", [("linear", [300]), ("linear", [65, 130]), ("linear", [70, 9, 81]), ("linear", [7, 90, 5, 6]),
                                         ("linear", [40, 33]), ("linear", [500, 470]), ("linear", [512, 512]), ("linear", [128, 120, 100]),
                                         ("nearest", [80, 70, 90])], ids=str)
def test_rectilinear_bucket_records(oracle, monkeypatch, dtype, method, axis):
    """Axes too long for the lane-resident search are searched through per-bucket records — one LDS
    access {g[k-1], g[k], g[k+1], k} per cell query instead of table words + scan + brackets
    (interpn_device.h::axis_cell; multilinear/rectilinear.rs:353-370).  Jittered axes (records
    built), with and without the records (option axis_records), forced LDS search on short axes
    too, and a clustered axis whose buckets hold several coordinates (no records: the old search):
    always the oracle's bits, exact nodes / domain ends / NaN included.  Round 4: where the full
    records exceed the kernel's LDS budget (2-D 512^2 in f64: 64 KiB; 3-D 128 x 120 x 100) the compact
    form {g[k], k} + coordinates is built instead (option axis_rec_mode = 2), and the image a kernel
    stages without records is coordinates + tables only, whatever records exist (nearest: none)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n = len(axis)
    for variant in ("jittered", "clustered"):
        case = synthetic_case(method, "rectilinear", n, axis, 120_007, 9950 + sum(axis), dtype, extrap=0.1)
        if variant == "clustered":  # 5 coordinates squeezed into one bucket's width on the longest axis
            d = int(np.argmax(axis))
            g = case.grids[d].astype(np.float64).copy()
            w = (g[-1] - g[0]) / (2 * g.size)
            g[10:15] = g[10] + np.arange(5) * w / 8
            g[15:] = np.maximum(g[15:], g[14] + w / 4 * (1 + np.arange(g.size - 15)))
            g = g.astype(dtype)
            assert np.all(np.diff(g) > 0)
            case.grids[d] = g
        case.obs[0][5] = np.nan
        want = run_oracle(oracle, case, True)
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        for records, axis_regs in ((1, -1), (0, -1), (1, 0)):
            it = _make_interp(interpn_amd, case)
            mode = it.get_option("axis_rec_mode")
            table_words = sum(2 * a + 1 for a in axis)  # (M + 1) u32 per axis, M = 2n
            assert it.get_option("axis_image_bytes") <= sum(axis) * np.dtype(dtype).itemsize + 4 * table_words + 16 * 3 * n + 280 * n
            if method == "nearest" or variant == "clustered":
                assert mode == 0
            elif axis in ([512, 512], [128, 120, 100]) and dtype == np.float64:
                assert mode == 2 and it.get_option("axis_rec_bytes") <= (60 if n <= 2 else 20) * 1024
            elif max(axis) > 64:
                assert mode in (1, 2)
            it.set_option("axis_records", records)
            it.set_option("axis_regs", axis_regs)
            got = it.eval_tensors(obs).cpu().numpy()
            it.finish()
            same = (got == want) | (np.isnan(got) & np.isnan(want))
            assert np.all(same), (variant, records, axis_regs, int((~same).sum()))
            it.close()


def test_stage_timing_of_a_sorted_evaluation(oracle, monkeypatch):
    """`interpn_hip_stage_ms` (bench.py's per-stage split of cfg4): with option stage_timing a
    single-slice sorted evaluation records events between its launches; the four durations are
    positive and add up to about the whole evaluation; without the option (or before any sorted
    evaluation) the call reports INVALID_ARGUMENT; results are unchanged by the option."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    dev = torch.device("cuda:0")
    case = synthetic_case("cubic", "regular", 4, [6, 7, 5, 6], 400_000, 9960, np.float64, extrap=0.1, specials=False)
    want = run_oracle(oracle, case, True)
    it = _make_interp(interpn_amd, case)
    it.set_option("binned", 1)
    obs = [torch.from_numpy(o).to(dev) for o in case.obs]
    with pytest.raises(ValueError):
        it.stage_ms()
    out = it.eval_tensors(obs)
    it.finish()
    with pytest.raises(ValueError):  # sorted, but not timed
        it.stage_ms()
    it.set_option("stage_timing", 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out2 = it.eval_tensors(obs)
    b.record()
    it.finish()
    st = it.stage_ms()
    assert set(st) == {"hist", "scan", "scatter", "kernel"} and all(v > 0 for v in st.values())
    assert sum(st.values()) <= a.elapsed_time(b) * 1.05 + 0.05
    assert np.array_equal(out.cpu().numpy(), want) and np.array_equal(out2.cpu().numpy(), want)
    # several slices: no stage record
    it.set_option("bin_slice_log2", 16)
    it.eval_tensors(obs)
    it.finish()
    with pytest.raises(ValueError):
        it.stage_ms()
    it.close()


def _sweep_case(kind, axis, nobs, seed, extrap=0.1, specials=True, dtype=np.float64):
    return synthetic_case("linear", kind, 3, axis, nobs, seed, dtype, extrap=extrap, specials=specials)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("kind,axis,env", [("regular", [20, 17, 33], None), ("regular", [64, 9, 12], None),
                                           ("regular", [130, 6, 7], None),  # leading cell index >> 2 for the 64 bins
                                           ("rectilinear", [24, 11, 40], None), ("rectilinear", [64, 64, 5], None),
                                           ("rectilinear", [24, 11, 40], {"axis_regs": 1}),
                                           # axes longer than a wave: searched in the LDS-staged axis image (records, or
                                           # coordinates + tables), or through L1/L2 where it exceeds the sweep's 20 KiB
                                           ("rectilinear", [70, 33, 90], None), ("rectilinear", [70, 33, 90], {"axis_records": 0}),
                                           ("rectilinear", [900, 12, 9], None), ("rectilinear", [33, 30, 31], {"axis_regs": 0})],
                         ids=["reg", "reg64", "reg130", "rect", "rect64", "rect_probe_sequence", "rect_long", "rect_long_tables",
                              "rect_900", "rect_lds_forced"])
def test_sweep_evaluation(oracle, kind, axis, env, fma, dtype):
    """The sweep evaluation of 3-D multilinear batches (linear_sweep.h: every wave sorts 1024 (f64 regular: 12 rows
    in registers + 4 parked in LDS) / 896 (f64 rectilinear: 12 + 2) / 1536 (f32; 1280 on rectilinear grids) points by leading cell index on chip and walks its rows in
    step with a clock; f32 on its 2 x 4 x 4 bricks) against the oracle
    and, bit for bit, against the brick kernel: batches of one point, of one round less / plus one
    point, of many ragged rounds; extrapolated and special points; with the clock (measured period,
    a fixed one) and without; both cargo flavours (multilinear/regular.rs:296-404, rectilinear.rs:244-370)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    case = _sweep_case(kind, axis, 300_007, 900 + sum(axis), dtype=dtype)
    want = run_oracle(oracle, case, fma)
    tname = "double" if dtype == np.float64 else "float"
    if kind == "regular":
        it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals, fma=fma)
    else:
        it = interpn_amd.Interpolator.rectilinear("linear", case.grids, case.vals, fma=fma)
    try:
        for k, v in (env or {}).items():
            it.set_option(k, v)
        assert it.get_option("sweep_table_bytes") > 0 and it.get_option("sweep_layout") in (11, 12)
        full = [torch.from_numpy(o).to(dev) for o in case.obs]
        for count, period in ((1, 0), (767, 0), (768, 1), (769, 1500), (895, 0), (896, 1), (897, 0), (1023, 0), (1024, 1), (1025, 0), (1279, 0), (1280, 0), (1281, 1), (1535, 0), (1537, 0), (100_003, 0),
                              (300_007, 1), (300_007, 0), (300_007, 0), (300_007, 700)):
            obs = [t[:count].clone() for t in full]
            it.set_option("sweep", 1)
            it.set_option("sweep_period", period)
            got = it.eval_tensors(obs)
            assert it.last_path == "sweep", (it.last_path, it.last_path_reason)
            assert it.kernel_name().startswith(f"interpn::k_linear_sweep<{tname}, " + ("true" if kind == "rectilinear" else "false")), it.kernel_name()
            it.finish()
            it.set_option("sweep", 0)
            ref = it.eval_tensors(obs)
            assert it.last_path == "in_place" and it.kernel_name().startswith("interpn::k_linear_brick"), it.kernel_name()
            it.finish()
            assert torch.equal(got, ref), (count, period)
            w = want[:count]
            g = got.cpu().numpy()
            same = (g == w) | (np.isnan(g) & np.isnan(w))
            assert np.all(same), (count, period, int((~same).sum()))
        assert it.get_option("evals_sweep") == 20
    finally:
        it.close()


@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("steps,starts", [((2.0 / 19, 2.0 / 16, 2.0 / 32), (-1.0, -1.0, -1.0)),
                                          ((0.1, 0.3, 1.0 / 3.0), (-0.0, 0.7, -5.0)),          # steps whose quotients round up to integers; a negative-zero start
                                          ((1e-30, 3e30, 7.0), (1e-25, -4e31, 2.0)),           # inside [2^-128, 2^128]: the short forms
                                          ((2.0 ** -128, 2.0 ** 128, 1.0), (0.0, 0.0, 0.0)),    # the ends of that range
                                          ((1e-40, 1.0, 1.0), (0.0, -3.0, 4.0)),                # one step outside it: the divide sequences for every row
                                          ((1.0, 1e60, 5e-324 * 2 ** 60), (0.0, 0.0, 0.0))],
                         ids=["linspace", "thirds", "wide", "range_ends", "tiny_step", "huge_and_subnormalish"])
def test_sweep_cell_index_and_t_without_divisions(oracle, steps, starts, fma):
    _step_cell_case(oracle, steps, starts, fma, np.float64)


@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("steps,starts", [((2.0 / 19, 2.0 / 16, 2.0 / 32), (-1.0, -1.0, -1.0)),
                                          ((0.1, 0.3, 1.0 / 3.0), (-0.0, 0.7, -5.0)),
                                          ((2.0 ** -16, 2.0 ** 16, 7.0), (0.0, 0.0, 2.0)),       # the ends of the f32 range of steps
                                          ((1e-6, 1.0, 1e6), (0.0, -3.0, 4.0))],               # outside it: the divide sequences for every row
                         ids=["linspace", "thirds", "range_ends", "outside"])
def test_sweep_cell_index_and_t_without_divisions_f32(oracle, steps, starts, fma):
    _step_cell_case(oracle, steps, starts, fma, np.float32)


def _step_cell_case(oracle, steps, starts, fma, dtype):
    """interpn_device.h::step_cell_fast (the sweep kernel's cell index and normalized coordinate on regular f64
    grids from the step's reciprocal and fma remainders, no divide sequence) against the oracle's divisions,
    bit for bit, on the points that sit on or next to every condition of the short forms: grid planes and
    their floating-point neighbours (quotients that are integers or within a few ulp of one), points 2^-21 ..
    2^-19 of a cell away from a plane (either side of the near-integer threshold), x - izl of 0, of subnormal
    size and beyond 2^256, +-0, infinities, NaN, far extrapolation up to where the reference reports an
    unrepresentable coordinate; steps at and beyond the ends of the range the host admits
    (multilinear/regular.rs:334-339, :415-422).  f32: the same with its own ranges."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(77)
    dims = [20, 17, 33]
    nobs = 40_000
    obs = []
    f32 = dtype == np.float32
    steps = tuple(float(dtype(v)) for v in steps)
    starts = tuple(float(dtype(v)) for v in starts)
    for d in range(3):
        st, s0, n = steps[d], starts[d], dims[d]
        k = rng.integers(-2, n + 2, nobs).astype(np.float64)
        frac = rng.random(nobs)
        x = s0 + st * (k + frac)
        sel = rng.integers(0, 12, nobs)
        plane = (s0 + st * k).astype(dtype).astype(np.float64)
        x = np.where(sel == 0, plane, x)
        x = np.where(sel == 1, np.nextafter(plane.astype(dtype), dtype(np.inf)).astype(np.float64), x)
        x = np.where(sel == 2, np.nextafter(plane.astype(dtype), dtype(-np.inf)).astype(np.float64), x)
        for e, code in (((-13, 3), (-11, 4), (-9, 5)) if f32 else ((-21, 3), (-20, 4), (-19, 5))):
            x = np.where(sel == code, s0 + st * (k + rng.choice([-1.0, 1.0], nobs) * 2.0 ** e * (1 + 0.5 * frac)), x)
        x = np.where(sel == 6, plane + st * 2.0 ** -300 * frac, x)       # x - izl far below 2^-256 steps (or absorbed: = the plane)
        x = np.where(sel == 7, s0 + st * (k + frac) * 2.0 ** rng.integers(20, 70, nobs), x)  # far outside, up to |floc| ~ 2^75: unrepresentable beyond 2^63
        specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324, 1e300, -1e300, 2.0 ** 31 * st + s0, -(2.0 ** 31) * st + s0,
                             2.0 ** 63 * st + s0, 2.0 ** 62 * st + s0, -(2.0 ** 63) * st + s0, np.nextafter(-(2.0 ** 63) * st + s0, -np.inf)])
        x = np.where(sel == 8, specials[rng.integers(0, specials.size, nobs)], x)
        with np.errstate(over="ignore", invalid="ignore"):
            obs.append(np.ascontiguousarray(x.astype(dtype)))
    # (an unrepresentable coordinate aborts the reference at its first such point: check the prefix the reference
    #  writes AND, by cutting the batch there, that every later stretch evaluates bit for bit as well)
    vals = rng.uniform(-1.0, 1.0, int(np.prod(dims))).astype(dtype)
    it = interpn_amd.Interpolator.regular("linear", dims, np.array(starts, dtype), np.array(steps, dtype), vals, fma=fma)
    try:
        it.set_option("sweep", 1)
        lo = 0
        stretches = 0
        while lo < nobs and stretches < 400:
            o = [a[lo:] for a in obs]
            want, status = np.full(o[0].size, -7.0, dtype), None
            try:
                oracle.linear_regular(dims, np.array(starts, dtype), np.array(steps, dtype), vals, o, want, fma=fma)
            except AssertionError as e:  # pyoracle.OracleError: "Unrepresentable coordinate value" at e.first_bad
                status = e.first_bad
                assert status is not None
            got = it.eval_tensors([torch.from_numpy(a).to(dev) for a in o])
            assert it.last_path == "sweep", (it.last_path, it.last_path_reason)
            bad = None
            try:
                it.finish()
            except AssertionError as e:
                bad = getattr(e, "first_bad_index", None)
                assert bad is not None
            g = got.cpu().numpy()
            upto = o[0].size if status is None else status
            assert (bad is None) == (status is None) and (bad is None or bad == status), (lo, bad, status)
            same = (g[:upto] == want[:upto]) | (np.isnan(g[:upto]) & np.isnan(want[:upto]))
            assert np.all(same), (lo, int((~same).sum()), np.flatnonzero(~same)[:5], [a[np.flatnonzero(~same)[:3]] for a in o])
            stretches += 1
            if status is None:
                break
            lo += status + 1
        assert stretches >= 1
    finally:
        it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("axis,layout", [([9, 12], None), ([9, 12], "11"), ([12, 9, 14], None), ([12, 9, 14], "11"), ([12, 9, 14], "24"), ([21, 38, 7, 22], "22"),
                                         ([8, 9, 7, 10], "11"), ([6, 7, 5, 6, 4], None), ([5, 4, 6, 4, 5, 4], None)], ids=str)
def test_regular_cubic_boundary_slope_where_two_dy_overflows(oracle, monkeypatch, axis, layout, fma, dtype):
    """The saturated classes' second slope is `two.mul_add(dy, -k0)` under the reference's `fma` feature and `two * dy - k0`
    without it (multicubic/regular.rs:525-528, :546-549, :580-583, :602-605; the recursive arm, N >= 5, the same except
    OutsideLow, which stays unfused: regular_recursive.rs:536).  The product is exact, so the forms agree until 2 dy overflows:
    grid values near the top of the type's range and points far outside the grid (cubic extrapolation) reach that, and the
    kernels must then return the infinities and NaNs the reference's form gives.  Every regular multicubic kernel family: the
    C-order kernels, the tiled ones, the binned / column evaluation, the sweep kernel, the runtime-N kernels of both arms."""
    import torch

    import interpn_amd

    if layout:
        monkeypatch.setenv("INTERPN_HIP_BRICKS", layout)
    dev = torch.device("cuda:0")
    n = len(axis)
    nobs = 30_011 if n <= 4 else 4_001
    case = synthetic_case("cubic", "regular", n, axis, nobs, 600 + sum(axis), dtype, linearize=False, extrap=2.0, specials=n <= 4 and min(axis) >= 8)
    rng = np.random.default_rng(9 + n)
    v = case.vals.astype(np.float64)
    k = int(rng.integers(0, v.size // 2))
    v[k:k + v.size // 3] *= 1e307 if dtype == np.float64 else 1e37
    with np.errstate(all="ignore"):
        case.vals = v.astype(dtype)
        want = run_oracle(oracle, case, fma)
    assert np.isinf(want).sum() + np.isnan(want).sum() > 0 and np.isfinite(want).sum() > 0
    it = interpn_amd.Interpolator.regular("cubic", case.dims, case.starts, case.steps, case.vals, linearize_extrapolation=False, fma=fma)
    try:
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        modes = [("binned", 0), ("binned", 1)] if (layout and n <= 4) else [("binned", 0)]
        if n in (2, 3) and layout == "11":
            modes.append(("sweep", 1))
        for opt, val in modes:
            it.set_option("binned", 0)
            it.set_option("sweep", 0)
            it.set_option(opt, val)
            got = it.eval_tensors(obs).cpu().numpy()
            name = it.kernel_name()
            it.finish()
            same = (got == want) | (np.isnan(got) & np.isnan(want))
            assert np.all(same), (opt, val, name, int((~same).sum()), [(float(got[i]), float(want[i])) for i in np.flatnonzero(~same)[:3]])
    finally:
        it.close()


def _awkward_values(vals, n, dtype, tiny, huge, rng):
    sl = lambda lo, hi: tuple(slice(lo, hi) for _ in range(n))
    vals[sl(0, 4)] = dtype(0.375)                       # equal neighbours
    z = np.zeros_like(vals[sl(4, 8)])
    z[rng.random(z.shape) < 0.5] = -0.0
    vals[sl(4, 8)] = z                                  # zeros of both signs
    vals[sl(2, 6)][..., 0] *= dtype(tiny)               # tiny beside ordinary values
    vals[tuple(slice(6, 8) for _ in range(n - 1)) + (slice(None),)] *= dtype(huge)
    flat = vals.reshape(-1)
    flat[rng.integers(0, flat.size, 3)] = np.inf
    flat[rng.integers(0, flat.size, 2)] = np.nan
    flat[rng.integers(0, flat.size, 2)] = -np.inf
    return flat


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("axis,ratios", [([40, 36], "plain"), ([40, 36], "extreme"), ([40, 36], "quantized"), ([20, 17, 33], "plain"), ([20, 17, 33], "extreme"),
                                         ([20, 17, 33], "quantized"), ([9, 8, 10, 11], "plain"), ([9, 8, 10, 11], "quantized")],
                         ids=["2d", "2d_extreme_ratios", "2d_quantized", "3d", "3d_extreme_ratios", "3d_quantized", "4d", "4d_quantized"])
@pytest.mark.parametrize("records", ["64", "0"], ids=["cell_records", "no_records"])
def test_rectilinear_cubic_short_divisions_on_awkward_values(oracle, monkeypatch, records, axis, ratios, fma, dtype):
    """The rectilinear multicubic node on the fully overlapped tile table takes its two spacing-ratio divisions per node as a
    reciprocal-and-correction sequence where the operands allow it and evaluates the wave again with the divide sequences
    where they do not (interpn_device.h::cubic_rect_node_fast; multicubic/rectilinear.rs:413-545, mod.rs:103-117).  Here the
    grid values and spacings are what that form must hand back: equal neighbours (differences +0), zeros of both signs, blocks
    of tiny and of huge magnitude (differences outside the admitted exponent window, products that overflow), infinities and
    NaN among the values, and axes whose neighbouring spacings differ by more than the admitted ratio; "quantized": values on
    a lattice of 1/4, so that every wave meets differences that are exactly +0 and none that the short form refuses (the short
    form's own results are what is compared).  With the per-cell records of the axes (cubic_cell_record.h: the seven divisions
    of a dimension's setup done once at creation, t from the record's reciprocal) and without them.  In place and through
    the sweep kernel, bit for bit against the oracle."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    monkeypatch.setenv("INTERPN_HIP_CUBIC_RECORDS", records)
    dev = torch.device("cuda:0")
    n = len(axis)
    f64 = dtype == np.float64
    case = synthetic_case("cubic", "rectilinear", n, axis, 120_003, 4400 + sum(axis), dtype, linearize=bool(n % 2), extrap=0.2, specials=True)
    rng = np.random.default_rng(77 + n)
    if ratios == "extreme":
        for d in range(n):
            g = case.grids[d].astype(np.float64)
            k = 3 + 2 * d
            gap = (1e-160 if f64 else 1e-7) * max(1.0, abs(g[k]))  # f64: h ratio 1e158 > 2^128 (the sum g + gap must still differ from g)
            if f64:
                g = g - g[k]  # the pair straddles zero so that the tiny spacing is representable
                g[k + 1] = gap
            else:
                g[k + 1] = np.float32(g[k]) + np.float32(max(gap, np.spacing(np.float32(g[k]))))
                g[-1] = g[-2] + 3e4  # spacing ratio 3e4 / 0.03 > 2^16
            g = g.astype(dtype)
            assert np.all(np.diff(g) > 0)
            case.grids[d] = g
    if ratios == "quantized":
        case.vals = (np.round(case.vals.astype(np.float64) * 4) / 4 + 0.0).astype(dtype)  # (+ 0.0: no negative zeros)
    vals = case.vals.reshape(axis).copy()
    tiny, huge = (1e-300, 1e200) if f64 else (1e-36, 1e30)
    if ratios != "quantized":
        vals = _awkward_values(vals, n, dtype, tiny, huge, rng)
    case.vals = vals.reshape(-1).astype(dtype)
    with np.errstate(all="ignore"):
        want = run_oracle(oracle, case, fma)
    it = interpn_amd.Interpolator.rectilinear("cubic", case.grids, case.vals, linearize_extrapolation=case.linearize, fma=fma)
    try:
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        modes = (0, 1) if n in (2, 3) else (0,)
        for sweep in modes:
            it.set_option("sweep", sweep)
            got = it.eval_tensors(obs).cpu().numpy()
            name = it.kernel_name()
            it.finish()
            if sweep == 0:
                assert it.last_path == "in_place" and name.startswith("interpn::k_cubic_brick<") and name.endswith("1, 1>"), name
            elif it.last_path != "sweep":
                continue  # (a handle without the table the sweep kernel reads)
            same = (got == want) & (np.signbit(got) == np.signbit(want)) | (np.isnan(got) & np.isnan(want))
            assert np.all(same), (sweep, name, int((~same).sum()), np.flatnonzero(~same)[:5])
        assert np.isfinite(want).all() if ratios == "quantized" else np.isfinite(want).sum() > want.size // 2
    finally:
        it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("linearize", [False, True], ids=["cubic_extrap", "linearized"])
@pytest.mark.parametrize("kind,axis", [("regular", [150, 140]), ("regular", [300, 90]), ("rectilinear", [130, 160])], ids=["reg", "reg_long_dim0", "rect"])
def test_cubic2_sweep_evaluation(oracle, kind, axis, linearize, fma, dtype):
    """The same for 2-D multicubic batches (cubic_sweep.h with N = 2: one plane of tiles, the points of a wave ordered by their
    dim-0 cell, the slowest dimension of the tile table): against the oracle and, bit for bit, against the tiled kernel in
    place, ragged rounds, special and extrapolated points, the first failing index
    (multicubic/regular.rs:297-623, rectilinear.rs:237-545)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    case = synthetic_case("cubic", kind, 2, axis, 200_003, 1300 + sum(axis), dtype, linearize=linearize, extrap=0.25, specials=True)
    want = run_oracle(oracle, case, fma)
    tname = "double" if dtype == np.float64 else "float"
    if kind == "regular":
        it = interpn_amd.Interpolator.regular("cubic", case.dims, case.starts, case.steps, case.vals, linearize_extrapolation=linearize, fma=fma)
    else:
        it = interpn_amd.Interpolator.rectilinear("cubic", case.grids, case.vals, linearize_extrapolation=linearize, fma=fma)
    try:
        full = [torch.from_numpy(o).to(dev) for o in case.obs]
        rows = 14 if dtype == np.float64 else 28
        for count, period in ((1, 0), (rows * 64 - 1, 0), (rows * 64, 1), (rows * 64 + 1, 1500), (100_003, 0), (200_003, 1), (200_003, 0), (200_003, 2500)):
            obs = [t[:count].clone() for t in full]
            it.set_option("sweep", 1)
            it.set_option("sweep_period", period)
            got = it.eval_tensors(obs)
            assert it.last_path == "sweep", (it.last_path, it.last_path_reason)
            assert it.kernel_name().startswith(f"interpn::k_cubic_sweep<{tname}, " + ("true" if kind == "rectilinear" else "false")), it.kernel_name()
            assert it.kernel_name().endswith(", 2>"), it.kernel_name()
            it.finish()
            it.set_option("sweep", 0)
            ref = it.eval_tensors(obs)
            assert it.last_path == "in_place" and it.kernel_name().startswith("interpn::k_cubic_"), it.kernel_name()
            it.finish()
            g, r = got.cpu().numpy(), ref.cpu().numpy()
            assert np.all((g == r) | (np.isnan(g) & np.isnan(r))), (count, period)
            w = want[:count]
            same = (g == w) | (np.isnan(g) & np.isnan(w))
            assert np.all(same), (count, period, int((~same).sum()))
        if kind == "regular":
            bad = [t.clone() for t in full]
            bad[1][150_000] = float("nan")
            bad[0][60_001] = float("inf")
            bad[1][60_002] = float("nan")
            it.set_option("sweep", 1)
            out = it.eval_tensors(bad)
            assert it.last_path == "sweep"
            with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
                it.finish()
            assert ei.value.first_bad_index == 60_001
            assert np.array_equal(out.cpu().numpy()[:60_001], want[:60_001])
        it.set_option("sweep", -1)
        it.eval_tensors(full)
        assert it.last_path == "in_place"
        it.finish()
    finally:
        it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("linearize", [False, True], ids=["cubic_extrap", "linearized"])
@pytest.mark.parametrize("kind,axis", [("regular", [20, 17, 33]), ("regular", [9, 8, 150]),  # dim-2 cell index >> 2 for the 64 bins
                                       ("regular", [64, 9, 12]), ("rectilinear", [24, 11, 40]), ("rectilinear", [8, 70, 90])],
                         ids=["reg", "reg_long_dim2", "reg_flat", "rect", "rect_long"])
def test_cubic_sweep_evaluation(oracle, kind, axis, linearize, fma, dtype):
    """The sweep evaluation of 3-D multicubic batches (cubic_sweep.h: every wave sorts 512 (f64) /
    1280 (f32) points by their dim-2 cell on chip and walks its rows in step with a clock, rows = cubic_brick.h's on the fully
    overlapped tile table, regular grids without divide sequences) against the oracle and, bit for bit, against the tiled
    kernel in place: batches of one point, of a round less / plus one point, of many ragged rounds; extrapolated (both
    `linearize_extrapolation` values) and special points; with the clock and without; both cargo flavours
    (multicubic/regular.rs:297-623, rectilinear.rs:237-545)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    case = synthetic_case("cubic", kind, 3, axis, 200_003, 1200 + sum(axis), dtype, linearize=linearize, extrap=0.25, specials=True)
    want = run_oracle(oracle, case, fma)
    tname = "double" if dtype == np.float64 else "float"
    if kind == "regular":
        it = interpn_amd.Interpolator.regular("cubic", case.dims, case.starts, case.steps, case.vals, linearize_extrapolation=linearize, fma=fma)
    else:
        it = interpn_amd.Interpolator.rectilinear("cubic", case.grids, case.vals, linearize_extrapolation=linearize, fma=fma)
    try:
        full = [torch.from_numpy(o).to(dev) for o in case.obs]
        for count, period in ((1, 0), (511, 0), (512, 1), (513, 0), (639, 0), (640, 1), (641, 1500), (1279, 0), (1280, 0), (1281, 1), (100_003, 0),
                              (200_003, 1), (200_003, 0), (200_003, 0), (200_003, 2500)):
            obs = [t[:count].clone() for t in full]
            it.set_option("sweep", 1)
            it.set_option("sweep_period", period)
            got = it.eval_tensors(obs)
            assert it.last_path == "sweep", (it.last_path, it.last_path_reason)
            assert it.kernel_name().startswith(f"interpn::k_cubic_sweep<{tname}, " + ("true" if kind == "rectilinear" else "false")), it.kernel_name()
            it.finish()
            it.set_option("sweep", 0)
            ref = it.eval_tensors(obs)
            assert it.last_path == "in_place" and it.kernel_name().startswith("interpn::k_cubic_"), it.kernel_name()
            it.finish()
            g, r = got.cpu().numpy(), ref.cpu().numpy()
            assert np.all((g == r) | (np.isnan(g) & np.isnan(r))), (count, period)
            w = want[:count]
            same = (g == w) | (np.isnan(g) & np.isnan(w))
            assert np.all(same), (count, period, int((~same).sum()))
        # an unrepresentable coordinate: the first failing index, the prefix in front of it (regular grids; rectilinear ones never fail)
        if kind == "regular":
            bad = [t.clone() for t in full]
            bad[2][150_000] = float("nan")
            bad[0][60_001] = float("inf")
            bad[1][60_002] = float("nan")
            it.set_option("sweep", 1)
            out = it.eval_tensors(bad)
            assert it.last_path == "sweep"
            with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
                it.finish()
            assert ei.value.first_bad_index == 60_001
            assert np.array_equal(out.cpu().numpy()[:60_001], want[:60_001])
        # automatic mode: small batches stay in place
        it.set_option("sweep", -1)
        it.eval_tensors(full)
        assert it.last_path == "in_place"
        it.finish()
    finally:
        it.close()


def test_division_free_forms_against_the_divide_sequences():
    """interpn_device.h::divide_fast / floor_quotient_fast on the GPU against the hardware's own IEEE divide sequences
    (tools/ablate_linear3d.hip::ablate_division_selftest): 2^30 (a, b) pairs per type over the admitted exponent ranges,
    a quarter of them with quotients next to integers — every quotient bit for bit, every floor that the short form
    reports as exact (~8e7 of the f64 pairs, more in f32).  (The arithmetic itself is pinned on the CPU with exact rationals: tests/test_step_cell_cpu.py.)"""
    import ctypes
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "libinterpn_ablate.so")
    if not os.path.exists(path):
        pytest.skip("tools/libinterpn_ablate.so not built")
    lib = ctypes.CDLL(path)
    lib.ablate_division_selftest.argtypes = [ctypes.c_int, ctypes.c_ulonglong, ctypes.c_uint, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
    for f32 in (0, 1):
        bad, floors = ctypes.c_ulonglong(1), ctypes.c_ulonglong(0)
        rc = lib.ablate_division_selftest(f32, 0x5EED0000 + f32, 1024, ctypes.byref(bad), ctypes.byref(floors))
        assert rc == 0, rc
        assert bad.value == 0, (f32, bad.value)
        assert floors.value > 2**25, (f32, floors.value)  # the pairs whose quotient lies below 2^31 (f32: 2^20) and away from the integers


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("axis", [[70, 90], [200, 40], [40, 300]], ids=["70x90", "long_dim0", "long_dim1"])  # leading cell index >> 2 for the 64 bins; many bricks per row
def test_linear2_sweep_evaluation(oracle, axis, fma, dtype):
    """The sweep evaluation of 2-D multilinear batches on regular grids (k_linear2_brick.hip::k_linear2_sweep: every wave sorts
    1280 (f64: 16 rows in registers + 4 parked in LDS) / 2048 (f32) points by leading cell index on chip and walks its rows in
    step with a clock; rows = the 2-D brick kernel's lane-pair gather, cell index and t without divide sequences) against the
    oracle and, bit for bit, against the brick kernel: batches of one point, of a round less / plus one point, of many ragged
    rounds; extrapolated and special points (grid lines, +-0, infinities, NaN: the first-failing-index contract); with the
    clock and without; both cargo flavours (multilinear/regular.rs:296-404)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    case = synthetic_case("linear", "regular", 2, axis, 250_007, 1500 + sum(axis), dtype, extrap=0.2, specials=True)
    want = run_oracle(oracle, case, fma)
    tname = "double" if dtype == np.float64 else "float"
    it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals, fma=fma)
    try:
        full = [torch.from_numpy(o).to(dev) for o in case.obs]
        for count, period in ((1, 0), (1279, 0), (1280, 1), (1281, 0), (2047, 0), (2048, 1), (2049, 1500), (100_003, 0), (250_007, 1), (250_007, 0),
                              (250_007, 0), (250_007, 900)):
            obs = [t[:count].clone() for t in full]
            it.set_option("sweep", 1)
            it.set_option("sweep_period", period)
            got = it.eval_tensors(obs)
            assert it.last_path == "sweep", (it.last_path, it.last_path_reason)
            assert it.kernel_name().startswith(f"interpn::k_linear2_sweep<{tname}, "), it.kernel_name()
            it.finish()
            it.set_option("sweep", 0)
            ref = it.eval_tensors(obs)
            assert it.last_path == "in_place" and it.kernel_name().startswith("interpn::k_linear2_brick"), it.kernel_name()
            it.finish()
            g, r = got.cpu().numpy(), ref.cpu().numpy()
            assert np.all((g == r) | (np.isnan(g) & np.isnan(r))), (count, period)
            w = want[:count]
            same = (g == w) | (np.isnan(g) & np.isnan(w))
            assert np.all(same), (count, period, int((~same).sum()))
        bad = [t.clone() for t in full]
        bad[1][200_000] = float("nan")
        bad[0][70_001] = float("inf")
        bad[1][70_002] = float("nan")
        it.set_option("sweep", 1)
        out = it.eval_tensors(bad)
        assert it.last_path == "sweep"
        with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
            it.finish()
        assert ei.value.first_bad_index == 70_001
        assert np.array_equal(out.cpu().numpy()[:70_001], want[:70_001])
        it.set_option("sweep", -1)
        it.eval_tensors(full)
        assert it.last_path == "in_place"  # (a small batch)
        it.finish()
    finally:
        it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("fma", [True, False], ids=["fma", "nofma"])
@pytest.mark.parametrize("axis", [[20, 17, 33], [150, 6, 7], [70, 90], [300, 9]], ids=["3d", "3d_long_dim0", "2d", "2d_long_dim0"])
def test_nearest_sweep_evaluation(oracle, axis, fma, dtype):
    """The sweep evaluation of 2-D / 3-D nearest-neighbour batches on regular grids (k_nearest.hip::k_nearest_sweep: the
    multilinear sweep kernel's scaffold and division-free index stage, one gather per point) against the oracle and, bit for
    bit, against the one-pass kernel: ragged rounds, extrapolated and special points (dt exactly 0.5, grid planes, +-0,
    infinities, NaN: the first-failing-index contract), with the clock and without; both cargo flavours
    (nearest/regular.rs:234-317)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n = len(axis)
    case = synthetic_case("nearest", "regular", n, axis, 250_007, 1700 + sum(axis), dtype, extrap=0.2, specials=True)
    # points exactly half way between two nodes (dt == 0.5 where the arithmetic is exact) and their neighbours
    for d in range(n):
        mid = (case.starts[d] + case.steps[d] * (np.arange(40) % (axis[d] - 1) + 0.5)).astype(dtype)
        case.obs[d][1000:1040] = mid
        case.obs[d][1040:1080] = np.nextafter(mid, dtype(np.inf))
        case.obs[d][1080:1120] = np.nextafter(mid, dtype(-np.inf))
    want = run_oracle(oracle, case, fma)
    tname = "double" if dtype == np.float64 else "float"
    it = interpn_amd.Interpolator.regular("nearest", case.dims, case.starts, case.steps, case.vals, fma=fma)
    try:
        full = [torch.from_numpy(o).to(dev) for o in case.obs]
        for count, period in ((1, 0), (1023, 0), (1024, 1), (1025, 0), (1279, 0), (1281, 1), (1535, 0), (1537, 0), (2049, 1500), (100_003, 0),
                              (250_007, 1), (250_007, 0), (250_007, 0), (250_007, 900)):
            obs = [t[:count].clone() for t in full]
            it.set_option("sweep", 1)
            it.set_option("sweep_period", period)
            got = it.eval_tensors(obs)
            assert it.last_path == "sweep", (it.last_path, it.last_path_reason)
            assert it.kernel_name().startswith(f"interpn::k_nearest_sweep<{tname}, {n}, "), it.kernel_name()
            it.finish()
            it.set_option("sweep", 0)
            ref = it.eval_tensors(obs)
            assert it.last_path == "in_place" and it.kernel_name().startswith("interpn::k_nearest<"), it.kernel_name()
            it.finish()
            g, r = got.cpu().numpy(), ref.cpu().numpy()
            assert np.array_equal(g, r), (count, period)
            assert np.array_equal(g, want[:count]), (count, period)
        bad = [t.clone() for t in full]
        bad[-1][200_000] = float("nan")
        bad[0][70_001] = float("inf")
        it.set_option("sweep", 1)
        out = it.eval_tensors(bad)
        assert it.last_path == "sweep"
        with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
            it.finish()
        assert ei.value.first_bad_index == 70_001
        assert np.array_equal(out.cpu().numpy()[:70_001], want[:70_001])
        it.set_option("sweep", -1)
        it.eval_tensors(full)
        assert it.last_path == "in_place"  # (a small batch)
        it.finish()
    finally:
        it.close()


@pytest.mark.parametrize("method,dims,count", [("cubic", [64, 64, 64], 10_000_000), ("linear", [1000, 1000], 12_000_000), ("nearest", [128, 128, 128], 20_000_000),
                                               ("nearest", [1200, 1000], 16_000_000), ("cubic", [512, 512], 30_000_000)],
                         ids=["cubic3", "linear2", "nearest3", "nearest2", "cubic2"])
def test_sweep_family_automatic_paths_full_size(oracle, method, dims, count):
    """The sweep kernels of 3-D multicubic, 2-D multilinear and nearest-neighbour at sizes the automatic rules take them by
    themselves: the path is reported as the sweep, the whole batch equals the one-pass kernel's bit for bit, 1e5 sampled points
    equal the oracle."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n = len(dims)
    rng = np.random.default_rng(91)
    starts = np.full(n, -1.0)
    steps = np.array([2.0 / (d - 1) for d in dims])
    vals = rng.uniform(-1, 1, int(np.prod(dims)))
    gen = torch.Generator(device=dev)
    gen.manual_seed(92)
    obs = [torch.rand(count, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(n)]
    it = interpn_amd.Interpolator.regular(method, dims, starts, steps, vals)
    try:
        got = it.eval_tensors(obs)
        assert it.last_path == "sweep", (it.last_path, it.last_path_reason, it.kernel_name())
        name = it.kernel_name()
        it.finish()
        it.set_option("sweep", 0)
        ref = it.eval_tensors(obs)
        assert it.last_path == "in_place"
        it.finish()
        assert torch.equal(got, ref), name
        idx = torch.randint(0, count, (100_000,), device=dev, generator=gen)
        sub = [o[idx].cpu().numpy() for o in obs]
        w = np.zeros(idx.numel())
        if method == "cubic":
            oracle.cubic_regular(dims, starts, steps, vals, False, sub, w)
        elif method == "linear":
            oracle.linear_regular(dims, starts, steps, vals, sub, w)
        else:
            oracle.nearest_regular(dims, starts, steps, vals, sub, w)
        assert np.array_equal(got[idx].cpu().numpy(), w)
    finally:
        it.close()


def test_sweep_first_bad_index_alignment_and_streams(oracle):
    """The sweep path keeps the reference's abort contract (the smallest failing index of the batch,
    multilinear/regular.rs:277-280, 418), leaves batches whose streams are not 16-byte aligned and
    batches under the automatic threshold to the brick kernel, and serves two streams on one handle
    at once (a scratch block with the work words per stream)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n = 250_003
    case = _sweep_case("regular", [33, 30, 31], n, 4242, specials=False)
    want = run_oracle(oracle, case, True)
    it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
    try:
        it.set_option("sweep", 1)
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        # not 16-byte aligned: the points as they are
        shifted = [torch.cat([t[:1], t])[1:] for t in obs]
        assert all(t.data_ptr() % 16 == 8 for t in shifted)
        got = it.eval_tensors(shifted)
        assert it.last_path == "in_place"
        it.finish()
        assert np.array_equal(got.cpu().numpy(), want)
        # failing points: the smallest index wins, whichever wave meets it
        bad = [t.clone() for t in obs]
        bad[1][200_000] = float("nan")
        bad[2][77_777] = float("inf")
        bad[0][77_778] = float("nan")
        out = it.eval_tensors(bad)
        assert it.last_path == "sweep"
        with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
            it.finish()
        assert ei.value.first_bad_index == 77_777
        assert np.array_equal(out.cpu().numpy()[:77_777], want[:77_777])
        # two streams at once, several launches each (the blocks' work words come back to zero)
        s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        torch.cuda.synchronize()
        outs = []
        for rep in range(3):
            outs.append((it.eval_tensors(obs, stream=s1), it.eval_tensors(obs, stream=s2)))
        it.finish()
        for a, b in outs:
            assert np.array_equal(a.cpu().numpy(), want) and np.array_equal(b.cpu().numpy(), want)
        # scratch reserved up front: no allocation on the launch path, also when allocation is forbidden
        it.reserve(n, 2)
        allocs = it.get_option("scratch_allocs")
        a = it.eval_tensors(obs, stream=s1, no_alloc=True)
        b = it.eval_tensors(obs, stream=s2, no_alloc=True)
        assert it.last_path == "sweep" and it.get_option("scratch_allocs") == allocs
        it.finish()
        assert np.array_equal(a.cpu().numpy(), want) and np.array_equal(b.cpu().numpy(), want)
        # automatic mode: only for batches that give every wave a few rounds (eight where the L2 holds the table, four beyond)
        it.set_option("sweep", -1)
        it.eval_tensors(obs)
        assert it.last_path == "in_place"
        it.finish()
    finally:
        it.close()
    big = _sweep_case("regular", [64, 64, 64], 1000, 7, specials=False)
    it = interpn_amd.Interpolator.regular("linear", big.dims, big.starts, big.steps, big.vals)
    try:
        assert it.get_option("sweep_layout") == 11 and it.table_layout()[1:] == (1, 2)
        gen = torch.Generator(device=dev)
        gen.manual_seed(11)
        for count, path in ((1_000_000, "in_place"), (13_000_003, "sweep")):
            obs = [torch.rand(count, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(3)]
            got = it.eval_tensors(obs)
            assert it.last_path == path, (count, it.last_path, it.last_path_reason)
            it.finish()
            idx = torch.randint(0, count, (200_000,), device=dev, generator=gen)
            sub = [o[idx].cpu().numpy() for o in obs]
            w = np.zeros(idx.numel())
            oracle.linear_regular(big.dims, big.starts, big.steps, big.vals, sub, w)
            assert np.array_equal(got[idx].cpu().numpy(), w)
    finally:
        it.close()


def test_pool_trim_releases_what_destroyed_handles_left(oracle):
    """`interpn_hip_trim`: the blocks of destroyed handles that the per-device pool keeps for reuse go
    back to the driver on request (no public trim call existed; parked blocks could not be
    reused).  After a trim a second one finds nothing; a new handle still works."""
    import interpn_amd

    case = synthetic_case("linear", "regular", 3, [40, 40, 40], 10_000, 31, np.float64, specials=False)
    want = run_oracle(oracle, case, True)
    it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
    assert np.array_equal(it.eval_host(case.obs, np.zeros_like(want)), want)
    it.close()
    freed = interpn_amd.trim()
    assert freed >= 40**3 * 8  # at least the grid's block came back
    assert interpn_amd.trim() == 0
    it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
    assert np.array_equal(it.eval_host(case.obs, np.zeros_like(want)), want)
    it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
@pytest.mark.parametrize("axis", [[6, 7, 5], [4, 4, 4], [20, 17, 33], [64, 64, 9], [33, 12, 130]], ids=str)
def test_column_evaluation_of_sorted_3d_multicubic(oracle, monkeypatch, dtype, axis, kind):
    """Column evaluation for N = 3 (cubic3_column.h): large 3-D multicubic batches are sorted by the
    saturation-class pair of dims 0, 1 (the 4-D sort with dim 2 twice in the record) and a 256-thread
    workgroup evaluates its bin's points out of the cell's column of n2 tiles in LDS.  Forced here at
    small sizes: batches from one point to several parts per bin with ragged tails, ~30 % of the points
    extrapolating (every node form), both `linearize_extrapolation` values, every fifth point sorted into
    the wrong bin (`bin_scramble`: the out-of-cell path, the same tree from the table in global memory),
    the scatter staged or direct, NaN / inf coordinates (regular: the first failing ORIGINAL index;
    rectilinear: they propagate): always the oracle's bits.  [64, 64, 9]: 3969 class-pair bins; [33, 12, 130]:
    a column of 130 tiles.  src/multicubic/regular.rs:325-623, rectilinear.rs:265-545."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    dev = torch.device("cuda:0")
    want_t = torch.float64 if dtype == np.float64 else torch.float32
    for case_no, (nobs, scramble) in enumerate(((1, 0), (700, 0), (5_000, 1), (40_001, 0), (250_013, 1), (250_014, 0))):
        lin = bool((nobs + case_no) % 2)
        case = synthetic_case("cubic", kind, 3, axis, nobs, 8800 + sum(axis) + nobs, dtype, linearize=lin, extrap=0.3,
                              specials=min(axis) >= 8)
        want = run_oracle(oracle, case, True)
        it = _make_interp(interpn_amd, case)
        for k, v in (("binned", 1), ("column", 1), ("bin_scramble", scramble), ("scatter_staged", case_no % 2)):
            it.set_option(k, v)
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        out_full = torch.full((nobs + 2,), -5.0, dtype=want_t, device=dev)
        out = out_full[1:1 + nobs]
        it.eval_tensors(obs, out)
        it.finish()
        assert it.last_path == "binned" and it.kernel_name().startswith("interpn::k_cubic3_column<"), it.kernel_name()
        got = out.cpu().numpy()
        same = (got == want) | (np.isnan(got) & np.isnan(want))
        assert np.all(same), (case_no, nobs, scramble, int((~same).sum()))
        assert float(out_full[0]) == -5.0 and float(out_full[-1]) == -5.0  # nothing outside the batch was written
        if nobs > 1000:
            odd = [o.clone() for o in obs]
            odd[2][nobs // 2] = float("nan")
            odd[0][nobs // 2 + 17] = float("inf")
            if kind == "rectilinear":  # never fails per point: NaN / inf go through the search and propagate
                ocase = synthetic_case("cubic", kind, 3, axis, nobs, 8800 + sum(axis) + nobs, dtype, linearize=lin, extrap=0.3,
                                       specials=min(axis) >= 8)
                ocase.obs[2][nobs // 2] = np.nan
                ocase.obs[0][nobs // 2 + 17] = np.inf
                owant = run_oracle(oracle, ocase, True)
                og = it.eval_tensors(odd).cpu().numpy()
                it.finish()
                assert np.all((og == owant) | (np.isnan(og) & np.isnan(owant)))
            else:
                it.eval_tensors(odd, out)
                with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
                    it.finish()
                assert ei.value.first_bad_index == nobs // 2
        # the same handle with the points evaluated in place: the same bits
        it.set_option("binned", 0)
        ref = it.eval_tensors(obs)
        it.finish()
        rg = ref.cpu().numpy()
        assert it.last_path == "in_place" and np.all((rg == got) | (np.isnan(rg) & np.isnan(got)))
        it.close()


def test_column_evaluation_3d_only_on_request(oracle, monkeypatch):
    """The 3-D column path is never taken by itself (64^3 f64 at 1e7 points: sort 0.35 ms + kernel 0.55 against
    0.65 in place, profiles/REJECTED.md round 5; what large batches take by themselves is the sweep kernel,
    cubic_sweep.h: 0.46 ms); a handle created with INTERPN_HIP_BINNED=1 takes it: the oracle's bits either way."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n = 64
    g = np.linspace(-1.0, 1.0, n)
    dims, starts, steps = [n] * 3, np.full(3, -1.0), np.full(3, g[1] - g[0])
    vals = np.random.default_rng(31).uniform(-1, 1, n**3)
    gen = torch.Generator(device=dev)
    gen.manual_seed(32)
    count = 4_000_000
    obs = [torch.rand(count, dtype=torch.float64, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(3)]
    idx = torch.randint(0, count, (100_000,), device=dev, generator=gen)
    sub = [o[idx].cpu().numpy() for o in obs]
    w = np.zeros(idx.numel())
    oracle.cubic_regular(dims, starts, steps, vals, True, sub, w)
    for env, path in ((None, "sweep"), ("1", "binned")):
        if env:
            monkeypatch.setenv("INTERPN_HIP_BINNED", env)
        it = interpn_amd.Interpolator.regular("cubic", dims, starts, steps, vals, linearize_extrapolation=True)
        try:
            got = it.eval_tensors(obs)
            assert it.last_path == path, (it.last_path, it.last_path_reason)
            if path == "binned":
                assert it.kernel_name().startswith("interpn::k_cubic3_column<double, false, true>"), it.kernel_name()
            it.finish()
            assert np.array_equal(got[idx].cpu().numpy(), w)
        finally:
            it.close()


# ---- the sweep family on observation sets that are not i.i.d. uniform (round 6) -------------------------------------
_SWEEP_SHAPES = {
    # name: (method, kind, dims, the dimension whose cell index is the per-wave sort key)
    "linear3": ("linear", "regular", [40, 19, 23], 0),
    "linear3_rect": ("linear", "rectilinear", [40, 19, 23], 0),
    "linear3_rect_long": ("linear", "rectilinear", [130, 9, 11], 0),
    "linear2": ("linear", "regular", [200, 40], 0),
    "nearest3": ("nearest", "regular", [40, 19, 23], 0),
    "nearest2": ("nearest", "regular", [300, 9], 0),
    "cubic3": ("cubic", "regular", [12, 11, 90], 2),
    "cubic3_rect": ("cubic", "rectilinear", [12, 11, 90], 2),
    "cubic2": ("cubic", "regular", [150, 40], 0),
}


def structured_obs(case, dist, keydim, seed):
    """Observation sets that drive the per-wave counting sort of the sweep kernels (sweep_rounds.h, linear_sweep.h) through
    its corners, in place in `case.obs`:
      one_cell   every point of the first half inside ONE interior cell, of the second half in the top cell and beyond it
                 (one bin per wave: all 1024 points of a round rank into the same counter);
      sorted     i.i.d. points ordered by the key dimension's coordinate (every wave's points in a narrow band of bins, bands
                 moving with the wave's place in the batch: no two waves sweep the same slab);
      lattice    a regular lattice finer than the grid in C order (re-gridding: long runs of one key, neighbours share lines);
      on_planes  every coordinate exactly a grid coordinate (x - izl == 0: the division-free forms refuse every point, the whole
                 batch takes the wave-uniform divide sequences; cubic: every class boundary);
      half_nan   the second half of the batch with NaN in a random dimension of every other point (regular grids: the first
                 failing index and the prefix; rectilinear: NaN propagates)."""
    rng = np.random.default_rng(seed)
    n = len(case.grids)
    nobs = case.obs[0].size
    dtype = case.vals.dtype
    if dist == "one_cell":
        h = nobs // 2
        for d in range(n):
            g = case.grids[d].astype(np.float64)
            c = int(rng.integers(1, max(2, g.size - 2)))
            case.obs[d][:h] = rng.uniform(g[c], g[c + 1], h).astype(dtype)
            case.obs[d][h:] = rng.uniform(g[-2], g[-1] + 0.7 * (g[-1] - g[-2]), nobs - h).astype(dtype)
    elif dist == "sorted":
        order = np.argsort(case.obs[keydim], kind="stable")
        for d in range(n):
            case.obs[d][:] = case.obs[d][order]
    elif dist == "lattice":
        m = int(np.ceil(nobs ** (1.0 / n)))
        axes = [np.linspace(float(case.grids[d][0]) - 0.02, float(case.grids[d][-1]) + 0.02, m) for d in range(n)]
        mesh = np.meshgrid(*axes, indexing="ij")
        for d in range(n):
            case.obs[d][:] = mesh[d].ravel()[:nobs].astype(dtype)
    elif dist == "on_planes":
        for d in range(n):
            case.obs[d][:] = case.grids[d][rng.integers(0, case.grids[d].size, nobs)]
    elif dist == "half_nan":
        h = nobs // 2
        idx = np.arange(h, nobs, 2)
        dim = rng.integers(0, n, idx.size)
        for d in range(n):
            case.obs[d][idx[dim == d]] = np.nan
    else:
        raise ValueError(dist)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("dist", ["one_cell", "sorted", "lattice", "on_planes", "half_nan"])
@pytest.mark.parametrize("shape", list(_SWEEP_SHAPES), ids=list(_SWEEP_SHAPES))
def test_sweep_family_on_structured_observation_sets(oracle, shape, dist, dtype):
    """Every sweep kernel (3-D / 2-D multilinear, nearest, 3-D / 2-D multicubic; regular and rectilinear grids) forced onto
    batches whose points are NOT i.i.d. uniform — all in one cell, already ordered by the kernel's own key, a lattice finer than
    the grid, every point on grid planes, half the batch NaN — against the oracle and, bit for bit, against the one-pass
    kernel, with the clock and without.  The reference also measures un-shuffled grids of points
    (benches/bench.rs:554-571); a point's result depends on its own coordinates only (multilinear/regular.rs:276-280)."""
    import torch

    import interpn_amd

    method, kind, dims, keydim = _SWEEP_SHAPES[shape]
    dev = torch.device("cuda:0")
    nobs = 150_011
    case = synthetic_case(method, kind, len(dims), dims, nobs, 6100 + sum(dims), dtype, linearize=True, extrap=0.1, specials=False)
    clean = [o.copy() for o in case.obs]
    structured_obs(case, dist, keydim, 6200 + sum(dims))
    fails = dist == "half_nan" and kind == "regular"
    if fails:  # the oracle stops at the first NaN like the reference: expected values from the points without them
        keep = case.obs
        case.obs = clean
        want = run_oracle(oracle, case, True)
        case.obs = keep
        first_bad = int(min(np.flatnonzero(np.isnan(o))[0] for o in case.obs if np.isnan(o).any()))
    else:
        want = run_oracle(oracle, case, True)
    if kind == "regular":
        it = interpn_amd.Interpolator.regular(method, case.dims, case.starts, case.steps, case.vals, linearize_extrapolation=True)
    else:
        it = interpn_amd.Interpolator.rectilinear(method, case.grids, case.vals, linearize_extrapolation=True)
    try:
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        for period in (0, 1, 900):
            it.set_option("sweep", 1)
            it.set_option("sweep_period", period)
            got = it.eval_tensors(obs)
            assert it.last_path == "sweep", (it.last_path, it.last_path_reason)
            assert "_sweep<" in it.kernel_name(), it.kernel_name()
            if fails:
                with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
                    it.finish()
                assert ei.value.first_bad_index == first_bad
                g = got.cpu().numpy()[:first_bad]
                assert np.array_equal(g, want[:first_bad]), period
                continue
            it.finish()
            it.set_option("sweep", 0)
            ref = it.eval_tensors(obs)
            assert it.last_path == "in_place", it.last_path
            it.finish()
            g, r = got.cpu().numpy(), ref.cpu().numpy()
            assert np.all((g == r) | (np.isnan(g) & np.isnan(r))), (period, "sweep != one-pass kernel")
            same = (g == want) | (np.isnan(g) & np.isnan(want))
            assert np.all(same), (period, int((~same).sum()))
    finally:
        it.close()


def test_column_kernel_build_that_was_miscompiled():
    """Round 5 found a build of the rectilinear 4-D column kernel (k_cubic_column<float, true, ...>) that differed from the
    product by an unused kernel argument and a never-taken branch and returned wrong values for every point whose class along
    dim 1 is High.  Cause (profiles/NOTES.md section H): the node's three class arms formed a switch whose default arm (High)
    is entered from both halves of the lowered decision tree, and the compiler's StructurizeCFG gave the High lanes entering
    from the `sat >= 1` side the Low arm's y0 and an undefined y1.  The nodes are now one two-way branch with selects
    (interpn_device.h::cubic_rect_saturated); this test runs the binned 4-D rectilinear cases through THAT build
    (tools/libinterpn_colvariant.so: the product's objects with the column kernel compiled with
    -DINTERPN_COLUMN_CREC_VARIANT) in a child process (multicubic/rectilinear.rs:509-542 are the arms in question)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "tools", "libinterpn_colvariant.so")
    if not os.path.exists(lib):
        pytest.skip("tools/libinterpn_colvariant.so not built (make -C interpn_amd/csrc colvariant)")
    ids = [f"tests/test_gpu_parity.py::test_binned_multicubic_evaluation[{axis}-11-rectilinear-{t}]"
           for axis in ("[5, 6, 4, 7]", "[40, 37, 5, 4]") for t in ("f32", "f64")]
    env = dict(os.environ, INTERPN_AMD_LIB=lib)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + ids, cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "4 passed" in r.stdout, r.stdout[-1000:]


@pytest.mark.parametrize("order", ["sweep_then_sorted", "sorted_then_sweep"])
def test_sweep_and_sorted_evaluation_share_a_scratch_block(oracle, monkeypatch, order):
    """One handle, one stream, both paths through the same scratch block: a sweep launch of at least four rounds per wave
    leaves its measured period inside the block's work words — bytes that are one of the sort's bin counters — so each
    path has its own "block is clean" flag and clears the other's (abi_internal.h::BinSlot).  Sweep first, then the sorted
    evaluation (the order that started the sort from a non-zero counter before round 6), and the reverse; the results are
    the in-place kernel's bit for bit and the oracle's on a sample."""
    import torch

    import interpn_amd

    monkeypatch.setenv("INTERPN_HIP_BRICKS", "11")
    dev = torch.device("cuda:0")
    nobs = 9_000_000
    case = synthetic_case("cubic", "regular", 3, [24, 22, 26], nobs, 6400, np.float64, linearize=True, extrap=0.1, specials=True)
    it = _make_interp(interpn_amd, case)
    try:
        obs = [torch.from_numpy(o).to(dev) for o in case.obs]
        it.set_option("sweep", 0)
        it.set_option("binned", 0)
        ref = it.eval_tensors(obs).clone()
        it.finish()
        assert it.last_path == "in_place"
        sample = np.random.default_rng(5).choice(nobs, 100_000, replace=False)
        sub = kat.Case("s", "cubic", "regular", case.grids, case.vals, [o[sample] for o in case.obs], np.zeros(sample.size), 0.0, linearize=True)
        assert np.array_equal(ref.cpu().numpy()[sample], run_oracle(oracle, sub, True))

        def sweep():
            it.set_option("binned", 0)
            it.set_option("sweep", 1)
            got = it.eval_tensors(obs)
            assert it.last_path == "sweep", (it.last_path, it.last_path_reason)
            it.finish()
            assert torch.equal(got, ref)

        def sorted_():
            it.set_option("sweep", 0)
            it.set_option("binned", 1)
            got = it.eval_tensors(obs)
            assert it.last_path == "binned", (it.last_path, it.last_path_reason)
            it.finish()
            assert torch.equal(got, ref)

        for step in ((sweep, sweep, sorted_, sorted_, sweep, sorted_) if order == "sweep_then_sorted" else (sorted_, sweep, sweep, sorted_)):
            step()
    finally:
        it.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", ["regular", "rectilinear"])
def test_automatic_path_samples_the_batch_on_the_device(oracle, kind, dtype):
    """Large 3-D multilinear batches in automatic mode: a sampling kernel in front decides on the device whether the sweep
    kernel (unordered points) or the one-pass brick kernel (points that are coherent as they stand: a fine lattice, a cluster)
    evaluates the batch; both launches are enqueued and one returns at once (k_linear_sweep.hip::k_sweep_probe,
    abi_sweep.hip).  Either way the results are the forced kernels' bit for bit and the oracle's on a sample, the verdict is
    the expected one, and an unrepresentable coordinate is reported with its index whichever kernel ran
    (multilinear/regular.rs:268-283)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n = 128
    P = 20_000_000
    td = torch.float64 if dtype == np.float64 else torch.float32
    case = synthetic_case("linear", kind, 3, [n] * 3, 64, 6500, dtype, extrap=0.0, specials=False)
    if kind == "regular":
        it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
    else:
        it = interpn_amd.Interpolator.rectilinear("linear", case.grids, case.vals)
    try:
        it.set_option("sweep_probe", 1)  # a sample in front of every automatic launch (the default thins them out: next test)
        m = 272
        ax = torch.linspace(-1.02, 1.02, m, dtype=td, device=dev)
        lat = [t.reshape(-1)[:P].contiguous() for t in torch.meshgrid(ax, ax, ax, indexing="ij")]
        gen = torch.Generator(device=dev)
        gen.manual_seed(11)
        rnd = [torch.rand(P, dtype=td, device=dev, generator=gen) * 2.1 - 1.05 for _ in range(3)]
        half = [torch.cat([a[:P // 2], b[P // 2:]]) for a, b in zip(lat, rnd)]
        sample = np.random.default_rng(7).choice(P, 50_000, replace=False)
        for name, obs, verdict in (("lattice", lat, 1), ("random", rnd, 0), ("half", half, None)):
            it.set_option("sweep", 1)
            a = it.eval_tensors(obs).clone()
            it.finish()
            assert it.last_path == "sweep" and it.get_option("sweep_probe_took_brick") == -1
            it.set_option("sweep", 0)
            b = it.eval_tensors(obs).clone()
            it.finish()
            assert torch.equal(a, b), name
            it.set_option("sweep", -1)
            out = torch.full_like(a, -7.0)
            it.eval_tensors(obs, out)
            it.finish()
            assert it.last_path == "sweep", (name, it.last_path, it.last_path_reason)
            took = it.get_option("sweep_probe_took_brick")
            assert took in (0, 1) and (verdict is None or took == verdict), (name, took)
            assert torch.equal(out, a), name
            sub = kat.Case("s", "linear", kind, case.grids, case.vals, [o[sample].cpu().numpy() for o in obs], np.zeros(sample.size, dtype), 0.0)
            assert np.array_equal(out[sample].cpu().numpy(), run_oracle(oracle, sub, True)), name
            # the sample's verdict can be switched off: then the sweep kernel takes the batch whatever it looks like
            it.set_option("sweep_probe", 0)
            it.eval_tensors(obs, out)
            it.finish()
            assert it.get_option("sweep_probe_took_brick") == -1 and torch.equal(out, a)
            it.set_option("sweep_probe", 1)
        if kind == "regular":  # the first failing index, through the gated pair
            for obs in (lat, rnd):
                bad = [o.clone() for o in obs]
                bad[1][P - 5] = float("nan")
                bad[2][7_000_003] = float("inf")
                good = it.eval_tensors(obs).clone()
                it.finish()
                out = it.eval_tensors(bad)
                with pytest.raises(AssertionError, match="Unrepresentable coordinate value") as ei:
                    it.finish()
                assert ei.value.first_bad_index == 7_000_003
                assert torch.equal(out[:7_000_003], good[:7_000_003])
    finally:
        it.close()


@pytest.mark.parametrize("method,dims,count", [("nearest", [128, 128, 128], 20_000_000), ("nearest", [1000, 1000], 20_000_000),
                                               ("linear", [1000, 1000], 25_000_000), ("cubic", [512, 512], 30_000_000)],
                         ids=["nearest3", "nearest2", "linear2", "cubic2"])
def test_automatic_path_samples_the_batch_on_the_device_family(oracle, method, dims, count):
    """The same device-side sample in front of the automatic launches of the other sweep kernels whose one-pass kernel wins on
    coherent batches (nearest-neighbour 2-D / 3-D, 2-D multilinear, 2-D multicubic): a fine lattice goes to the one-pass
    kernel, unordered points to the sweep kernel, the results are both forced kernels' bit for bit and the oracle's on a
    sample (nearest/regular.rs:234-317, multilinear/regular.rs:268-283, multicubic/regular.rs:297-313)."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    nd = len(dims)
    case = synthetic_case(method, "regular", nd, dims, 64, 6600 + sum(dims), np.float64, linearize=True, extrap=0.0, specials=False)
    it = interpn_amd.Interpolator.regular(method, case.dims, case.starts, case.steps, case.vals, linearize_extrapolation=True)
    try:
        it.set_option("sweep_probe", 1)
        m = int(np.ceil(count ** (1.0 / nd)))
        ax = torch.linspace(-1.01, 1.01, m, dtype=torch.float64, device=dev)
        lat = [t.reshape(-1)[:count].contiguous() for t in torch.meshgrid(*([ax] * nd), indexing="ij")]
        gen = torch.Generator(device=dev)
        gen.manual_seed(12)
        rnd = [torch.rand(count, dtype=torch.float64, device=dev, generator=gen) * 2.04 - 1.02 for _ in range(nd)]
        sample = np.random.default_rng(8).choice(count, 50_000, replace=False)
        for name, obs, verdict in (("lattice", lat, 1), ("random", rnd, 0)):
            it.set_option("sweep", 1)
            a = it.eval_tensors(obs).clone()
            it.finish()
            assert it.last_path == "sweep", (name, it.last_path, it.last_path_reason)
            it.set_option("sweep", 0)
            b = it.eval_tensors(obs).clone()
            it.finish()
            assert it.last_path == "in_place" and torch.equal(a, b), name
            it.set_option("sweep", -1)
            out = torch.full_like(a, -7.0)
            it.eval_tensors(obs, out)
            it.finish()
            assert it.last_path == "sweep", (name, it.last_path, it.last_path_reason)
            assert it.get_option("sweep_probe_took_brick") == verdict, name
            assert torch.equal(out, a), name
            sub = kat.Case("s", method, "regular", case.grids, case.vals, [o[sample].cpu().numpy() for o in obs], np.zeros(sample.size), 0.0, linearize=True)
            assert np.array_equal(out[sample].cpu().numpy(), run_oracle(oracle, sub, True)), name
    finally:
        it.close()


def test_automatic_path_thins_the_samples_out(oracle):
    """Default policy of the automatic 3-D multilinear path (option sweep_probe = 2, abi_sweep.hip): the batch is sampled on
    every launch until three samples in a row came out unordered, then on every 16th launch; one coherent verdict brings the
    per-launch sample back.  The host learns the verdicts from a pinned word the sampling kernel writes, without ever
    synchronising.  Whatever is skipped or taken, the results are the same bits."""
    import torch

    import interpn_amd

    dev = torch.device("cuda:0")
    n = 128
    P = 14_000_000
    case = synthetic_case("linear", "regular", 3, [n] * 3, 64, 6700, np.float64, extrap=0.0, specials=False)
    it = interpn_amd.Interpolator.regular("linear", case.dims, case.starts, case.steps, case.vals)
    try:
        assert it.get_option("sweep_probe") == 2
        gen = torch.Generator(device=dev)
        gen.manual_seed(13)
        rnd = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(3)]
        m = 242
        ax = torch.linspace(-1.0, 1.0, m, dtype=torch.float64, device=dev)
        lat = [t.reshape(-1)[:P].contiguous() for t in torch.meshgrid(ax, ax, ax, indexing="ij")]
        it.set_option("sweep", 0)
        want_rnd = it.eval_tensors(rnd).clone()
        want_lat = it.eval_tensors(lat).clone()
        it.finish()
        it.set_option("sweep", -1)
        seen = []
        for _ in range(40):  # unordered batches: sampled at first, then mostly not
            out = it.eval_tensors(rnd)
            it.finish()
            assert it.last_path == "sweep" and torch.equal(out, want_rnd)
            seen.append(it.get_option("sweep_probe_took_brick"))
        assert seen[0] == 0 and set(seen) <= {0, -1}
        assert seen.count(-1) >= 20, seen          # most launches went without a sample ...
        assert 0 in seen[8:], seen                 # ... but not all of them
        seen = []
        for _ in range(40):  # now a fine lattice: within 16 launches a sample finds it, from then on every launch is sampled
            out = it.eval_tensors(lat)
            it.finish()
            assert torch.equal(out, want_lat)
            seen.append(it.get_option("sweep_probe_took_brick"))
        assert 1 in seen[:17], seen
        first = seen.index(1)
        # ... and after three coherent samples in a row the mirror image: the one-pass kernel alone, sampled every 16th launch
        assert set(seen[first + 2:]) <= {1, -1}, seen
        assert seen[first + 2:].count(-1) >= 10 and 1 in seen[first + 8:], seen
        names = []
        for _ in range(20):  # unordered again: within 16 launches a sample says so and the sweep kernel is back
            out = it.eval_tensors(rnd)
            it.finish()
            assert it.last_path == "sweep" and torch.equal(out, want_rnd)
            seen.append(it.get_option("sweep_probe_took_brick"))
            names.append(it.kernel_name())
        assert 0 in seen[-20:-3], seen
        assert names[-1].startswith("interpn::k_linear_sweep"), names
    finally:
        it.close()
