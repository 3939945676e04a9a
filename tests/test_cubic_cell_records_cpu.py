"""The per-cell records of a rectilinear multicubic axis (interpn_amd/csrc/cubic_cell_record.h), built by the header's own
host code (compiled here with g++) and compared value by value with the per-point setup of the reference restated in numpy
scalars of the same type (multicubic/rectilinear.rs:413-545, mod.rs:103-117): a kernel that reads a record must hold the bits
the per-point divisions give.  No GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "cubic_cell_record.h"
extern "C" void records_f64(const double* g, int n, double* out) {
  std::vector<interpn::CubicCellRecord<double>> r;
  interpn::build_cubic_cell_records<double>(g, n, r);
  for (size_t k = 0; k < r.size(); ++k) { const double* f = &r[k].gref; for (int e = 0; e < 12; ++e) out[k * 12 + e] = f[e]; }
}
extern "C" void records_f32(const float* g, int n, float* out) {
  std::vector<interpn::CubicCellRecord<float>> r;
  interpn::build_cubic_cell_records<float>(g, n, r);
  for (size_t k = 0; k < r.size(); ++k) { const float* f = &r[k].gref; for (int e = 0; e < 12; ++e) out[k * 12 + e] = f[e]; }
}
'''


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    d = tmp_path_factory.mktemp("crec")
    src = d / "crec.cpp"
    src.write_text(SRC)
    so = d / "libcrec.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", os.path.join(ROOT, "interpn_amd", "csrc"),
                    str(src), "-o", str(so)], check=True)
    return ctypes.CDLL(str(so))


def expected(g, k):
    """[gref, h, rh, r0, a0, c0, rr0, r1, a1, c1, rr1] of cell k in g's own type."""
    T = g.dtype.type
    one = T(1)
    n = g.size
    r1 = a1 = c1 = one
    with np.errstate(all="ignore"):
        if k == 0:
            h01, h12 = g[1] - g[0], g[2] - g[1]
            r0 = h12 / h01
            a0, c0 = one / (one + r0), r0 / (r0 + one)
            gref, h = g[1], h01
        elif k == n - 2:
            h12, h23 = g[n - 2] - g[n - 3], g[n - 1] - g[n - 2]
            r0 = h12 / h23
            a0, c0 = r0 / (r0 + one), one / (one + r0)
            gref, h = g[n - 2], h23
        else:
            h01, h12, h23 = g[k] - g[k - 1], g[k + 1] - g[k], g[k + 2] - g[k + 1]
            r0 = h01 / h12
            a0, c0 = r0 / (r0 + one), one / (one + r0)
            r1 = h23 / h12
            a1, c1 = one / (one + r1), r1 / (r1 + one)
            gref, h = g[k], h12
        return [gref, h, one / h, r0, a0, c0, one / r0, r1, a1, c1, one / r1]


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_records_equal_the_per_point_setup(lib, dtype):
    rng = np.random.default_rng(3)
    f64 = dtype == np.float64
    fn = lib.records_f64 if f64 else lib.records_f32
    lo, hi = (2.0 ** -128, 2.0 ** 128) if f64 else (2.0 ** -16, 2.0 ** 16)
    for n in (4, 5, 6, 17, 64, 301):
        for flavour in ("jitter", "wild", "unsorted"):
            g = np.cumsum(rng.uniform(0.2, 1.8, n))
            if flavour == "wild":
                g = np.cumsum(10.0 ** rng.uniform(-6 if not f64 else -170, 4 if not f64 else 100, n))
            if flavour == "unsorted":
                g[n // 2] = g[0] - 1.0  # legal input: the reference's `new` only checks g[1] > g[0]
            g = np.ascontiguousarray(g.astype(dtype))
            out = np.zeros((n - 1) * 12, dtype=dtype)
            fn(g.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(n), out.ctypes.data_as(ctypes.c_void_p))
            out = out.reshape(n - 1, 12)
            flags = 0
            for k in range(n - 1):
                want = np.array(expected(g, k), dtype=dtype)
                got = out[k, :11]
                same = (got.view(np.uint64 if f64 else np.uint32) == want.view(np.uint64 if f64 else np.uint32)) | (np.isnan(got) & np.isnan(want))
                assert same.all(), (n, flavour, k, got, want)
                ok = all(lo <= float(v) < hi for v in (want[1], want[3], want[7]))
                assert out[k, 11] == (1.0 if ok else 0.0), (n, flavour, k)
                flags += int(ok)
            if flavour == "jitter":
                assert flags == n - 1
