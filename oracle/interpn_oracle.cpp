// =============================================================================
// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU restatement ("oracle") of the batched per-observation-point hot path of
// jlogan03/interpn v0.8.2 (multilinear, multicubic, and the nearest-neighbour sibling).  Only `tests/`, `__graft_entry__.smoke()` and the
// `cpu_baseline` leg of `bench.py` may load this library; the product path
// (interpn_amd/csrc, include/interpn_hip.h) never links, loads or calls it.
//
// Parity status: PINNED.  The reference cannot be compiled (no rustc/cargo in the
// image) or imported (its Python package needs the compiled cdylib), so this
// restatement is pinned against the reference's own known-answer tests
// (tests/test_oracle_kat.py re-creates every hot-path row of SURVEY.md §4) and
// against an independent exact-rational evaluation of the same formulas
// (oracle/exact_rational.py).
//
// Each function cites the reference file:line it follows (paths relative to the
// reference repository root).  Arithmetic order, FMA sites, index conversion,
// bisection semantics, validation order and error strings follow the Rust source.
//
// Third-party semantics restated (not present under /root/reference):
//   * num-traits 0.2.19  <isize as NumCast>::from(float): Some(trunc) iff
//     -2^63 <= f < 2^63 (NaN -> None).
//   * core::slice::partition_point: count of leading elements satisfying the
//     predicate on a partitioned slice, by bisection (restated below so that
//     unsorted input also takes the same probes as Rust's binary search).
//
// Build: see oracle/Makefile.  -ffp-contract=off is REQUIRED: every fused
// multiply-add below is explicit (the `fma` cargo feature of the reference).
// =============================================================================
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace {

enum Status : int {
  OK = 0,
  ERR_DIM_MISMATCH = 1,         // "Dimension mismatch"
  ERR_MIN_TWO_ENTRIES = 2,      // "All grids must have at least two entries"     (linear regular)
  ERR_MIN_2_ENTRIES = 3,        // "All grids must have at least 2 entries"       (linear rectilinear)
  ERR_MIN_FOUR_ENTRIES = 4,     // "All grids must have at least four entries"    (cubic regular)
  ERR_MIN_4_ENTRIES = 5,        // "All grids must have at least 4 entries"       (cubic rectilinear)
  ERR_NOT_MONOTONIC = 6,        // "All grids must be monotonically increasing"
  ERR_UNREPRESENTABLE = 7,      // "Unrepresentable coordinate value"
  ERR_TOO_MANY_DIMS = 8,        // "Dimension exceeds maximum (8). Use interpolator struct directly for higher dimensions."
  ERR_REFERENCE_PANIC = 9,      // the reference would panic (slice->array try_into().unwrap(), usize overflow)
  ERR_TOO_MANY_DIMS_6 = 10,     // "Dimension exceeds maximum (6)."  (nearest)
};

constexpr int MAXDIMS = 8;  // src/python.rs:10

enum Sat : int { SatNone = 0, InsideLow, OutsideLow, InsideHigh, OutsideHigh };  // src/multicubic/mod.rs:59-66

template <bool FMA, typename T>
inline T mul_add(T a, T b, T c) {  // a*b + c ; num-traits Float::mul_add when FMA, two roundings otherwise
  if (FMA) return std::fma(a, b, c);
  T p = a * b;
  return p + c;
}

// num-traits 0.2.19 float -> isize (see header).
template <typename T>
inline bool to_isize(T f, int64_t* out) {
  const T lo = (T)-9223372036854775808.0;
  const T hi = (T)9223372036854775808.0;
  if (f >= lo && f < hi) {
    *out = (int64_t)f;
    return true;
  }
  return false;
}

// core::slice::partition_point(|g| *g < v) — Rust std binary_search_by restated:
// size-halving loop, returns the first index whose element does not satisfy pred.
template <typename T>
inline size_t partition_point_lt(const T* g, size_t n, T v) {
  size_t size = n;
  if (size == 0) return 0;
  size_t base = 0;
  while (size > 1) {
    size_t half = size / 2;
    size_t mid = base + half;
    // cmp = if pred(mid) {Less} else {Greater}; base = if cmp == Greater { base } else { mid }
    base = (g[mid] < v) ? mid : base;
    size -= half;
  }
  return base + ((g[base] < v) ? 1 : 0);
}

// ---------------------------------------------------------------------------
// src/multicubic/mod.rs:72-91
template <bool FMA, typename T>
inline T normalized_hermite_spline(T t, T y0, T dy, T k0, T k1) {
  T a = k0 - dy;
  T b = -k1 + dy;
  T c1 = dy + a;
  T c2 = b - (a + a);
  T c3 = a - b;
  if (FMA) {
    return std::fma(std::fma(std::fma(c3, t, c2), t, c1), t, y0);  // mod.rs:89
  } else {
    T i0 = t * c3;
    T i1 = c2 + i0;
    T i2 = t * i1;
    T i3 = c1 + i2;
    T i4 = t * i3;
    return y0 + i4;  // mod.rs:85
  }
}

// src/multicubic/mod.rs:103-117
template <bool FMA, typename T>
inline T centered_difference_nonuniform(T y0, T y1, T y2, T h01, T h12) {
  T a = h01 / (h01 + h12);
  T b = (y2 - y1) / h12;
  T c = h12 / (h12 + h01);
  T d = (y1 - y0) / h01;
  if (FMA) {
    return std::fma(a, b, c * d);  // mod.rs:115
  } else {
    T ab = a * b;
    T cd = c * d;
    return ab + cd;  // mod.rs:111
  }
}

// src/multicubic/regular.rs:474-623 (flattened) and regular_recursive.rs:470-609 (recursive).
// `two.mul_add(dy, -k0)` and `two * dy - k0` round identically while 2*dy is finite (the product is exact); where it
// overflows they do not, so the one place the recursive arm keeps the unfused form under the `fma` feature —
// OutsideLow, regular_recursive.rs:536 — is followed (`recursive_arm`: N >= 5; every other saturated class of both arms is
// cfg-switched: regular.rs:525-528, :546-549, :580-583, :602-605, regular_recursive.rs:516-519, :567-570, :589-592).
template <bool FMA, typename T>
inline T cubic_regular_inner(const T* vals, T t, Sat sat, bool linearize, bool recursive_arm) {
  const T one = (T)1, two = (T)2;
  switch (sat) {
    case SatNone: {
      T y0 = vals[1];
      T dy = vals[2] - vals[1];
      T k0 = (vals[2] - vals[0]) / two;
      T k1 = (vals[3] - vals[1]) / two;
      return normalized_hermite_spline<FMA>(t, y0, dy, k0, k1);
    }
    case InsideLow:
    case OutsideLow: {
      T tt = -t;
      T y0 = vals[1];
      T y1 = vals[0];
      T dy = vals[0] - vals[1];
      T k0 = -(vals[2] - vals[0]) / two;
      T k1 = (recursive_arm && sat == OutsideLow) ? two * dy - k0 : mul_add<FMA>(two, dy, -k0);
      if (sat == OutsideLow && linearize) return mul_add<FMA>(k1, tt - one, y1);  // regular.rs:553-561
      return normalized_hermite_spline<FMA>(tt, y0, dy, k0, k1);
    }
    case InsideHigh:
    case OutsideHigh: {
      T tt = t - one;
      T y0 = vals[2];
      T y1 = vals[3];
      T dy = vals[3] - vals[2];
      T k0 = (vals[3] - vals[1]) / two;
      T k1 = mul_add<FMA>(two, dy, -k0);
      if (sat == OutsideHigh && linearize) return mul_add<FMA>(k1, tt - one, y1);  // regular.rs:609-617
      return normalized_hermite_spline<FMA>(tt, y0, dy, k0, k1);
    }
  }
  return (T)0;
}

// src/multicubic/rectilinear.rs:413-545 (flattened: linearized branch is NEVER fused,
// :500,:539) and rectilinear_recursive.rs:385-545 (recursive: fused under `fma`, :467,:527).
template <bool FMA, typename T>
inline T cubic_rectilinear_inner(const T* vals, const T* g, T x, Sat sat, bool linearize,
                                 bool fma_linearize) {
  const T one = (T)1, two = (T)2;
  switch (sat) {
    case SatNone: {
      T y0 = vals[1];
      T dy = vals[2] - vals[1];
      T h01 = g[1] - g[0];
      T h12 = g[2] - g[1];
      T h23 = g[3] - g[2];
      T k0 = centered_difference_nonuniform<FMA>(vals[0], vals[1], vals[2], h01 / h12, one);
      T k1 = centered_difference_nonuniform<FMA>(vals[1], vals[2], vals[3], one, h23 / h12);
      T t = (x - g[1]) / h12;
      return normalized_hermite_spline<FMA>(t, y0, dy, k0, k1);
    }
    case InsideLow:
    case OutsideLow: {
      T y0 = vals[1];
      T y1 = vals[0];
      T dy = vals[0] - vals[1];
      T h01 = g[1] - g[0];
      T h12 = g[2] - g[1];
      T k0 = -centered_difference_nonuniform<FMA>(vals[0], vals[1], vals[2], one, h12 / h01);
      T k1 = two * dy - k0;  // rectilinear.rs:471,493,515,532: never fused, with or without the `fma` feature
      T t = -(x - g[1]) / h01;
      if (sat == OutsideLow && linearize) {
        if (FMA && fma_linearize) return std::fma(k1, t - one, y1);
        T p = k1 * (t - one);
        return y1 + p;
      }
      return normalized_hermite_spline<FMA>(t, y0, dy, k0, k1);
    }
    case InsideHigh:
    case OutsideHigh: {
      T y0 = vals[2];
      T y1 = vals[3];
      T dy = vals[3] - vals[2];
      T h12 = g[2] - g[1];
      T h23 = g[3] - g[2];
      T k0 = centered_difference_nonuniform<FMA>(vals[1], vals[2], vals[3], h12 / h23, one);
      T k1 = two * dy - k0;
      T t = (x - g[2]) / h23;
      if (sat == OutsideHigh && linearize) {
        if (FMA && fma_linearize) return std::fma(k1, t - one, y1);
        T p = k1 * (t - one);
        return y1 + p;
      }
      return normalized_hermite_spline<FMA>(t, y0, dy, k0, k1);
    }
  }
  return (T)0;
}

// ---------------------------------------------------------------------------
// Per-point state shared by the four methods.
template <typename T>
struct Point {
  size_t origin[MAXDIMS];
  size_t dimprod[MAXDIMS];
  T dts[MAXDIMS];
  T x[MAXDIMS];
  Sat sat[MAXDIMS];
};

// Tree reduction.  Same dependency tree as the flattened const loops
// (multilinear/regular.rs:347-403, multicubic/regular.rs:368-421) and as the
// recursive `populate` (regular_recursive.rs:348-389): dim 0 is reduced first
// (innermost), dim N-1 last.  Leaves: src/lib.rs:119-144 (Σ loc[j]*dimprod[j]).
template <typename Node, int FP, typename T>
inline T populate(int dim, size_t base, const Point<T>& p, const T* vals, const Node& node) {
  if (dim == 0) return vals[base];
  const int nd = dim - 1;
  T v[FP];
  for (int i = 0; i < FP; ++i) v[i] = populate<Node, FP, T>(nd, base + (size_t)i * p.dimprod[nd], p, vals, node);
  return node(v, nd, p);
}

inline bool checked_product(const size_t* dims, size_t n, size_t* out) {
  size_t acc = 1;
  for (size_t i = 0; i < n; ++i) {
    if (__builtin_mul_overflow(acc, dims[i], &acc)) return false;  // Cargo.toml:45 overflow-checks => panic
  }
  *out = acc;
  return true;
}

template <typename T>
inline void fill_dimprod(Point<T>& p, const size_t* dims, int n) {
  size_t acc = 1;
  for (int i = 0; i < n; ++i) {  // regular_recursive.rs:297-301
    p.dimprod[n - i - 1] = acc;
    acc *= dims[n - i - 1];
  }
}

template <typename T>
inline size_t origin_base(const Point<T>& p, int n) {
  size_t b = 0;
  for (int j = 0; j < n; ++j) b += p.origin[j] * p.dimprod[j];
  return b;
}

// ---------------------------------------------------------------------------
// multilinear::regular — src/multilinear/regular.rs:51-117 (dispatch), :225-259 (new),
// :268-283 (interp), :296-425 (interp_one/get_loc); N=7,8: regular_recursive.rs:191-344.
template <typename T, bool FMA>
int linear_regular(const size_t* dims, size_t ndims, const T* starts, size_t nstarts, const T* steps,
                   size_t nsteps, const T* vals, size_t nvals, const T* const* obs, const size_t* obs_lens,
                   size_t nobs, T* out, size_t nout, size_t* first_bad) {
  if (nstarts != ndims || nsteps != ndims || nobs != ndims) return ERR_DIM_MISMATCH;  // regular.rs:60
  if (ndims < 1 || ndims > (size_t)MAXDIMS) return ERR_TOO_MANY_DIMS;                 // regular.rs:111-113
  const int n = (int)ndims;
  size_t prod;
  if (!checked_product(dims, ndims, &prod)) return ERR_REFERENCE_PANIC;
  if (nvals != prod) return ERR_DIM_MISMATCH;                                   // regular.rs:239
  for (int i = 0; i < n; ++i) if (dims[i] < 2) return ERR_MIN_TWO_ENTRIES;      // regular.rs:243
  for (int i = 0; i < n; ++i) if (!(steps[i] > (T)0)) return ERR_NOT_MONOTONIC;  // regular.rs:248
  for (int i = 0; i < n; ++i) if (obs_lens[i] != nout) return ERR_DIM_MISMATCH;  // regular.rs:271
  // FMA site that differs between the two arms: index_zero_loc is fused only in the
  // flattened struct (regular.rs:334-337), never in the recursive one (regular_recursive.rs:310-313).
  const bool fma_index = FMA && n <= 6;

  Point<T> p;
  fill_dimprod(p, dims, n);
  auto node = [](const T* v, int d, const Point<T>& q) -> T {
    T y0 = v[0];
    T dy = v[1] - y0;
    return mul_add<FMA>(q.dts[d], dy, y0);  // regular.rs:378-385
  };
  for (size_t k = 0; k < nout; ++k) {
    for (int i = 0; i < n; ++i) {
      T x = obs[i][k];
      T floc = std::floor((x - starts[i]) / steps[i]);  // regular.rs:415
      int64_t iloc;
      if (!to_isize(floc, &iloc)) { *first_bad = k; return ERR_UNREPRESENTABLE; }  // regular.rs:418
      int64_t nn = (int64_t)dims[i];
      int64_t dimmax = nn - 2 > 0 ? nn - 2 : 0;
      int64_t loc = iloc > 0 ? iloc : 0;
      loc = loc < dimmax ? loc : dimmax;  // regular.rs:420-422
      p.origin[i] = (size_t)loc;
      T origin_f = (T)p.origin[i];  // regular.rs:330
      T index_zero_loc = fma_index ? std::fma(steps[i], origin_f, starts[i])
                                   : mul_add<false>(steps[i], origin_f, starts[i]);
      p.dts[i] = (x - index_zero_loc) / steps[i];  // regular.rs:339
    }
    out[k] = populate<decltype(node), 2, T>(n, origin_base(p, n), p, vals, node);
  }
  return OK;
}

// multilinear::rectilinear — src/multilinear/rectilinear.rs:49-83, :175-201, :210-231,
// :244-370; N=7,8: rectilinear_recursive.rs:160-337.
template <typename T, bool FMA>
int linear_rectilinear(const T* const* grids, const size_t* grid_lens, size_t ngrids, const T* vals,
                       size_t nvals, const T* const* obs, const size_t* obs_lens, size_t nobs, T* out,
                       size_t nout, size_t* first_bad) {
  (void)first_bad;
  const size_t ndims = ngrids;
  if (nobs != ndims) return ERR_DIM_MISMATCH;                          // rectilinear.rs:59
  if (ndims < 1 || ndims > (size_t)MAXDIMS) return ERR_TOO_MANY_DIMS;  // rectilinear.rs:77-79
  const int n = (int)ndims;
  size_t prod;
  if (!checked_product(grid_lens, ndims, &prod)) return ERR_REFERENCE_PANIC;
  if (nvals != prod) return ERR_DIM_MISMATCH;                                               // rectilinear.rs:186
  for (int i = 0; i < n; ++i) if (grid_lens[i] < 2) return ERR_MIN_2_ENTRIES;               // rectilinear.rs:190
  for (int i = 0; i < n; ++i) if (!(grids[i][1] > grids[i][0])) return ERR_NOT_MONOTONIC;    // rectilinear.rs:195
  for (int i = 0; i < n; ++i) if (obs_lens[i] != nout) return ERR_DIM_MISMATCH;              // rectilinear.rs:219

  Point<T> p;
  fill_dimprod(p, grid_lens, n);
  auto node = [grids](const T* v, int d, const Point<T>& q) -> T {
    T x0 = grids[d][q.origin[d]];
    T x1 = grids[d][q.origin[d] + 1];
    T step = x1 - x0;
    T t = (q.x[d] - x0) / step;  // rectilinear.rs:310-313 (recomputed at every node)
    T y0 = v[0];
    T dy = v[1] - y0;
    return mul_add<FMA>(t, dy, y0);  // rectilinear.rs:318-321
  };
  for (size_t k = 0; k < nout; ++k) {
    for (int i = 0; i < n; ++i) {
      T x = obs[i][k];
      p.x[i] = x;
      int64_t iloc = (int64_t)partition_point_lt(grids[i], grid_lens[i], x) - 1;  // rectilinear.rs:363
      int64_t nn = (int64_t)grid_lens[i];
      int64_t dimmax = nn - 2 > 0 ? nn - 2 : 0;
      int64_t loc = iloc > 0 ? iloc : 0;
      loc = loc < dimmax ? loc : dimmax;  // rectilinear.rs:365-367
      p.origin[i] = (size_t)loc;
    }
    out[k] = populate<decltype(node), 2, T>(n, origin_base(p, n), p, vals, node);
  }
  return OK;
}

// multicubic::regular — src/multicubic/regular.rs:52-136, :239-288, :297-313, :325-469;
// N=5..8: regular_recursive.rs:236-283, :318-427.
template <typename T, bool FMA>
int cubic_regular(const size_t* dims, size_t ndims, const T* starts, size_t nstarts, const T* steps,
                  size_t nsteps, const T* vals, size_t nvals, int linearize, const T* const* obs,
                  const size_t* obs_lens, size_t nobs, T* out, size_t nout, size_t* first_bad) {
  if (ndims < 1 || ndims > (size_t)MAXDIMS) return ERR_TOO_MANY_DIMS;  // regular.rs:130-132
  const int n = (int)ndims;
  // regular.rs:66-73: `starts.try_into().unwrap()` / `steps.try_into().unwrap()` panic on a
  // length mismatch before `new` runs; `obs.try_into().unwrap()` panics after `new(..)?`.
  if (n <= 4 && (nstarts != ndims || nsteps != ndims)) return ERR_REFERENCE_PANIC;
  size_t prod;
  if (!checked_product(dims, ndims, &prod)) return ERR_REFERENCE_PANIC;
  if (!(nstarts == ndims && nsteps == ndims && nvals == prod)) return ERR_DIM_MISMATCH;  // regular.rs:254
  for (int i = 0; i < n; ++i) if (dims[i] < 4) return ERR_MIN_FOUR_ENTRIES;              // regular.rs:259
  for (int i = 0; i < n; ++i) if (!(steps[i] > (T)0)) return ERR_NOT_MONOTONIC;          // regular.rs:264
  if (n <= 4 && nobs != ndims) return ERR_REFERENCE_PANIC;                       // regular.rs:73
  if (nobs != ndims) return ERR_DIM_MISMATCH;                                    // regular_recursive.rs interp
  for (int i = 0; i < n; ++i) if (obs_lens[i] != nout) return ERR_DIM_MISMATCH;  // regular.rs:301

  Point<T> p;
  fill_dimprod(p, dims, n);
  const bool lin = linearize != 0;
  const bool recursive_arm = n > 4;  // regular.rs:52-136: N = 1..4 flattened, above that the recursive arm
  auto node = [lin, recursive_arm](const T* v, int d, const Point<T>& q) -> T {
    return cubic_regular_inner<FMA>(v, q.dts[d], q.sat[d], lin, recursive_arm);
  };
  for (size_t k = 0; k < nout; ++k) {
    for (int i = 0; i < n; ++i) {
      T x = obs[i][k];
      T floc = std::floor((x - starts[i]) / steps[i]);  // regular.rs:435
      int64_t iloc;
      if (!to_isize(floc, &iloc)) { *first_bad = k; return ERR_UNREPRESENTABLE; }  // regular.rs:438
      if (iloc == INT64_MIN) { *first_bad = k; return ERR_REFERENCE_PANIC; }        // `- 1` overflows => panic
      iloc -= 1;
      int64_t nn = (int64_t)dims[i];
      int64_t dimmax = nn - 4 > 0 ? nn - 4 : 0;
      int64_t loc = iloc > 0 ? iloc : 0;
      loc = loc < dimmax ? loc : dimmax;  // regular.rs:440-442
      p.origin[i] = (size_t)loc;
      Sat s;  // regular.rs:445-466
      if (iloc < -1) s = OutsideLow;
      else if (iloc == -1) s = InsideLow;
      else if (iloc > nn - 3) s = OutsideHigh;
      else if (iloc == nn - 3) s = InsideHigh;
      else s = SatNone;
      p.sat[i] = s;
      // regular.rs:356-360 — never fused, in either arm.
      T index_one_loc = mul_add<false>(steps[i], (T)(p.origin[i] + 1), starts[i]);
      p.dts[i] = (x - index_one_loc) / steps[i];
    }
    out[k] = populate<decltype(node), 4, T>(n, origin_base(p, n), p, vals, node);
  }
  return OK;
}

// multicubic::rectilinear — src/multicubic/rectilinear.rs:54-104, :193-228, :237-253,
// :265-408; N=5..8: rectilinear_recursive.rs:169-201, :294-380.
template <typename T, bool FMA>
int cubic_rectilinear(const T* const* grids, const size_t* grid_lens, size_t ngrids, const T* vals,
                      size_t nvals, int linearize, const T* const* obs, const size_t* obs_lens, size_t nobs,
                      T* out, size_t nout, size_t* first_bad) {
  (void)first_bad;
  const size_t ndims = ngrids;
  if (ndims < 1 || ndims > (size_t)MAXDIMS) return ERR_TOO_MANY_DIMS;  // rectilinear.rs:98-100
  const int n = (int)ndims;
  size_t prod;
  if (!checked_product(grid_lens, ndims, &prod)) return ERR_REFERENCE_PANIC;
  if (nvals != prod) return ERR_DIM_MISMATCH;                                               // rectilinear.rs:208
  for (int i = 0; i < n; ++i) if (grid_lens[i] < 4) return ERR_MIN_4_ENTRIES;               // rectilinear.rs:212
  for (int i = 0; i < n; ++i) if (!(grids[i][1] > grids[i][0])) return ERR_NOT_MONOTONIC;    // rectilinear.rs:217
  if (n <= 4 && nobs != ndims) return ERR_REFERENCE_PANIC;  // rectilinear.rs:71 obs.try_into().unwrap()
  if (nobs != ndims) return ERR_DIM_MISMATCH;                                    // rectilinear_recursive.rs:211
  for (int i = 0; i < n; ++i) if (obs_lens[i] != nout) return ERR_DIM_MISMATCH;  // rectilinear.rs:241

  Point<T> p;
  fill_dimprod(p, grid_lens, n);
  const bool lin = linearize != 0;
  const bool fma_linearize = n >= 5;  // recursive arm only
  auto node = [grids, lin, fma_linearize](const T* v, int d, const Point<T>& q) -> T {
    return cubic_rectilinear_inner<FMA>(v, grids[d] + q.origin[d], q.x[d], q.sat[d], lin, fma_linearize);
  };
  for (size_t k = 0; k < nout; ++k) {
    for (int i = 0; i < n; ++i) {
      T x = obs[i][k];
      p.x[i] = x;
      int64_t iloc = (int64_t)partition_point_lt(grids[i], grid_lens[i], x) - 2;  // rectilinear.rs:377
      int64_t nn = (int64_t)grid_lens[i];
      int64_t dimmax = nn - 4 > 0 ? nn - 4 : 0;
      int64_t loc = iloc > 0 ? iloc : 0;
      loc = loc < dimmax ? loc : dimmax;  // rectilinear.rs:379-381
      p.origin[i] = (size_t)loc;
      Sat s;  // rectilinear.rs:384-405
      if (iloc == -2) s = OutsideLow;
      else if (iloc == -1) s = InsideLow;
      else if (iloc == nn - 2) s = OutsideHigh;
      else if (iloc == nn - 3) s = InsideHigh;
      else s = SatNone;
      p.sat[i] = s;
    }
    out[k] = populate<decltype(node), 4, T>(n, origin_base(p, n), p, vals, node);
  }
  return OK;
}

// nearest::regular — src/nearest/regular.rs:41-101 (dispatch, N = 1..6), :163-195 (new),
// :206-222 (interp), :234-317 (interp_one / get_loc).
template <typename T, bool FMA>
int nearest_regular(const size_t* dims, size_t ndims, const T* starts, size_t nstarts, const T* steps,
                    size_t nsteps, const T* vals, size_t nvals, const T* const* obs, const size_t* obs_lens,
                    size_t nobs, T* out, size_t nout, size_t* first_bad) {
  if (nstarts != ndims || nsteps != ndims || nobs != ndims) return ERR_DIM_MISMATCH;  // regular.rs:50
  if (ndims < 1 || ndims > 6) return ERR_TOO_MANY_DIMS_6;                              // regular.rs:97
  const int n = (int)ndims;
  size_t prod;
  if (!checked_product(dims, ndims, &prod)) return ERR_REFERENCE_PANIC;
  if (nvals != prod) return ERR_DIM_MISMATCH;                                   // regular.rs:177
  for (int i = 0; i < n; ++i) if (dims[i] < 2) return ERR_MIN_TWO_ENTRIES;      // regular.rs:182
  for (int i = 0; i < n; ++i) if (!(steps[i] > (T)0)) return ERR_NOT_MONOTONIC;  // regular.rs:187
  for (int i = 0; i < n; ++i) if (obs_lens[i] != nout) return ERR_DIM_MISMATCH;  // regular.rs:210
  Point<T> p;
  fill_dimprod(p, dims, n);
  const T half = (T)1 / ((T)1 + (T)1);
  for (size_t k = 0; k < nout; ++k) {
    size_t idx = 0;
    for (int i = 0; i < n; ++i) {
      T x = obs[i][k];
      T floc = std::floor((x - starts[i]) / steps[i]);  // regular.rs:306
      int64_t iloc;
      if (!to_isize(floc, &iloc)) { *first_bad = k; return ERR_UNREPRESENTABLE; }
      int64_t nn = (int64_t)dims[i];
      int64_t dimmax = nn - 2 > 0 ? nn - 2 : 0;
      int64_t loc = iloc > 0 ? iloc : 0;
      loc = loc < dimmax ? loc : dimmax;
      T origin_f = (T)loc;
      T index_zero_loc = mul_add<FMA>(steps[i], origin_f, starts[i]);  // regular.rs:272-275
      T dt = (x - index_zero_loc) / steps[i];
      size_t offset = (dt <= half) ? 0 : 1;  // regular.rs:283-287 (NaN -> 1)
      idx += ((size_t)loc + offset) * p.dimprod[i];
    }
    out[k] = vals[idx];
  }
  return OK;
}

// nearest::rectilinear — src/nearest/rectilinear.rs:36-60, :120-147 (new), :193-246 (interp_one).
template <typename T>
int nearest_rectilinear(const T* const* grids, const size_t* grid_lens, size_t ngrids, const T* vals, size_t nvals,
                        const T* const* obs, const size_t* obs_lens, size_t nobs, T* out, size_t nout) {
  const size_t ndims = ngrids;
  if (nobs != ndims) return ERR_DIM_MISMATCH;
  if (ndims < 1 || ndims > 6) return ERR_TOO_MANY_DIMS_6;
  const int n = (int)ndims;
  size_t prod;
  if (!checked_product(grid_lens, ndims, &prod)) return ERR_REFERENCE_PANIC;
  if (nvals != prod) return ERR_DIM_MISMATCH;
  for (int i = 0; i < n; ++i) if (grid_lens[i] < 2) return ERR_MIN_2_ENTRIES;
  for (int i = 0; i < n; ++i) if (!(grids[i][1] > grids[i][0])) return ERR_NOT_MONOTONIC;
  for (int i = 0; i < n; ++i) if (obs_lens[i] != nout) return ERR_DIM_MISMATCH;
  Point<T> p;
  fill_dimprod(p, grid_lens, n);
  const T half = (T)1 / ((T)1 + (T)1);
  for (size_t k = 0; k < nout; ++k) {
    size_t idx = 0;
    for (int i = 0; i < n; ++i) {
      T x = obs[i][k];
      int64_t iloc = (int64_t)partition_point_lt(grids[i], grid_lens[i], x) - 1;  // rectilinear.rs:259
      int64_t nn = (int64_t)grid_lens[i];
      int64_t dimmax = nn - 2 > 0 ? nn - 2 : 0;
      int64_t loc = iloc > 0 ? iloc : 0;
      loc = loc < dimmax ? loc : dimmax;
      T x0 = grids[i][loc];
      T x1 = grids[i][loc + 1];
      T step = x1 - x0;
      T dt = (x - x0) / step;  // rectilinear.rs:223-227
      size_t offset = (dt <= half) ? 0 : 1;
      idx += ((size_t)loc + offset) * p.dimprod[i];
    }
    out[k] = vals[idx];
  }
  return OK;
}

// check_bounds — src/multilinear/regular.rs:145-182, rectilinear.rs:109-134.
template <typename T>
int check_bounds_regular(const size_t* dims, size_t ndims, const T* starts, const T* steps,
                         const T* const* obs, const size_t* obs_lens, size_t nobs, T atol, uint8_t* out,
                         size_t nout) {
  if (!(nobs == ndims && nout == ndims)) return ERR_DIM_MISMATCH;
  for (size_t i = 0; i < ndims; ++i) {
    T first = starts[i];
    T last = starts[i] + steps[i] * (T)(dims[i] - 1);
    T lo = std::fmin(first, last);
    T hi = std::fmax(first, last);
    bool bad = false;
    for (size_t k = 0; k < obs_lens[i]; ++k) {
      T x = obs[i][k];
      if ((x - lo) <= -atol || (x - hi) >= atol) { bad = true; break; }
    }
    out[i] = bad ? 1 : 0;
  }
  return OK;
}

template <typename T>
int check_bounds_rectilinear(const T* const* grids, const size_t* grid_lens, size_t ngrids,
                             const T* const* obs, const size_t* obs_lens, size_t nobs, T atol, uint8_t* out,
                             size_t nout) {
  const size_t ndims = ngrids;
  if (!(nobs == ndims && nout == ndims)) return ERR_DIM_MISMATCH;
  for (size_t i = 0; i < ndims; ++i) if (grid_lens[i] == 0) return ERR_DIM_MISMATCH;
  for (size_t i = 0; i < ndims; ++i) {
    T lo = grids[i][0];
    T hi = grids[i][grid_lens[i] - 1];
    bool bad = false;
    for (size_t k = 0; k < obs_lens[i]; ++k) {
      T x = obs[i][k];
      if ((x - lo) <= -atol || (x - hi) >= atol) { bad = true; break; }
    }
    out[i] = bad ? 1 : 0;
  }
  return OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// C entry points.  `fma` selects the reference's cargo feature (1 = what the
// published wheels build with, pyproject.toml:72; 0 = plain `cargo test`).
// `first_bad` receives the index of the first failing observation point when the
// status is ERR_UNREPRESENTABLE (out[0..first_bad) is written, the rest untouched).
extern "C" {

#define ORACLE_DEFINE(T, SUFFIX)                                                                          \
  int oracle_linear_regular_##SUFFIX(int fma, const size_t* dims, size_t ndims, const T* starts,         \
                                     size_t nstarts, const T* steps, size_t nsteps, const T* vals,       \
                                     size_t nvals, const T* const* obs, const size_t* obs_lens,          \
                                     size_t nobs, T* out, size_t nout, size_t* first_bad) {              \
    return fma ? linear_regular<T, true>(dims, ndims, starts, nstarts, steps, nsteps, vals, nvals, obs,  \
                                         obs_lens, nobs, out, nout, first_bad)                           \
               : linear_regular<T, false>(dims, ndims, starts, nstarts, steps, nsteps, vals, nvals, obs, \
                                          obs_lens, nobs, out, nout, first_bad);                         \
  }                                                                                                       \
  int oracle_linear_rectilinear_##SUFFIX(int fma, const T* const* grids, const size_t* grid_lens,        \
                                         size_t ngrids, const T* vals, size_t nvals, const T* const* obs, \
                                         const size_t* obs_lens, size_t nobs, T* out, size_t nout,       \
                                         size_t* first_bad) {                                            \
    return fma ? linear_rectilinear<T, true>(grids, grid_lens, ngrids, vals, nvals, obs, obs_lens, nobs, \
                                             out, nout, first_bad)                                       \
               : linear_rectilinear<T, false>(grids, grid_lens, ngrids, vals, nvals, obs, obs_lens,      \
                                              nobs, out, nout, first_bad);                               \
  }                                                                                                       \
  int oracle_cubic_regular_##SUFFIX(int fma, const size_t* dims, size_t ndims, const T* starts,          \
                                    size_t nstarts, const T* steps, size_t nsteps, const T* vals,        \
                                    size_t nvals, int linearize, const T* const* obs,                    \
                                    const size_t* obs_lens, size_t nobs, T* out, size_t nout,            \
                                    size_t* first_bad) {                                                 \
    return fma ? cubic_regular<T, true>(dims, ndims, starts, nstarts, steps, nsteps, vals, nvals,        \
                                        linearize, obs, obs_lens, nobs, out, nout, first_bad)            \
               : cubic_regular<T, false>(dims, ndims, starts, nstarts, steps, nsteps, vals, nvals,       \
                                         linearize, obs, obs_lens, nobs, out, nout, first_bad);          \
  }                                                                                                       \
  int oracle_cubic_rectilinear_##SUFFIX(int fma, const T* const* grids, const size_t* grid_lens,         \
                                        size_t ngrids, const T* vals, size_t nvals, int linearize,       \
                                        const T* const* obs, const size_t* obs_lens, size_t nobs,        \
                                        T* out, size_t nout, size_t* first_bad) {                        \
    return fma ? cubic_rectilinear<T, true>(grids, grid_lens, ngrids, vals, nvals, linearize, obs,       \
                                            obs_lens, nobs, out, nout, first_bad)                        \
               : cubic_rectilinear<T, false>(grids, grid_lens, ngrids, vals, nvals, linearize, obs,      \
                                             obs_lens, nobs, out, nout, first_bad);                      \
  }                                                                                                       \
  int oracle_nearest_regular_##SUFFIX(int fma, const size_t* dims, size_t ndims, const T* starts,        \
                                      size_t nstarts, const T* steps, size_t nsteps, const T* vals,      \
                                      size_t nvals, const T* const* obs, const size_t* obs_lens,         \
                                      size_t nobs, T* out, size_t nout, size_t* first_bad) {             \
    return fma ? nearest_regular<T, true>(dims, ndims, starts, nstarts, steps, nsteps, vals, nvals, obs, \
                                          obs_lens, nobs, out, nout, first_bad)                          \
               : nearest_regular<T, false>(dims, ndims, starts, nstarts, steps, nsteps, vals, nvals,     \
                                           obs, obs_lens, nobs, out, nout, first_bad);                   \
  }                                                                                                       \
  int oracle_nearest_rectilinear_##SUFFIX(int fma, const T* const* grids, const size_t* grid_lens,       \
                                          size_t ngrids, const T* vals, size_t nvals,                    \
                                          const T* const* obs, const size_t* obs_lens, size_t nobs,      \
                                          T* out, size_t nout, size_t* first_bad) {                      \
    (void)fma; (void)first_bad;                                                                          \
    return nearest_rectilinear<T>(grids, grid_lens, ngrids, vals, nvals, obs, obs_lens, nobs, out, nout); \
  }                                                                                                       \
  int oracle_check_bounds_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts,            \
                                           const T* steps, const T* const* obs, const size_t* obs_lens,  \
                                           size_t nobs, T atol, uint8_t* out, size_t nout) {             \
    return check_bounds_regular<T>(dims, ndims, starts, steps, obs, obs_lens, nobs, atol, out, nout);    \
  }                                                                                                       \
  int oracle_check_bounds_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens,           \
                                               size_t ngrids, const T* const* obs,                       \
                                               const size_t* obs_lens, size_t nobs, T atol,              \
                                               uint8_t* out, size_t nout) {                              \
    return check_bounds_rectilinear<T>(grids, grid_lens, ngrids, obs, obs_lens, nobs, atol, out, nout);  \
  }

ORACLE_DEFINE(double, f64)
ORACLE_DEFINE(float, f32)

// Error text, identical to the reference's &'static str values.
const char* oracle_strerror(int status) {
  switch (status) {
    case OK: return "";
    case ERR_DIM_MISMATCH: return "Dimension mismatch";
    case ERR_MIN_TWO_ENTRIES: return "All grids must have at least two entries";
    case ERR_MIN_2_ENTRIES: return "All grids must have at least 2 entries";
    case ERR_MIN_FOUR_ENTRIES: return "All grids must have at least four entries";
    case ERR_MIN_4_ENTRIES: return "All grids must have at least 4 entries";
    case ERR_NOT_MONOTONIC: return "All grids must be monotonically increasing";
    case ERR_UNREPRESENTABLE: return "Unrepresentable coordinate value";
    case ERR_TOO_MANY_DIMS:
      return "Dimension exceeds maximum (8). Use interpolator struct directly for higher dimensions.";
    case ERR_REFERENCE_PANIC: return "reference would panic (slice length / integer overflow)";
    case ERR_TOO_MANY_DIMS_6: return "Dimension exceeds maximum (6).";
    default: return "unknown status";
  }
}

}  // extern "C"
