// Device-side building blocks of the MI355X (gfx950) interpolation kernels.
//
// One wavefront lane evaluates one observation point.  Observation coordinates are
// struct-of-arrays (one contiguous array per dimension, as the reference's `obs: &[&[T]]`,
// src/multilinear/regular.rs:56), so a wave reads each coordinate as one coalesced 512-B
// (f64) request.  Per-dimension cell index / normalized coordinate / saturation class live
// in registers; the 2^N (linear) or 4^N (cubic) corner values are gathered from the
// read-only grid, which is small enough to stay resident in the XCD L2 / Infinity Cache.
// Along the last (stride-1) dimension the footprint is contiguous in memory, so corners are
// fetched as 16-B (linear f64) / 32-B (cubic f64) vectors: 2^(N-1) resp. 4^(N-1) gathers
// per point instead of 2^N / 4^N.
//
// Numerics: every operation below is written in the order of the reference source, and the
// translation unit is compiled with -ffp-contract=off, so the only fused multiply-adds are
// the explicit ones at the reference's `#[cfg(feature = "fma")]` sites.  f64/f32 division,
// floor and fma are IEEE-correct on CDNA4, so results are bit-identical to the Rust code
// built with the same feature set.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cubic_cell_record.h"

namespace interpn {

constexpr int kMaxDims = 8;  // src/python.rs:10
constexpr unsigned long long kNoBadIndex = ~0ull;

enum Sat : int { kSatNone = 0, kSatLow = 1, kSatHigh = 2 };  // src/multicubic/mod.rs:59-66 (Inside/Outside kept apart in `outside`)

// Order LDS traffic between the lanes of ONE wave (exchange patterns where lane A stores and
// lane B loads).  The hardware executes a wave's DS instructions in order; what must be stopped is
// the compiler moving accesses across the exchange point: llvm.amdgcn.wave.barrier alone is
// IntrNoMem, so it is paired with a wavefront-scope fence.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// Coordinates are read once and results written once: mark them non-temporal so that the L2 treats
// them as streams.  Measured with 16-B-per-lane accesses: 3-D linear 64^3 1.48 -> 1.35 ms.
template <typename V>
__device__ __forceinline__ V stream_load(const V* p) { return __builtin_nontemporal_load(p); }
template <typename V>
__device__ __forceinline__ void stream_store(V* p, V v) { __builtin_nontemporal_store(v, p); }

// LDS words that are also accessed through another element type (type-based alias analysis off).
typedef unsigned __attribute__((may_alias)) lds_u32;

// made-up coordinates of the measurement builds (tools/: ABL forms of the brick and sweep kernels)
template <typename T>
__device__ __forceinline__ T ablate_coord(size_t i, int d, T start, T step, int n) {
  unsigned h = (unsigned)i * 2654435761u + (unsigned)(i >> 32) * 40503u + (unsigned)d * 0x9E3779B9u;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return start + step * ((T)(n - 1) * ((T)(h >> 8) * (T)(1.0 / 16777216.0)));
}


// ---------------------------------------------------------------------------
// Scalar helpers
template <typename T> __device__ __forceinline__ T dev_fma(T a, T b, T c);
template <> __device__ __forceinline__ double dev_fma<double>(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <> __device__ __forceinline__ float dev_fma<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

template <typename T> __device__ __forceinline__ T dev_floor(T a);
template <> __device__ __forceinline__ double dev_floor<double>(double a) { return __builtin_floor(a); }
template <> __device__ __forceinline__ float dev_floor<float>(float a) { return __builtin_floorf(a); }

// a*b + c with one rounding when FMA (Float::mul_add), two otherwise.
template <bool FMA, typename T>
__device__ __forceinline__ T mul_add(T a, T b, T c) {
  if constexpr (FMA) {
    return dev_fma<T>(a, b, c);
  } else {
    T p = a * b;
    return p + c;
  }
}

// Vector leaf types: W contiguous elements of the last dimension, element-aligned only.
template <typename T, int W> struct LeafVec;
template <> struct LeafVec<double, 2> { typedef double type __attribute__((ext_vector_type(2), aligned(8))); };
template <> struct LeafVec<double, 4> { typedef double type __attribute__((ext_vector_type(4), aligned(8))); };
template <> struct LeafVec<float, 2> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };
template <> struct LeafVec<float, 4> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); };

template <typename T, int W>
struct Leaf {
  T v[W];
};

template <typename T, int W>
__device__ __forceinline__ Leaf<T, W> load_leaf(const T* __restrict__ p) {
  typedef typename LeafVec<T, W>::type V;
  V x = *reinterpret_cast<const V*>(p);
  Leaf<T, W> r;
#pragma unroll
  for (int i = 0; i < W; ++i) r.v[i] = x[i];
  return r;
}

// ---------------------------------------------------------------------------
// Cell location
//
// Regular grid, linear: src/multilinear/regular.rs:414-425.  Returns false when the
// reference's float->isize conversion fails (NaN, +-inf, |floc| >= 2^63).
// `dimmax` = dims-2 (linear) or dims-4 (cubic); host guarantees 0 <= dimmax < 2^31-256.
template <typename T>
__device__ __forceinline__ bool regular_floc(T x, T start, T step, T* floc_out) {
  T floc = dev_floor<T>((x - start) / step);
  *floc_out = floc;
  // num-traits 0.2.19 <isize as NumCast>::from: Some iff -2^63 <= f < 2^63
  return (floc >= (T)-9223372036854775808.0) && (floc < (T)9223372036854775808.0);
}

template <typename T>
__device__ __forceinline__ int clamp_loc(T floc_shifted, int dimmax) {
  // iloc.max(0).min(dimmax): done in the float domain up to 2^31, then in int.
  const T big = sizeof(T) == 8 ? (T)2147483647.0 : (T)2147483520.0;
  T c = floc_shifted;
  c = c > (T)0 ? c : (T)0;  // NaN never reaches here (caller checked)
  c = c < big ? c : big;
  int li = (int)c;
  return li < dimmax ? li : dimmax;
}

// ---------------------------------------------------------------------------
// Cell index and normalized coordinate on a regular grid WITHOUT the divide sequences, bit for bit
// what regular_floc / clamp_loc / `(x - izl) / step` give (f64; round 5).
//
// Why: an IEEE f64 division is ~14 issue slots here (2 v_div_scale, v_rcp_f64 at quarter rate, 5 fma,
// v_mul, v_div_fmas, v_div_fixup) and a 3-D multilinear point makes six of them — half of the vector
// instructions of a row of the sweep kernel, which holds three waves per SIMD and therefore does not
// hide them (tools/sweep_clock_probe.py, measurement builds: 1.03 ms per 1e8 points with the
// divisions, 0.93 with inexact reciprocal multiplications, 0.72 without any table access).  The
// divisor is the grid's step: the same for every point, so the host supplies rb = RN(1 / b) (one
// correctly rounded IEEE division), and:
//
//  * t = RN(a / b), a = x - izl (multilinear/regular.rs:339): q0 = RN(a rb); r0 = RN(a - b q0);
//    q1 = RN(q0 + r0 rb); r1 = a - b q1 (exact); t = RN(q1 + r1 rb).  q0 is within 2^-52 |a / b| of the
//    quotient, so q1 is computed from a value within 2^-104 |a / b| of it and is one of the two
//    neighbours of a / b (a faithful rounding); the remainder of a faithful quotient is exactly
//    representable and fma delivers it; and for a faithful q, its exact remainder r and the correctly
//    rounded reciprocal rb, RN(q + r rb) IS the correctly rounded quotient (Markstein 1990, "Computation
//    of elementary functions on the IBM RISC System/6000 processor", theorem on the final step of
//    fma-based division; Muller et al., Handbook of Floating-Point Arithmetic, section on
//    Newton-Raphson division with an fma: a / b lies at least 2^-p ulp / |b'| (b' = b scaled into
//    [1, 2)) away from every midpoint of two neighbours, the error of q + r rb against a / b is
//    |r / b| |b rb - 1| <= (ulp / 2 + gap) b' 2^-p-1, which is below that gap for every b' < 2).
//    Holds while nothing overflows, underflows or is not finite: the fast form is taken for
//    2^-128 <= |b| <= 2^128 (host, per handle) and 2^-256 <= |a| < 2^256 (per point: the exponent
//    field; a = 0 — a point on a grid plane —, infinities and NaN fail it): then 2^-385 < |a / b| <
//    2^385 and every remainder is a multiple of 2^(-128 - 52 - 385 - 52), far above the subnormals.
//  * the cell index floor(RN(a0 / b)), a0 = x - start (regular.rs:415-422): qt = RN(a0 rb) lies within
//    2^-51.6 |a0 / b| of RN(a0 / b); for |qt| < 2^31 that is less than 2^-20, so if qt is further than
//    2^-20 from every integer the two have the same floor.  Otherwise (2e-6 of random points; every
//    point on a grid plane) the lane asks for the divisions.  |qt| < 2^31 also makes the reference's
//    "Unrepresentable coordinate value" check (|floc| >= 2^63, NaN) pass by construction.
// A lane for which a condition fails reports exact = false and the caller runs the divide sequences
// for its wave (a wave-uniform branch); results are the reference's either way.
template <typename T>
struct StepCell {
  T t;
  int loc;
  bool exact;
};

__device__ __forceinline__ bool exponent_within_256(double a) {
  const unsigned e = ((unsigned)__double2hiint(a) >> 20) & 0x7FFu;
  return e - (1023u - 256u) < 512u;
}

template <bool FMA>
__device__ __forceinline__ StepCell<double> step_cell_fast(double x, double start, double step, double rstep, int dimmax) {
  StepCell<double> r;
  const double qt = (x - start) * rstep;
  const double d = __builtin_amdgcn_fract(qt);  // qt - floor(qt): exact for |qt| < 2^31 (0.999.. for a tiny negative qt: rejected below)
  r.exact = (__builtin_fabs(d - 0.5) < 0.5 - 0x1p-20) && (__builtin_fabs(qt) < 0x1p31);  // (NaN: false)
  // clamp_loc: iloc.max(0).min(dimmax), regular.rs:420-422.  Truncation serves as floor here: they differ for
  // negative qt only, which the clamp takes to 0 either way.  (Where |qt| >= 2^31 or qt is NaN the conversion's
  // value is not used: `exact` is false and the caller replaces everything derived from it.)
  int li = (int)qt;
  li = li > 0 ? li : 0;
  li = li < dimmax ? li : dimmax;
  r.loc = li;
  const double lf = (double)li;
  const double izl = mul_add<FMA>(step, lf, start);    // regular.rs:334-337 ((T)loc == lf: an integer in [0, dimmax])
  const double a = x - izl;                            // regular.rs:339, the dividend
  r.exact = r.exact && exponent_within_256(a);
  const double q0 = a * rstep;
  const double r0 = __builtin_fma(-step, q0, a);
  const double q1 = __builtin_fma(r0, rstep, q0);
  const double r1 = __builtin_fma(-step, q1, a);
  r.t = __builtin_fma(r1, rstep, q1);
  return r;
}

// The two halves of step_cell_fast on their own (linear_sweep.h: the cell — what the gather's address needs — in front of
// the loads, the quotient — what only the lerps need — behind them): identical operations, identical bits.
template <typename T>
struct StepCellIndex {
  T a;       // the dividend x - izl
  int loc;
  bool exact;
};
template <bool FMA>
__device__ __forceinline__ StepCellIndex<double> step_cell_index(double x, double start, double step, double rstep, int dimmax) {
  StepCellIndex<double> r;
  const double qt = (x - start) * rstep;
  const double d = __builtin_amdgcn_fract(qt);
  r.exact = (__builtin_fabs(d - 0.5) < 0.5 - 0x1p-20) && (__builtin_fabs(qt) < 0x1p31);
  int li = (int)qt;
  li = li > 0 ? li : 0;
  li = li < dimmax ? li : dimmax;
  r.loc = li;
  const double izl = mul_add<FMA>(step, (double)li, start);
  r.a = x - izl;
  r.exact = r.exact && exponent_within_256(r.a);
  return r;
}
__device__ __forceinline__ double step_cell_quotient(double a, double step, double rstep) {
  const double q0 = a * rstep;
  const double r0 = __builtin_fma(-step, q0, a);
  const double q1 = __builtin_fma(r0, rstep, q0);
  const double r1 = __builtin_fma(-step, q1, a);
  return __builtin_fma(r1, rstep, q1);
}

// f32: the same forms with p = 24.  Admitted: 2^-16 <= |step| <= 2^16 (host) and 2^-24 <= |x - izl| < 2^24 (per
// point): quotients in (2^-41, 2^41), remainders multiples of 2^(-16 - 23 - 41 - 23) — normal numbers.  The
// cell index: RN(a0 rb) lies within 2^-22.4 |a0 / b| of RN(a0 / b); the short form is kept while qt is further
// than 2^-21 (|qt| + 1) from every integer, which also bounds |qt| below 2^20 (and rejects NaN and infinities).
__device__ __forceinline__ bool exponent_within_24(float a) {
  const unsigned e = (__float_as_uint(a) >> 23) & 0xFFu;
  return e - (127u - 24u) < 48u;
}

template <bool FMA>
__device__ __forceinline__ StepCell<float> step_cell_fast(float x, float start, float step, float rstep, int dimmax) {
  StepCell<float> r;
  const float qt = (x - start) * rstep;
  const float d = __builtin_amdgcn_fractf(qt);
  const float margin = __builtin_fmaf(__builtin_fabsf(qt), 0x1p-21f, 0x1p-21f);
  r.exact = __builtin_fabsf(d - 0.5f) + margin < 0.5f;  // (NaN, infinities, |qt| >= 2^20: false)
  int li = (int)qt;  // see the f64 form
  li = li > 0 ? li : 0;
  li = li < dimmax ? li : dimmax;
  r.loc = li;
  const float lf = (float)li;
  const float izl = mul_add<FMA>(step, lf, start);
  const float a = x - izl;
  r.exact = r.exact && exponent_within_24(a);
  const float q0 = a * rstep;
  const float r0 = __builtin_fmaf(-step, q0, a);
  const float q1 = __builtin_fmaf(r0, rstep, q0);
  const float r1 = __builtin_fmaf(-step, q1, a);
  r.t = __builtin_fmaf(r1, rstep, q1);
  return r;
}

template <bool FMA>
__device__ __forceinline__ StepCellIndex<float> step_cell_index(float x, float start, float step, float rstep, int dimmax) {
  StepCellIndex<float> r;
  const float qt = (x - start) * rstep;
  const float d = __builtin_amdgcn_fractf(qt);
  const float margin = __builtin_fmaf(__builtin_fabsf(qt), 0x1p-21f, 0x1p-21f);
  r.exact = __builtin_fabsf(d - 0.5f) + margin < 0.5f;
  int li = (int)qt;
  li = li > 0 ? li : 0;
  li = li < dimmax ? li : dimmax;
  r.loc = li;
  const float izl = mul_add<FMA>(step, (float)li, start);
  r.a = x - izl;
  r.exact = r.exact && exponent_within_24(r.a);
  return r;
}
__device__ __forceinline__ float step_cell_quotient(float a, float step, float rstep) {
  const float q0 = a * rstep;
  const float r0 = __builtin_fmaf(-step, q0, a);
  const float q1 = __builtin_fmaf(r0, rstep, q0);
  const float r1 = __builtin_fmaf(-step, q1, a);
  return __builtin_fmaf(r1, rstep, q1);
}

// The two short forms on their own (cubic_sweep.h: the saturation class of a point needs floor(RN(a0 / b)) itself,
// not only the clamped cell index).  Same conditions, same proofs as in step_cell_fast; the return value says
// whether the result is the reference's (false: the caller must use the divide sequence).
__device__ __forceinline__ bool floor_quotient_fast(double a0, double rb, double* floc) {
  const double qt = a0 * rb;
  const double d = __builtin_amdgcn_fract(qt);
  *floc = __builtin_floor(qt);
  return (__builtin_fabs(d - 0.5) < 0.5 - 0x1p-20) && (__builtin_fabs(qt) < 0x1p31);
}
__device__ __forceinline__ bool floor_quotient_fast(float a0, float rb, float* floc) {
  const float qt = a0 * rb;
  const float d = __builtin_amdgcn_fractf(qt);
  *floc = __builtin_floorf(qt);
  return __builtin_fabsf(d - 0.5f) + __builtin_fmaf(__builtin_fabsf(qt), 0x1p-21f, 0x1p-21f) < 0.5f;
}
__device__ __forceinline__ bool divide_fast(double a, double b, double rb, double* q) {
  const double q0 = a * rb;
  const double r0 = __builtin_fma(-b, q0, a);
  const double q1 = __builtin_fma(r0, rb, q0);
  const double r1 = __builtin_fma(-b, q1, a);
  *q = __builtin_fma(r1, rb, q1);
  return exponent_within_256(a);
}
__device__ __forceinline__ bool divide_fast(float a, float b, float rb, float* q) {
  const float q0 = a * rb;
  const float r0 = __builtin_fmaf(-b, q0, a);
  const float q1 = __builtin_fmaf(r0, rb, q0);
  const float r1 = __builtin_fmaf(-b, q1, a);
  *q = __builtin_fmaf(r1, rb, q1);
  return exponent_within_24(a);
}

// the steps the host may hand to step_cell_fast (per element type)
template <typename T> struct StepCellRange;
template <> struct StepCellRange<double> { static constexpr double lo = 0x1p-128, hi = 0x1p128; };
template <> struct StepCellRange<float> { static constexpr double lo = 0x1p-16, hi = 0x1p16; };

// The quotient sequence of divide_fast alone, and the two tests that admit its operands where the divisor is not a
// host constant (the spacing ratios of a rectilinear multicubic cell, cubic_rect_node_fast below):
//   divisor b: positive, 2^-128 <= b < 2^128 (f32: 2^-16 <= b < 2^16), rb = RN(1 / b) by a division in T;
//   numerator a: 2^-256 <= |a| < 2^256 (f32: 2^-24 .. 2^24) or +0.  (+0 with b > 0: q0 = +0, both remainders
//   +0 - 0 = +0, both corrections +0 + +0 = +0 = +0 / b.  -0 is not admitted: the corrections would add +0 to -0.)
__device__ __forceinline__ double quotient_fast(double a, double b, double rb) {
  const double q0 = a * rb;
  const double r0 = __builtin_fma(-b, q0, a);
  const double q1 = __builtin_fma(r0, rb, q0);
  const double r1 = __builtin_fma(-b, q1, a);
  return __builtin_fma(r1, rb, q1);
}
__device__ __forceinline__ float quotient_fast(float a, float b, float rb) {
  const float q0 = a * rb;
  const float r0 = __builtin_fmaf(-b, q0, a);
  const float q1 = __builtin_fmaf(r0, rb, q0);
  const float r1 = __builtin_fmaf(-b, q1, a);
  return __builtin_fmaf(r1, rb, q1);
}
__device__ __forceinline__ bool fast_numerator(double a) { return exponent_within_256(a) || __builtin_amdgcn_class(a, 0x40); }
__device__ __forceinline__ bool fast_numerator(float a) { return exponent_within_24(a) || __builtin_amdgcn_classf(a, 0x40); }
__device__ __forceinline__ bool fast_divisor(double b) {  // sign bit in the field: negative divisors fail
  return (((unsigned)__double2hiint(b) >> 20) - (1023u - 128u)) < 256u;
}
__device__ __forceinline__ bool fast_divisor(float b) { return ((__float_as_uint(b) >> 23) - (127u - 16u)) < 32u; }

// core::slice::partition_point(|g| *g < x) restated with Rust std's probe sequence
// (size-halving binary search); trip count depends only on n, so a wave never diverges.
template <typename T, typename GridPtr>
__device__ __forceinline__ int partition_point_lt(GridPtr g, int n, T x) {
  int size = n;
  int base = 0;
  while (size > 1) {
    int half = size >> 1;
    int mid = base + half;
    base = (g[mid] < x) ? mid : base;
    size -= half;
  }
  return base + ((g[base] < x) ? 1 : 0);
}

// The same probe sequence on an axis of at most 64 coordinates held one per lane (`greg` =
// g[lane]); probes are cross-lane reads (ds_bpermute): no LDS memory, hence none of the bank
// conflicts random 8-byte gathers on a tiny axis image suffer (they cost the rectilinear kernel
// ~0.3 ms per 1e8 points).  Must be called with the whole wave active.
template <typename T>
__device__ __forceinline__ int partition_point_lt_lanes(T greg, int n, T x) {
  int size = n;
  int base = 0;
  while (size > 1) {  // trip count depends on n only
    const int half = size >> 1;
    const int mid = base + half;
    base = (__shfl(greg, mid) < x) ? mid : base;
    size -= half;
  }
  return base + ((__shfl(greg, base) < x) ? 1 : 0);
}

// Rectilinear axis as the kernels see it: coordinates plus (for strictly increasing, finite
// axes) a bucket table that brackets the bisection.  tab[b] = number of coordinates whose own
// bucket index is < b, so for a query x in bucket b the answer lies in [tab[b], tab[b+1]].
// bucket_of() is monotone non-decreasing in x (subtract, multiply by a positive constant,
// truncate, clamp), and the same function is applied to the coordinates when the table is
// built (k_build_buckets), therefore every coordinate in an earlier bucket is < x and every one
// in a later bucket is >= x, for any rounding: the count of coordinates < x is exactly what
// core::slice::partition_point returns on a sorted slice.
template <typename T>
struct Axis {
  const T* g;
  const unsigned* tab;
  int n;
  int M;  // number of buckets, 0 = no table (axis not proven sorted): std probe sequence
  T g0;
  T scale;
  const unsigned char* rec = nullptr;  // per-bucket records (below), or null
  bool compact = false;                // `rec` holds AxisRecordC (g[k] and k only; the brackets are read from `g`)
};

// Per-bucket record of an axis whose buckets hold at most one coordinate each: with k = tab[b],
// {gm1, g0, gp1} = {g[k-1], g[k], g[k+1]} (entries outside the axis are never selected).
template <typename T> struct AxisRecord;
template <> struct __attribute__((aligned(16))) AxisRecord<double> { double gm1, g0, gp1; unsigned k, pad; };
template <> struct __attribute__((aligned(16))) AxisRecord<float> { float gm1, g0, gp1; unsigned k; };

// Compact form (round 4), for axes whose full records do not fit the kernel's LDS budget: {g[k], k}
// — 16 bytes in f64, 8 in f32 — next to the coordinates themselves (8n / 4n bytes): 40n bytes per
// f64 axis instead of 64n, a second (dependent) LDS access for the two brackets.
template <typename T> struct AxisRecordC;
template <> struct __attribute__((aligned(16))) AxisRecordC<double> { double g0; unsigned k, pad; };
template <> struct __attribute__((aligned(8))) AxisRecordC<float> { float g0; unsigned k; };

constexpr int kLaneBuckets = 255;  // buckets of a lane table (256 byte entries = 64 lanes x 4)

template <typename T>
__device__ __forceinline__ int bucket_of(T x, T g0, T scale, int M) {
  const T u = (x - g0) * scale;
  return u >= (T)(M - 1) ? (M - 1) : (u > (T)0 ? (int)u : 0);
}

template <typename T>
__device__ __forceinline__ int axis_partition_point(const Axis<T>& ax, T x) {
  if (ax.M > 0) {
    if (!(x == x)) return 0;  // NaN: `g < NaN` is false for every g
    const int b = bucket_of<T>(x, ax.g0, ax.scale, ax.M);
    int idx = (int)ax.tab[b];
    const int hi = (int)ax.tab[b + 1];
    while (idx < hi && ax.g[idx] < x) ++idx;
    return idx;
  }
  return partition_point_lt<T>(ax.g, ax.n, x);
}

// Cell of a multilinear / nearest query on a rectilinear axis: loc = clamp(partition_point - 1,
// 0, n-2) and the two bracketing coordinates g[loc], g[loc+1] (multilinear/rectilinear.rs:353-370
// and :310-311).  (A speculative variant that fetched the four coordinates around the bucket start
// with independent reads and selected x0/x1 from registers was measured SLOWER — cfg3 2.01 vs
// 1.80 ms: the search is bound by the number of bank-conflicting LDS gathers on the tiny axis
// image, not by their dependency chain — so the short scan below stays.)
template <typename T>
__device__ __forceinline__ int axis_cell(const Axis<T>& ax, T x, T* x0, T* x1) {
  const int n = ax.n;
  if (ax.rec) {
    // One access.  k = tab[b] coordinates lie in earlier buckets, so every one of them is < x and
    // every coordinate of a later bucket is >= x (bucket_of is monotone and is the function the
    // table was built with); bucket b itself holds at most g[k].  Hence partition_point(g < x) =
    // k + (g[k] < x), and cell l = clamp(that - 1, 0, n - 2) is k (brackets g[k], g[k+1]) or k - 1
    // (brackets g[k-1], g[k]).  NaN: bucket 0, g[0] < NaN is false, l = 0 like the reference.
    const int b = bucket_of<T>(x, ax.g0, ax.scale, ax.M);
    if (ax.compact) {
      const AxisRecordC<T> r = reinterpret_cast<const AxisRecordC<T>*>(ax.rec)[b];
      const int k = (int)r.k;
      const bool hi = (r.g0 < x && k <= n - 2) || k == 0;
      const int l = hi ? k : k - 1;
      *x0 = ax.g[l];
      *x1 = ax.g[l + 1];
      return l;
    }
    const AxisRecord<T> r = reinterpret_cast<const AxisRecord<T>*>(ax.rec)[b];
    const int k = (int)r.k;
    const bool up = r.g0 < x;
    const bool hi = (up && k <= n - 2) || k == 0;
    *x0 = hi ? r.g0 : r.gm1;
    *x1 = hi ? r.gp1 : r.g0;
    return hi ? k : k - 1;
  }
  int l = axis_partition_point<T>(ax, x) - 1;
  l = l > 0 ? l : 0;
  l = l < n - 2 ? l : n - 2;
  *x0 = ax.g[l];
  *x1 = ax.g[l + 1];
  return l;
}

// ---------------------------------------------------------------------------
// Linear tree: reduce dims 0..D-1 (dim 0 innermost) on a W-wide leaf of the last dim.
// Same dependency tree as src/multilinear/regular.rs:347-393 / regular_recursive.rs:348-389.
template <typename T, typename IdxT, int D, bool FMA>
struct LinearTree {
  __device__ __forceinline__ static Leaf<T, 2> run(const T* __restrict__ vals, IdxT base, const IdxT* stride, const T* t) {
    Leaf<T, 2> a = LinearTree<T, IdxT, D - 1, FMA>::run(vals, base, stride, t);
    Leaf<T, 2> b = LinearTree<T, IdxT, D - 1, FMA>::run(vals, base + stride[D - 1], stride, t);
    Leaf<T, 2> r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      T y0 = a.v[i];
      T dy = b.v[i] - y0;
      r.v[i] = mul_add<FMA>(t[D - 1], dy, y0);  // regular.rs:378-385
    }
    return r;
  }
};
template <typename T, typename IdxT, bool FMA>
struct LinearTree<T, IdxT, 0, FMA> {
  __device__ __forceinline__ static Leaf<T, 2> run(const T* __restrict__ vals, IdxT base, const IdxT*, const T*) {
    return load_leaf<T, 2>(vals + base);
  }
};

// ---------------------------------------------------------------------------
// Cubic node, regular grid: src/multicubic/regular.rs:474-623 with the saturation match
// turned into selects (every case computes exactly the reference's expression for it):
//   None : t,    y0=v1, dy=v2-v1, k0=(v2-v0)/2,  k1=(v3-v1)/2
//   Low  : -t,   y0=v1, dy=v0-v1, k0=-(v2-v0)/2, k1=2dy-k0   (Inside/OutsideLow)
//   High : t-1,  y0=v2, dy=v3-v2, k0=(v3-v1)/2,  k1=2dy-k0   (Inside/OutsideHigh)
// `two.mul_add(dy, -k0)` == `two*dy - k0` bit for bit (2*dy is exact).
// Hermite: src/multicubic/mod.rs:72-91.
// The spline in two steps: its coefficients (everything that does not depend on t — the same for
// every point that shares the node's four values and its saturation arm) and Horner's evaluation.
// hermite() is the two in sequence, so a kernel that keeps coefficients (cubic_column.h) computes
// the very operations of the reference, in its order.
template <typename T>
struct HermiteCoef { T y0, c1, c2, c3; };

template <typename T>
__device__ __forceinline__ HermiteCoef<T> hermite_coef(T y0, T dy, T k0, T k1) {
  T a = k0 - dy;
  T b = -k1 + dy;
  HermiteCoef<T> c;
  c.y0 = y0;
  c.c1 = dy + a;
  c.c2 = b - (a + a);
  c.c3 = a - b;
  return c;
}

template <bool FMA, typename T>
__device__ __forceinline__ T hermite_eval(T t, T y0, T c1, T c2, T c3) {
  if constexpr (FMA) {
    return dev_fma<T>(dev_fma<T>(dev_fma<T>(c3, t, c2), t, c1), t, y0);
  } else {
    T i0 = t * c3;
    T i1 = c2 + i0;
    T i2 = t * i1;
    T i3 = c1 + i2;
    T i4 = t * i3;
    return y0 + i4;
  }
}

template <bool FMA, typename T>
__device__ __forceinline__ T hermite(T t, T y0, T dy, T k0, T k1) {
  const HermiteCoef<T> c = hermite_coef<T>(y0, dy, k0, k1);
  return hermite_eval<FMA, T>(t, c.y0, c.c1, c.c2, c.c3);
}

template <typename T>
struct CubicDimRegular {
  T tt;        // t (None), -t (Low), t-1 (High)
  int sat;     // Sat
  int linear;  // OutsideLow/OutsideHigh with linearize_extrapolation
  // the reference's recursive arm writes OutsideLow's k1 as `two * dy - k0` also under its `fma` feature
  // (regular_recursive.rs:536; every other saturated class, and every class of the flattened arm, is two.mul_add(dy, -k0):
  // regular.rs:525-528, :546-549, :580-583, :602-605, regular_recursive.rs:516-519, :567-570, :589-592).  2 dy is exact, so the two
  // forms differ only where 2 dy overflows.  Read by cubic_regular_node<FMA, T, true> alone (the runtime-N kernels).
  int k1_plain = 0;
};

// The same node when the saturation class is known to be None (interior cell, no linearized
// extrapolation): the reference's `Saturation::None` arm (multicubic/regular.rs:495-505) without
// the selects.  Used by the tiled kernels when a whole wave is interior along a dimension — the
// common case once the points are sorted by cell (binned evaluation): 80 of the 85 nodes of a 4-D
// point belong to dims 0 and 1.  Every operation is the one the select form evaluates for
// sat == None, so the bits are the same.
template <bool FMA, typename T>
__device__ __forceinline__ T cubic_regular_node_interior(T v0, T v1, T v2, T v3, T t) {
  const T two = (T)2;
  const T dy = v2 - v1;
  const T k0 = (v2 - v0) / two;
  const T k1 = (v3 - v1) / two;
  return hermite<FMA>(t, v1, dy, k0, k1);
}

// Two interior nodes of one dimension at once in PACKED f32 (v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32, two IEEE single-precision operations per lane and instruction on CDNA): lane-wise
// exactly the operations of cubic_regular_node_interior + hermite in the same order, each rounded
// like its scalar form (x / 2 == x * 0.5 bit for bit, subnormals included; -k1 + dy is one
// subtraction either way), so the two results are the scalar results.  Used where four nodes of a
// tile share their t (dim 0 of a 4 x 4 footprint): 28 instead of 56 instructions.
typedef float float_pair __attribute__((ext_vector_type(2)));
template <bool FMA>
__device__ __forceinline__ float_pair cubic_regular_node_interior2(float_pair v0, float_pair v1, float_pair v2, float_pair v3, float t) {
  const float_pair tt = {t, t};
  const float_pair half = {0.5f, 0.5f};
  const float_pair dy = v2 - v1;
  const float_pair k0 = (v2 - v0) * half;
  const float_pair k1 = (v3 - v1) * half;
  const float_pair a = k0 - dy;
  const float_pair b = dy - k1;
  const float_pair c1 = dy + a;
  const float_pair c2 = b - (a + a);
  const float_pair c3 = a - b;
  if constexpr (FMA) {
    return __builtin_elementwise_fma(__builtin_elementwise_fma(__builtin_elementwise_fma(c3, tt, c2), tt, c1), tt, v1);
  } else {
    const float_pair i0 = tt * c3;
    const float_pair i1 = c2 + i0;
    const float_pair i2 = tt * i1;
    const float_pair i3 = c1 + i2;
    const float_pair i4 = tt * i3;
    return v1 + i4;
  }
}

// dim 0 of a 4 x 4 tile (v[e], e = ei * 4 + ej) for all four ej, every lane interior: w[ej].
template <bool FMA, typename T>
__device__ __forceinline__ void cubic_tile_dim0_interior(const T (&v)[16], T t, T (&w)[4]) {
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const float_pair a0 = {v[2 * p], v[2 * p + 1]}, a1 = {v[4 + 2 * p], v[4 + 2 * p + 1]};
      const float_pair a2 = {v[8 + 2 * p], v[8 + 2 * p + 1]}, a3 = {v[12 + 2 * p], v[12 + 2 * p + 1]};
      const float_pair r = cubic_regular_node_interior2<FMA>(a0, a1, a2, a3, t);
      w[2 * p] = r.x;
      w[2 * p + 1] = r.y;
    }
  } else {
#pragma unroll
    for (int ej = 0; ej < 4; ++ej) w[ej] = cubic_regular_node_interior<FMA, T>(v[ej], v[4 + ej], v[8 + ej], v[12 + ej], t);
  }
}

template <bool FMA, typename T, bool ARMS = false>
__device__ __forceinline__ T cubic_regular_node(T v0, T v1, T v2, T v3, const CubicDimRegular<T>& d) {
  const T two = (T)2, one = (T)1;
  const bool low = d.sat == kSatLow;
  const bool high = d.sat == kSatHigh;
  T y0 = high ? v2 : v1;
  T ya = low ? v0 : (high ? v3 : v2);
  T dy = ya - y0;
  T cd = high ? (v3 - v1) : (v2 - v0);
  T k0 = cd / two;
  k0 = low ? -k0 : k0;
  T k1n = (v3 - v1) / two;
  T k1e = mul_add<FMA>(two, dy, -k0);  // regular.rs:525-528: fused under the `fma` feature (the same bits unless 2 dy overflows)
  if constexpr (ARMS && FMA) {
    if (d.k1_plain) k1e = two * dy - k0;
  }
  T k1 = (low || high) ? k1e : k1n;
  if (d.linear) {
    T y1 = ya;  // vals[0] (low) / vals[3] (high)
    return mul_add<FMA>(k1, d.tt - one, y1);  // regular.rs:553-561, :609-617
  }
  return hermite<FMA>(d.tt, y0, dy, k0, k1);
}

// Cubic node, rectilinear grid: src/multicubic/rectilinear.rs:413-545, with the
// non-uniform central difference of src/multicubic/mod.rs:103-117.  Everything that
// depends only on the dimension (spacings, spacing ratios, weights a and c of the
// non-uniform central difference, t) is computed once per point and dimension; the
// per-node part is the value-dependent remainder, evaluated in the reference's order.
template <typename T>
struct CubicDimRect {
  int sat;
  int linear;      // OutsideLow/OutsideHigh with linearize_extrapolation
  int fma_linear;  // recursive arm fuses the linearized branch (rectilinear_recursive.rs:467,527)
  T t;
  // cd(y0,y1,y2,hA,hB) = a*b + c*d, a = hA/(hA+hB), b = (y2-y1)/hB, c = hB/(hB+hA), d = (y1-y0)/hA
  //   None: k0 = cd(v0,v1,v2, h01/h12, 1),  k1 = cd(v1,v2,v3, 1, h23/h12)
  //   Low : k0 = -cd(v0,v1,v2, 1, h12/h01); High: k0 = cd(v1,v2,v3, h12/h23, 1)
  // One of (hA, hB) is always the literal 1 and x/1 == x exactly, so each central difference
  // carries a single real division; r is the other (non-unit) spacing ratio.
  T r0, a0, c0;  // k0's difference: hB == 1 for None/High, hA == 1 for Low
  T r1, a1, c1;  // k1's difference (None only): hA == 1
  // cubic_rect_node_fast only (RECIP setup): RN(1 / r0), RN(1 / r1); fast: both ratios pass fast_divisor
  T rr0, rr1;
  bool fast;
};

// Control flow of the class arms (here and in the two node functions below): ONE two-way branch — interior or saturated —
// with the saturated classes Low and High told apart by selects on the operands of a single body.  Written as three
// arms (`if None ... else if Low ... else High`) the chain becomes a switch on `sat` whose default arm is reachable from
// both halves of the lowered decision tree, and ROCm 7.2's StructurizeCFG (amdclang 22: the pass hoists zero-cost
// incoming values of such an arm's phis and then routes one of its two entries past them) gave the High lanes that
// enter from the `sat >= 1` side the LOW arm's y0 and an undefined y1 in one build of the 4-D column kernel
// (profiles/NOTES.md section H; found because every High-class point of dim 1 was wrong, round 5).  Every operation a
// lane's result depends on is the reference's for that lane's class, on the same operands, in the same order.
template <typename T, bool RECIP = false, typename GridPtr>
__device__ __forceinline__ void cubic_rect_dim_setup(GridPtr g, int loc, T x, CubicDimRect<T>& d) {
  const T one = (T)1;
  T g0 = g[loc], g1 = g[loc + 1], g2 = g[loc + 2], g3 = g[loc + 3];
  d.r1 = one; d.a1 = one; d.c1 = one;
  d.rr0 = one; d.rr1 = one; d.fast = false;
  const T h12 = g2 - g1;
  if (d.sat == kSatNone) {
    T h01 = g1 - g0, h23 = g3 - g2;
    d.r0 = h01 / h12;  // (hA, hB) = (r0, 1)
    d.a0 = d.r0 / (d.r0 + one);
    d.c0 = one / (one + d.r0);
    d.r1 = h23 / h12;  // (hA, hB) = (1, r1)
    d.a1 = one / (one + d.r1);
    d.c1 = d.r1 / (d.r1 + one);
    d.t = (x - g1) / h12;
  } else {
    // Low : r0 = h12 / h01, (hA, hB) = (1, r0): a0 = 1 / (1 + r0), c0 = r0 / (r0 + 1), t = -(x - g1) / h01
    // High: r0 = h12 / h23, (hA, hB) = (r0, 1): a0 = r0 / (r0 + 1), c0 = 1 / (1 + r0), t = (x - g2) / h23
    const bool low = d.sat == kSatLow;
    const T ho = low ? g1 - g0 : g3 - g2;
    d.r0 = h12 / ho;
    const T wr = d.r0 / (d.r0 + one), w1 = one / (one + d.r0);
    d.a0 = low ? w1 : wr;
    d.c0 = low ? wr : w1;
    const T num = x - (low ? g1 : g2);
    d.t = (low ? -num : num) / ho;
  }
  if constexpr (RECIP) {
    d.rr0 = one / d.r0;
    if (d.sat == kSatNone) d.rr1 = one / d.r1;
    d.fast = fast_divisor(d.r0) && fast_divisor(d.r1);
  }
}

// Central difference with hB == 1: b = (y2-y1)/1, d = (y1-y0)/r.
template <bool FMA, typename T>
__device__ __forceinline__ T cd_unit_b(T y0, T y1, T y2, T r, T a, T c) {
  T b = y2 - y1;
  T dd = (y1 - y0) / r;
  if constexpr (FMA) {
    return dev_fma<T>(a, b, c * dd);
  } else {
    T ab = a * b;
    T cdd = c * dd;
    return ab + cdd;
  }
}
// Central difference with hA == 1: b = (y2-y1)/r, d = (y1-y0)/1.
template <bool FMA, typename T>
__device__ __forceinline__ T cd_unit_a(T y0, T y1, T y2, T r, T a, T c) {
  T b = (y2 - y1) / r;
  T dd = y1 - y0;
  if constexpr (FMA) {
    return dev_fma<T>(a, b, c * dd);
  } else {
    T ab = a * b;
    T cdd = c * dd;
    return ab + cdd;
  }
}

// The saturated classes' first slope and Hermite operands (the select form of the note above cubic_rect_dim_setup):
//   Low : y0 = v1, y1 = v0, dy = v0 - v1, k0 = -cd_unit_a(v0, v1, v2): b = (v2 - v1) / r0, dd = v1 - v0
//   High: y0 = v2, y1 = v3, dy = v3 - v2, k0 =  cd_unit_b(v1, v2, v3): b = v3 - v2, dd = (v2 - v1) / r0
// — the one real division has the same operands in both.
template <bool FMA, typename T>
__device__ __forceinline__ void cubic_rect_saturated(T v0, T v1, T v2, T v3, const CubicDimRect<T>& d, T& y0, T& y1, T& dy, T& k0) {
  const bool low = d.sat == kSatLow;
  const T q = (v2 - v1) / d.r0;
  const T up = v3 - v2, dn = v1 - v0;
  const T b = low ? q : up;
  const T dd = low ? dn : q;
  T k;
  if constexpr (FMA) {
    k = dev_fma<T>(d.a0, b, d.c0 * dd);
  } else {
    T ab = d.a0 * b;
    T cdd = d.c0 * dd;
    k = ab + cdd;
  }
  k0 = low ? -k : k;
  y0 = low ? v1 : v2;
  y1 = low ? v0 : v3;
  dy = low ? v0 - v1 : up;
}

template <bool FMA, typename T>
__device__ __forceinline__ T cubic_rect_node(T v0, T v1, T v2, T v3, const CubicDimRect<T>& d) {
  const T two = (T)2, one = (T)1;
  if (d.sat == kSatNone) {
    T y0 = v1;
    T dy = v2 - v1;
    T k0 = cd_unit_b<FMA>(v0, v1, v2, d.r0, d.a0, d.c0);
    T k1 = cd_unit_a<FMA>(v1, v2, v3, d.r1, d.a1, d.c1);
    return hermite<FMA>(d.t, y0, dy, k0, k1);
  }
  T y0, y1, dy, k0;
  cubic_rect_saturated<FMA, T>(v0, v1, v2, v3, d, y0, y1, dy, k0);
  T k1 = two * dy - k0;
  if (d.linear) {
    if (FMA && d.fma_linear) return dev_fma<T>(k1, d.t - one, y1);
    T p = k1 * (d.t - one);
    return y1 + p;  // rectilinear.rs:500,:539 — never fused in the flattened arm
  }
  return hermite<FMA>(d.t, y0, dy, k0, k1);
}

// The same node for a wave whose lanes differ in class, without a branch per class and with the two spacing-ratio
// divisions as quotient_fast: every operation a lane's result depends on is the operation cubic_rect_node does for that
// lane's class, on the same operands, in the same order (the operands are chosen by selects; what a lane computes for the
// other classes is discarded) — the reference's bits where `ok` stays true (d.fast and every numerator in use admitted);
// where it does not the caller evaluates the point again with cubic_rect_node.
//   first difference  None: cd_unit_b(v0, v1, v2), Low: -cd_unit_a(v0, v1, v2), High: cd_unit_b(v1, v2, v3)   (r0, a0, c0)
//   second difference None: cd_unit_a(v1, v2, v3) (r1, a1, c1); Low / High: 2 dy - k0
template <bool FMA, typename T>
__device__ __forceinline__ T cubic_rect_node_fast(T v0, T v1, T v2, T v3, const CubicDimRect<T>& d, bool& ok) {
  const T two = (T)2, one = (T)1;
  const bool low = d.sat == kSatLow, high = d.sat == kSatHigh;
  const bool none = !(low || high);
  const T f01 = v1 - v0, f12 = v2 - v1, f23 = v3 - v2;
  const T e01 = high ? f12 : f01;  // y1 - y0 and y2 - y1 of the first difference's three values
  const T e12 = high ? f23 : f12;
  const T num0 = low ? e12 : e01;  // the side that is divided by the ratio: unit_a (Low) b = e12 / r, unit_b dd = e01 / r
  ok = ok && fast_numerator(num0);
  const T q0 = quotient_fast(num0, d.r0, d.rr0);
  const T b0 = low ? q0 : e12;
  const T d0 = low ? e01 : q0;
  T k0;
  if constexpr (FMA) k0 = dev_fma<T>(d.a0, b0, d.c0 * d0);
  else { const T ab = d.a0 * b0; const T cdd = d.c0 * d0; k0 = ab + cdd; }
  k0 = low ? -k0 : k0;
  ok = ok && (!none || fast_numerator(f23));
  const T q1 = quotient_fast(f23, d.r1, d.rr1);  // (r1 = rr1 = 1 on lanes that are not None: finite work, discarded)
  T k1n;
  if constexpr (FMA) k1n = dev_fma<T>(d.a1, q1, d.c1 * f12);
  else { const T ab = d.a1 * q1; const T cdd = d.c1 * f12; k1n = ab + cdd; }
  const T y0 = high ? v2 : v1;
  const T dy = low ? v0 - v1 : e12;
  const T k1 = none ? k1n : two * dy - k0;
  if (d.linear) {  // (outside the grid with linearize_extrapolation: Low or High)
    const T y1 = low ? v0 : v3;
    if (FMA && d.fma_linear) return dev_fma<T>(k1, d.t - one, y1);
    T p = k1 * (d.t - one);
    return y1 + p;
  }
  return hermite<FMA>(d.t, y0, dy, k0, k1);
}

// The coefficients of the same node (its Hermite arms only: a point that extrapolates linearly
// does not use them) — cubic_rect_node is hermite_eval(d.t, ...) of these.
template <bool FMA, typename T>
__device__ __forceinline__ HermiteCoef<T> cubic_rect_node_coef(T v0, T v1, T v2, T v3, const CubicDimRect<T>& d) {
  const T two = (T)2;
  if (d.sat == kSatNone) {
    T dy = v2 - v1;
    T k0 = cd_unit_b<FMA>(v0, v1, v2, d.r0, d.a0, d.c0);
    T k1 = cd_unit_a<FMA>(v1, v2, v3, d.r1, d.a1, d.c1);
    return hermite_coef<T>(v1, dy, k0, k1);
  }
  T y0, y1, dy, k0;
  cubic_rect_saturated<FMA, T>(v0, v1, v2, v3, d, y0, y1, dy, k0);
  T k1 = two * dy - k0;
  return hermite_coef<T>(y0, dy, k0, k1);
}

// Cubic tree on a 4-wide leaf of the last dim; NodeFn(v0..v3, dim) -> T.
template <typename T, typename IdxT, int D, typename DimT, typename NodeFn>
struct CubicTree {
  __device__ __forceinline__ static Leaf<T, 4> run(const T* __restrict__ vals, IdxT base, const IdxT* stride, const DimT* dim, NodeFn fn) {
    Leaf<T, 4> r0 = CubicTree<T, IdxT, D - 1, DimT, NodeFn>::run(vals, base, stride, dim, fn);
    Leaf<T, 4> r1 = CubicTree<T, IdxT, D - 1, DimT, NodeFn>::run(vals, base + stride[D - 1], stride, dim, fn);
    Leaf<T, 4> r2 = CubicTree<T, IdxT, D - 1, DimT, NodeFn>::run(vals, base + 2 * stride[D - 1], stride, dim, fn);
    Leaf<T, 4> r3 = CubicTree<T, IdxT, D - 1, DimT, NodeFn>::run(vals, base + 3 * stride[D - 1], stride, dim, fn);
    Leaf<T, 4> r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = fn(r0.v[i], r1.v[i], r2.v[i], r3.v[i], dim[D - 1]);
    return r;
  }
};
template <typename T, typename IdxT, typename DimT, typename NodeFn>
struct CubicTree<T, IdxT, 0, DimT, NodeFn> {
  __device__ __forceinline__ static Leaf<T, 4> run(const T* __restrict__ vals, IdxT base, const IdxT*, const DimT*, NodeFn) {
    return load_leaf<T, 4>(vals + base);
  }
};

}  // namespace interpn
