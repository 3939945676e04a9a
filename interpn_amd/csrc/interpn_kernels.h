// Kernels and launchers (templates).  Instantiated per method/grid kind in the *.hip files
// next to this header so that the translation units build in parallel.
//
// Kernel families in this header (C-ordered grid):
//   k_linear_regular<T,N,FMA,U>      N = 1..6   multilinear::regular      (flattened arm)
//   k_linear_rectilinear<T,N,FMA,U>  N = 1..6   multilinear::rectilinear  (flattened arm)
//   k_cubic_regular<T,N,FMA>         N = 1..4   multicubic::regular       (flattened arm)
//   k_cubic_rectilinear<T,N,FMA>     N = 1..4   multicubic::rectilinear   (flattened arm)
//   k_generic_n<T,METHOD,KIND,FMA,N,VEC>  the recursive arms (linear N = 7,8; cubic N = 5..8)
//   k_generic<T,METHOD,KIND,FMA>     runtime N <= 8, 64-bit indexing: grids of 2^32 elements and
//                                    more, and the reference form the static-N kernel is tested against.
// The kernels that carry the benchmarked shapes live next to it and read a re-laid copy of the
// grid: k_linear_brick.hip (multilinear N = 3..6), k_linear2_brick.hip (N = 2), cubic_brick.h
// (multicubic N = 2..4), k_nearest.hip.  The C-order kernels here remain the path for N = 1,
// for grids whose re-laid copy does not fit, and when INTERPN_HIP_BRICKS=off.
//
// Launch shape: 256-thread workgroups (4 waves, one per SIMD).  The C-order kernels run a grid of
// a few workgroups per CU that strides over the observation points (U points per lane and
// iteration keep several independent gathers in flight per wave); the brick kernels cover the
// batch in one pass of small workgroups (interpn_host.h::brick_iters).
#pragma once

#include "interpn_device.h"
#include "interpn_host.h"

namespace interpn {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------
template <typename T, int N>
struct RegularArgs {
  const T* vals;
  const T* obs[N];
  T* out;
  unsigned long long* first_bad;
  size_t npts;
  T start[N];
  T step[N];
  int n[N];            // points per axis
  unsigned stride[N];  // element stride of each dim (C order)
  int linearize;
};

// Axis image (GridDesc::axis_image) as kernel arguments.
template <typename T, int N>
struct AxisArgs {
  const unsigned char* image;  // device
  unsigned image_bytes;
  unsigned g_off[N];
  unsigned tab_off[N];
  int n[N];
  int M[N];
  T g0[N];
  T scale[N];
  int use_lds;
  unsigned ltab_off[N];  // lane tables (axes <= 64 coordinates), 0 = none
  T lscale[N];
  // Per-bucket search records (GridDesc::axis_rec_*): when use_rec is set, `image` / `image_bytes`
  // are the records region alone and rec_off[d] is axis d's offset inside it; g_off / tab_off are
  // then not to be dereferenced.  use_rec == 2: the compact form — `image` is the compact region
  // (per axis its coordinates, then its AxisRecordC entries), g_off[d] / rec_off[d] offsets inside it.
  int use_rec;
  unsigned rec_off[N];
};

template <typename T, int N>
__device__ __forceinline__ Axis<T> make_axis(const AxisArgs<T, N>& a, const unsigned char* base, int d) {
  Axis<T> ax;
  ax.g = reinterpret_cast<const T*>(base + a.g_off[d]);
  ax.tab = reinterpret_cast<const unsigned*>(base + a.tab_off[d]);
  ax.n = a.n[d];
  ax.M = a.M[d];
  ax.g0 = a.g0[d];
  ax.scale = a.scale[d];
  ax.rec = a.use_rec ? base + a.rec_off[d] : nullptr;
  ax.compact = a.use_rec == 2;
  return ax;
}

// Copy the axis image into LDS (whole workgroup), 4 bytes per thread and step.
template <typename T, int N>
__device__ __forceinline__ void stage_axes(const AxisArgs<T, N>& a, unsigned char* lds) {
  const unsigned words = a.image_bytes >> 2;
  const unsigned* src = reinterpret_cast<const unsigned*>(a.image);
  unsigned* dst = reinterpret_cast<unsigned*>(lds);
  for (unsigned k = threadIdx.x; k < words; k += kBlock) dst[k] = src[k];
  __syncthreads();
}

template <typename T, int N>
struct RectArgs {
  const T* vals;
  const T* obs[N];
  T* out;
  size_t npts;
  AxisArgs<T, N> ax;
  unsigned stride[N];
  int linearize;
};

// ===========================================================================
// multilinear::regular — src/multilinear/regular.rs:268-283 (loop), :296-404 (interp_one)
template <typename T, int N, bool FMA, int U>
__global__ void __launch_bounds__(kBlock) k_linear_regular(const RegularArgs<T, N> a) {
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  for (size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x; i0 < a.npts; i0 += nthreads * U) {
    T x[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * nthreads;
#pragma unroll
      for (int d = 0; d < N; ++d) x[u][d] = (i < a.npts) ? stream_load(a.obs[d] + i) : a.start[d];
    }
    T t[U][N];
    unsigned base[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * nthreads;
      bool ok = true;
      base[u] = 0;
#pragma unroll
      for (int d = 0; d < N; ++d) {
        T floc;
        ok &= regular_floc<T>(x[u][d], a.start[d], a.step[d], &floc);  // regular.rs:415-418
        const int loc = clamp_loc<T>(floc, a.n[d] - 2);                // regular.rs:420-422
        // regular.rs:334-339: fused in the flattened arm when the `fma` feature is on
        const T index_zero_loc = mul_add<FMA>(a.step[d], (T)loc, a.start[d]);
        t[u][d] = (x[u][d] - index_zero_loc) / a.step[d];
        base[u] += (unsigned)loc * a.stride[d];
      }
      if (!ok && i < a.npts) atomicMin(a.first_bad, (unsigned long long)i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * nthreads;
      Leaf<T, 2> r = LinearTree<T, unsigned, N - 1, FMA>::run(a.vals, base[u], a.stride, t[u]);
      const T y0 = r.v[0];
      const T dy = r.v[1] - y0;
      const T res = mul_add<FMA>(t[u][N - 1], dy, y0);  // regular.rs:396-402
      if (i < a.npts) stream_store(a.out + i, res);
    }
  }
}

// ===========================================================================
// multilinear::rectilinear — src/multilinear/rectilinear.rs:210-231, :244-370
template <typename T, int N, bool FMA, int U, bool LDS>
__device__ __forceinline__ void linear_rectilinear_body(const RectArgs<T, N>& a, const unsigned char* lds) {
  const unsigned char* axbase = LDS ? lds : a.ax.image;
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  for (size_t i0 = (size_t)blockIdx.x * kBlock + threadIdx.x; i0 < a.npts; i0 += nthreads * U) {
    T x[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * nthreads;
#pragma unroll
      for (int d = 0; d < N; ++d) x[u][d] = (i < a.npts) ? stream_load(a.obs[d] + i) : (T)0;
    }
    T t[U][N];
    unsigned base[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      base[u] = 0;
#pragma unroll
      for (int d = 0; d < N; ++d) {
        const Axis<T> ax = make_axis<T, N>(a.ax, axbase, d);
        T x0, x1;
        const int loc = axis_cell<T>(ax, x[u][d], &x0, &x1);  // rectilinear.rs:353-370, :310-311
        const T step = x1 - x0;
        t[u][d] = (x[u][d] - x0) / step;  // rectilinear.rs:310-313 (same value at every node of dim d)
        base[u] += (unsigned)loc * a.stride[d];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * nthreads;
      Leaf<T, 2> r = LinearTree<T, unsigned, N - 1, FMA>::run(a.vals, base[u], a.stride, t[u]);
      const T y0 = r.v[0];
      const T dy = r.v[1] - y0;
      const T res = mul_add<FMA>(t[u][N - 1], dy, y0);  // rectilinear.rs:339-344
      if (i < a.npts) stream_store(a.out + i, res);
    }
  }
}

template <typename T, int N, bool FMA, int U>
__global__ void __launch_bounds__(kBlock) k_linear_rectilinear(const RectArgs<T, N> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (a.ax.use_lds) {
    stage_axes<T, N>(a.ax, smem_raw);
    linear_rectilinear_body<T, N, FMA, U, true>(a, smem_raw);
  } else {
    linear_rectilinear_body<T, N, FMA, U, false>(a, nullptr);
  }
}

// ===========================================================================
// multicubic::regular — src/multicubic/regular.rs:297-313, :325-469
template <typename T, bool FMA>
struct CubicRegularNode {
  __device__ __forceinline__ T operator()(T v0, T v1, T v2, T v3, const CubicDimRegular<T>& d) const {
    return cubic_regular_node<FMA, T>(v0, v1, v2, v3, d);
  }
};

template <typename T, int N, bool FMA>
__global__ void __launch_bounds__(kBlock) k_cubic_regular(const RegularArgs<T, N> a) {
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.npts; i += nthreads) {
    CubicDimRegular<T> dim[N];
    unsigned base = 0;
    bool ok = true;
#pragma unroll
    for (int d = 0; d < N; ++d) {
      const T x = stream_load(a.obs[d] + i);
      T floc;
      ok &= regular_floc<T>(x, a.start[d], a.step[d], &floc);  // regular.rs:435-438
      ok &= floc != (T)-9223372036854775808.0;  // `- 1` would overflow isize: the reference panics
      // iloc = isize(floc) - 1.  All comparisons below are on floc (an integer-valued float),
      // which is exact for |floc| < 2^63; beyond 2^53 neighbouring integers coincide in f64 but
      // every threshold involved (−1, 0, n−3) is far below that.
      const T n = (T)a.n[d];
      const int loc = clamp_loc<T>(floc - (T)1, a.n[d] - 4);  // regular.rs:440-442
      int sat;
      bool outside;
      // regular.rs:445-466 with iloc = floc-1: iloc < -1 <=> floc < 0; iloc == -1 <=> floc == 0;
      // iloc > n-3 <=> floc > n-2; iloc == n-3 <=> floc == n-2.
      if (floc < (T)0) { sat = kSatLow; outside = true; }
      else if (floc == (T)0) { sat = kSatLow; outside = false; }
      else if (floc > n - (T)2) { sat = kSatHigh; outside = true; }
      else if (floc == n - (T)2) { sat = kSatHigh; outside = false; }
      else { sat = kSatNone; outside = false; }
      // regular.rs:356-360 — never fused
      const T index_one_loc = mul_add<false>(a.step[d], (T)(loc + 1), a.start[d]);
      const T t = (x - index_one_loc) / a.step[d];
      dim[d].sat = sat;
      dim[d].linear = (outside && a.linearize) ? 1 : 0;
      dim[d].tt = sat == kSatLow ? -t : (sat == kSatHigh ? t - (T)1 : t);
      base += (unsigned)loc * a.stride[d];
    }
    if (!ok) atomicMin(a.first_bad, (unsigned long long)i);
    typedef CubicRegularNode<T, FMA> Node;
    Leaf<T, 4> r = CubicTree<T, unsigned, N - 1, CubicDimRegular<T>, Node>::run(a.vals, base, a.stride, dim, Node());
    stream_store(a.out + i, cubic_regular_node<FMA, T>(r.v[0], r.v[1], r.v[2], r.v[3], dim[N - 1]));  // regular.rs:415-421
  }
}

// ===========================================================================
// multicubic::rectilinear — src/multicubic/rectilinear.rs:237-253, :265-408
template <typename T, bool FMA>
struct CubicRectNode {
  __device__ __forceinline__ T operator()(T v0, T v1, T v2, T v3, const CubicDimRect<T>& d) const {
    return cubic_rect_node<FMA, T>(v0, v1, v2, v3, d);
  }
};

template <typename T, bool RECIP = false>
__device__ __forceinline__ int cubic_rect_locate(const Axis<T>& ax, T x, int linearize, bool fma_linear, CubicDimRect<T>& d) {
  const T* g = ax.g;
  const int n = ax.n;
  const int iloc = axis_partition_point<T>(ax, x) - 2;  // rectilinear.rs:377
  int loc = iloc > 0 ? iloc : 0;
  loc = loc < n - 4 ? loc : n - 4;  // rectilinear.rs:379-381
  bool outside = false;
  if (iloc == -2) { d.sat = kSatLow; outside = true; }  // rectilinear.rs:384-405
  else if (iloc == -1) { d.sat = kSatLow; }
  else if (iloc == n - 2) { d.sat = kSatHigh; outside = true; }
  else if (iloc == n - 3) { d.sat = kSatHigh; }
  else { d.sat = kSatNone; }
  d.linear = (outside && linearize) ? 1 : 0;
  d.fma_linear = fma_linear ? 1 : 0;
  cubic_rect_dim_setup<T, RECIP>(g, loc, x, d);
  return loc;
}

// The same from the axis' per-cell records: no division.  d.fast = false: the caller must locate again with
// cubic_rect_locate (a record whose divisors the short form does not take, or x - gref outside its window / not finite).
template <typename T>
__device__ __forceinline__ int cubic_rect_locate_rec(const Axis<T>& ax, const CubicCellRecord<T>* rec, T x, int linearize, CubicDimRect<T>& d) {
  const int n = ax.n;
  const int iloc = axis_partition_point<T>(ax, x) - 2;  // rectilinear.rs:377
  int loc = iloc > 0 ? iloc : 0;
  loc = loc < n - 4 ? loc : n - 4;  // rectilinear.rs:379-381
  int k = iloc > -1 ? iloc : -1;
  k = (k < n - 3 ? k : n - 3) + 1;  // cell: 0 .. n - 2
  const bool low = k == 0, high = k == n - 2;
  d.sat = low ? kSatLow : (high ? kSatHigh : kSatNone);  // rectilinear.rs:384-405
  d.linear = ((iloc == -2 || iloc == n - 2) && linearize) ? 1 : 0;
  d.fma_linear = 0;
  constexpr int PV = 16 / (int)sizeof(T);
  typedef T TV __attribute__((ext_vector_type(PV)));
  const TV* rv = reinterpret_cast<const TV*>(rec + k);
  T f[12];
#pragma unroll
  for (int q = 0; q < 12 / PV; ++q) {
    const TV v = rv[q];
#pragma unroll
    for (int e = 0; e < PV; ++e) f[q * PV + e] = v[e];
  }
  d.r0 = f[3]; d.a0 = f[4]; d.c0 = f[5]; d.rr0 = f[6];
  d.r1 = f[7]; d.a1 = f[8]; d.c1 = f[9]; d.rr1 = f[10];
  const T num = x - f[0];
  const T q = quotient_fast(num, f[1], f[2]);
  d.t = low ? -q : q;  // Low: -(x - g1) / h01 = -((x - g1) / h01), zeros included
  d.fast = f[11] != (T)0 && fast_numerator(num);
  return loc;
}

template <typename T, int N, bool FMA, bool LDS>
__device__ __forceinline__ void cubic_rectilinear_body(const RectArgs<T, N>& a, const unsigned char* lds) {
  const unsigned char* axbase = LDS ? lds : a.ax.image;
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.npts; i += nthreads) {
    CubicDimRect<T> dim[N];
    unsigned base = 0;
#pragma unroll
    for (int d = 0; d < N; ++d) {
      const T x = a.obs[d][i];
      const Axis<T> ax = make_axis<T, N>(a.ax, axbase, d);
      const int loc = cubic_rect_locate<T>(ax, x, a.linearize, /*fma_linear=*/false, dim[d]);
      base += (unsigned)loc * a.stride[d];
    }
    typedef CubicRectNode<T, FMA> Node;
    Leaf<T, 4> r = CubicTree<T, unsigned, N - 1, CubicDimRect<T>, Node>::run(a.vals, base, a.stride, dim, Node());
    stream_store(a.out + i, cubic_rect_node<FMA, T>(r.v[0], r.v[1], r.v[2], r.v[3], dim[N - 1]));  // rectilinear.rs:346-355
  }
}

template <typename T, int N, bool FMA>
__global__ void __launch_bounds__(kBlock) k_cubic_rectilinear(const RectArgs<T, N> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if (a.ax.use_lds) {
    stage_axes<T, N>(a.ax, smem_raw);
    cubic_rectilinear_body<T, N, FMA, true>(a, smem_raw);
  } else {
    cubic_rectilinear_body<T, N, FMA, false>(a, nullptr);
  }
}

// ===========================================================================
// Generic runtime-N kernel: the recursive arms of the reference's `interpn`
// (multilinear N = 7,8: regular_recursive.rs / rectilinear_recursive.rs;
//  multicubic N = 5..8: regular_recursive.rs / rectilinear_recursive.rs) and any grid whose
// element count does not fit 32-bit indexing.  Same dependency tree, written as the
// reference's flattened vertex loop (multilinear/regular.rs:347-393) with a runtime bound.
template <typename T>
struct GenericArgs {
  const T* vals;
  const T* obs[kMaxDims];
  T* out;
  unsigned long long* first_bad;
  size_t npts;
  int ndims;
  T start[kMaxDims];
  T step[kMaxDims];
  const T* grid[kMaxDims];
  int n[kMaxDims];
  unsigned long long stride[kMaxDims];
  int linearize;
  int fma_index;   // linear regular: index_zero_loc fused (flattened arm, N <= 6)
  int fma_linear;  // the reference's recursive arm (N >= 5): cubic rectilinear's linearized branch is fused; cubic regular's OutsideLow k1 is not
};

// Out-of-line node evaluators for the runtime-N kernel (keeps its code size bounded).
template <bool FMA, typename T>
__device__ __attribute__((noinline)) T cubic_rect_node_ool(T v0, T v1, T v2, T v3, int sat, int linear, int fma_linear,
                                                           T t, T r0, T a0, T c0, T r1, T a1, T c1) {
  CubicDimRect<T> dr;
  dr.sat = sat; dr.linear = linear; dr.fma_linear = fma_linear; dr.t = t;
  dr.r0 = r0; dr.a0 = a0; dr.c0 = c0; dr.r1 = r1; dr.a1 = a1; dr.c1 = c1;
  return cubic_rect_node<FMA, T>(v0, v1, v2, v3, dr);
}

template <typename T, int METHOD, int KIND, bool FMA>
__global__ void __launch_bounds__(kBlock) k_generic(const GenericArgs<T> a) {
  constexpr int FP = METHOD == kLinear ? 2 : 4;
  constexpr int BITS = METHOD == kLinear ? 1 : 2;
  const int N = a.ndims;
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.npts; i += nthreads) {
    T tlin[kMaxDims];
    CubicDimRegular<T> dreg[kMaxDims];
    // per-dimension state of the rectilinear cubic node, kept as separate arrays (runtime-indexed)
    int rc_sat[kMaxDims], rc_lin[kMaxDims];
    T rc_t[kMaxDims], rc_r0[kMaxDims], rc_a0[kMaxDims], rc_c0[kMaxDims], rc_r1[kMaxDims], rc_a1[kMaxDims], rc_c1[kMaxDims];
    unsigned long long base = 0;
    bool ok = true;
    for (int d = 0; d < N; ++d) {
      const T x = a.obs[d][i];
      int loc;
      if constexpr (KIND == kRegular) {
        T floc;
        ok &= regular_floc<T>(x, a.start[d], a.step[d], &floc);
        if constexpr (METHOD == kCubic) ok &= floc != (T)-9223372036854775808.0;
        if constexpr (METHOD == kLinear) {
          loc = clamp_loc<T>(floc, a.n[d] - 2);
          const T izl = (FMA && a.fma_index) ? dev_fma<T>(a.step[d], (T)loc, a.start[d])
                                             : mul_add<false>(a.step[d], (T)loc, a.start[d]);
          tlin[d] = (x - izl) / a.step[d];
        } else {
          const T n = (T)a.n[d];
          loc = clamp_loc<T>(floc - (T)1, a.n[d] - 4);
          int sat;
          bool outside;
          if (floc < (T)0) { sat = kSatLow; outside = true; }
          else if (floc == (T)0) { sat = kSatLow; outside = false; }
          else if (floc > n - (T)2) { sat = kSatHigh; outside = true; }
          else if (floc == n - (T)2) { sat = kSatHigh; outside = false; }
          else { sat = kSatNone; outside = false; }
          const T iol = mul_add<false>(a.step[d], (T)(loc + 1), a.start[d]);
          const T t = (x - iol) / a.step[d];
          dreg[d].sat = sat;
          dreg[d].linear = (outside && a.linearize) ? 1 : 0;
          dreg[d].k1_plain = (a.fma_linear != 0 && sat == kSatLow && outside) ? 1 : 0;  // recursive arm's OutsideLow (regular_recursive.rs:536)
          dreg[d].tt = sat == kSatLow ? -t : (sat == kSatHigh ? t - (T)1 : t);
        }
      } else {
        const T* g = a.grid[d];
        if constexpr (METHOD == kLinear) {
          loc = partition_point_lt<T>(g, a.n[d], x) - 1;
          loc = loc > 0 ? loc : 0;
          loc = loc < a.n[d] - 2 ? loc : a.n[d] - 2;
          const T x0 = g[loc];
          const T x1 = g[loc + 1];
          const T step = x1 - x0;
          tlin[d] = (x - x0) / step;
        } else {
          Axis<T> ax;
          ax.g = g; ax.tab = nullptr; ax.n = a.n[d]; ax.M = 0; ax.g0 = (T)0; ax.scale = (T)0;
          CubicDimRect<T> dr;
          loc = cubic_rect_locate<T>(ax, x, a.linearize, a.fma_linear != 0, dr);
          rc_sat[d] = dr.sat; rc_lin[d] = dr.linear; rc_t[d] = dr.t;
          rc_r0[d] = dr.r0; rc_a0[d] = dr.a0; rc_c0[d] = dr.c0;
          rc_r1[d] = dr.r1; rc_a1[d] = dr.a1; rc_c1[d] = dr.c1;
        }
      }
      base += (unsigned long long)loc * a.stride[d];
    }
    if (!ok) atomicMin(a.first_bad, (unsigned long long)i);

    auto node = [&](const T* v, int d) -> T {
      if constexpr (METHOD == kLinear) {
        const T y0 = v[0];
        const T dy = v[1] - y0;
        return mul_add<FMA>(tlin[d], dy, y0);
      } else if constexpr (KIND == kRegular) {
        return cubic_regular_node<FMA, T, true>(v[0], v[1], v[2], v[3], dreg[d]);
      } else {
        return cubic_rect_node_ool<FMA, T>(v[0], v[1], v[2], v[3], rc_sat[d], rc_lin[d], a.fma_linear, rc_t[d],
                                           rc_r0[d], rc_a0[d], rc_c0[d], rc_r1[d], rc_a1[d], rc_c1[d]);
      }
    };

    T store[kMaxDims][FP];
    const unsigned long long nverts = 1ull << (BITS * N);
    for (unsigned long long v = 0; v < nverts; ++v) {
      unsigned long long idx = base;
      for (int k = 0; k < N; ++k) idx += ((v >> (BITS * k)) & (FP - 1)) * a.stride[k];
      store[0][v & (FP - 1)] = a.vals[idx];
      for (int j = 1; j < N; ++j) {
        const unsigned long long q = 1ull << (BITS * j);
        if (((v + 1) & (q - 1)) == 0) {
          const int p = (int)((((v + 1) >> (BITS * j)) - 1) & (FP - 1));
          store[j][p] = node(store[j - 1], j - 1);
        }
      }
    }
#ifdef INTERPN_DEBUG_PRINT
    if (i < 16) printf("i=%d base=%llu v=%g %g %g %g res=%g\n", (int)i, base, (double)store[N-1][0], (double)store[N-1][1], (double)store[N-1][2], (double)store[N-1][3], (double)node(store[N - 1], N - 1));
#endif
    a.out[i] = node(store[N - 1], N - 1);
  }
}

// The same vertex loop with the dimension count as a template parameter: every per-dimension
// array is statically indexed (registers instead of scratch), the walk over the 2^(N-1) /
// 4^(N-1) leaf rows is uniform across the wave (row offsets come from the scalar unit), and a
// finished level hands its value upwards through a carry chain.  Serves the recursive arms
// (multilinear N = 7,8; multicubic N = 5..8) about 4-6x faster than the runtime-N form.
template <typename T, int METHOD, int KIND, bool FMA, int N, bool VEC>
__global__ void __launch_bounds__(kBlock) k_generic_n(const GenericArgs<T> a) {
  constexpr int FP = METHOD == kLinear ? 2 : 4;
  constexpr int BITS = METHOD == kLinear ? 1 : 2;
  constexpr unsigned long long NW = 1ull << (BITS * (N - 1));
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < a.npts; i += nthreads) {
    T tlin[N];
    CubicDimRegular<T> dreg[N];
    CubicDimRect<T> drect[N];
    unsigned long long base = 0;
    bool ok = true;
#pragma unroll
    for (int d = 0; d < N; ++d) {
      const T x = stream_load(a.obs[d] + i);
      int loc;
      if constexpr (KIND == kRegular) {
        T floc;
        ok &= regular_floc<T>(x, a.start[d], a.step[d], &floc);
        if constexpr (METHOD == kCubic) ok &= floc != (T)-9223372036854775808.0;
        if constexpr (METHOD == kLinear) {
          loc = clamp_loc<T>(floc, a.n[d] - 2);
          const T izl = (FMA && a.fma_index) ? dev_fma<T>(a.step[d], (T)loc, a.start[d])
                                             : mul_add<false>(a.step[d], (T)loc, a.start[d]);
          tlin[d] = (x - izl) / a.step[d];
        } else {
          const T n = (T)a.n[d];
          loc = clamp_loc<T>(floc - (T)1, a.n[d] - 4);
          int sat;
          bool outside;
          if (floc < (T)0) { sat = kSatLow; outside = true; }
          else if (floc == (T)0) { sat = kSatLow; outside = false; }
          else if (floc > n - (T)2) { sat = kSatHigh; outside = true; }
          else if (floc == n - (T)2) { sat = kSatHigh; outside = false; }
          else { sat = kSatNone; outside = false; }
          const T iol = mul_add<false>(a.step[d], (T)(loc + 1), a.start[d]);
          const T t = (x - iol) / a.step[d];
          dreg[d].sat = sat;
          dreg[d].linear = (outside && a.linearize) ? 1 : 0;
          dreg[d].k1_plain = (a.fma_linear != 0 && sat == kSatLow && outside) ? 1 : 0;  // recursive arm's OutsideLow (regular_recursive.rs:536)
          dreg[d].tt = sat == kSatLow ? -t : (sat == kSatHigh ? t - (T)1 : t);
        }
      } else {
        const T* g = a.grid[d];
        if constexpr (METHOD == kLinear) {
          loc = partition_point_lt<T>(g, a.n[d], x) - 1;
          loc = loc > 0 ? loc : 0;
          loc = loc < a.n[d] - 2 ? loc : a.n[d] - 2;
          const T x0 = g[loc];
          const T x1 = g[loc + 1];
          const T step = x1 - x0;
          tlin[d] = (x - x0) / step;
        } else {
          Axis<T> ax;
          ax.g = g; ax.tab = nullptr; ax.n = a.n[d]; ax.M = 0; ax.g0 = (T)0; ax.scale = (T)0;
          loc = cubic_rect_locate<T>(ax, x, a.linearize, a.fma_linear != 0, drect[d]);
        }
      }
      base += (unsigned long long)loc * a.stride[d];
    }
    if (!ok) atomicMin(a.first_bad, (unsigned long long)i);

    auto node = [&](T v0, T v1, T v2, T v3, int j) -> T {
      if constexpr (METHOD == kLinear) return mul_add<FMA>(tlin[j], v1 - v0, v0);
      else if constexpr (KIND == kRegular) return cubic_regular_node<FMA, T, true>(v0, v1, v2, v3, dreg[j]);
      else return cubic_rect_node<FMA, T>(v0, v1, v2, v3, drect[j]);
    };
    T result;
    const T* row0 = a.vals + base;
    if constexpr (VEC) {
      // One vector load per row of FP consecutive last-dimension values: FP trees over dims
      // 0..N-2 advance side by side, the last dimension is reduced at the end.  Every node sees
      // the operands the reference's tree gives it, so the result is unchanged; the L2 sees FP
      // times fewer requests.
      T store[N - 1][FP][FP];  // [level][slot][tree]
      T val[FP];
#pragma unroll 1
      for (unsigned long long w = 0; w < NW; ++w) {
        unsigned long long off = 0;
#pragma unroll
        for (int k = 0; k < N - 1; ++k) off += ((w >> (BITS * k)) & (FP - 1)) * a.stride[k];
        const Leaf<T, FP> leaf = load_leaf<T, FP>(row0 + off);
#pragma unroll
        for (int c = 0; c < FP; ++c) val[c] = leaf.v[c];
        bool carry = true;
#pragma unroll
        for (int j = 0; j < N - 1; ++j) {
          if (carry) {
            const int p = (int)((w >> (BITS * j)) & (FP - 1));
#pragma unroll
            for (int q = 0; q < FP; ++q)
              if (q == p) {
#pragma unroll
                for (int c = 0; c < FP; ++c) store[j][q][c] = val[c];
              }
            if (p == FP - 1) {
#pragma unroll
              for (int c = 0; c < FP; ++c)
                val[c] = node(store[j][0][c], store[j][1][c], store[j][FP - 2][c], store[j][FP - 1][c], j);
            } else {
              carry = false;
            }
          }
        }
      }
      result = node(val[0], val[1], val[FP - 2], val[FP - 1], N - 1);
    } else {
      T store[N][FP];
      result = (T)0;
#pragma unroll 1
      for (unsigned long long w = 0; w < NW; ++w) {
        // leaf row of this step: footprint digits of dims 1..N-1 are the base-FP digits of w
        unsigned long long off = 0;
#pragma unroll
        for (int k = 1; k < N; ++k) off += ((w >> (BITS * (k - 1))) & (FP - 1)) * a.stride[k];
        T leaf[FP];
#pragma unroll
        for (int k = 0; k < FP; ++k) leaf[k] = row0[off + (unsigned long long)k * a.stride[0]];
        T val = node(leaf[0], leaf[1], leaf[FP - 2], leaf[FP - 1], 0);
        bool carry = true;
#pragma unroll
        for (int j = 1; j < N; ++j) {
          if (carry) {
            const int p = (int)((w >> (BITS * (j - 1))) & (FP - 1));
#pragma unroll
            for (int q = 0; q < FP; ++q)
              if (q == p) store[j][q] = val;
            if (p == FP - 1) val = node(store[j][0], store[j][1], store[j][FP - 2], store[j][FP - 1], j);
            else carry = false;
          }
        }
        result = val;  // after the last row every level has carried: val is the root
      }
    }
    stream_store(a.out + i, result);
  }
}

// ---------------------------------------------------------------------------
// Launch geometry
inline unsigned grid_blocks(size_t npts, int points_per_thread, const LaunchConfig& cfg) {
  size_t per_block = (size_t)kBlock * points_per_thread;
  size_t want = (npts + per_block - 1) / per_block;
  size_t cap = (size_t)cfg.num_cus * cfg.blocks_per_cu;
  if (want < 1) want = 1;
  return (unsigned)(want < cap ? want : cap);
}

// One pass over the batch (every workgroup runs its grid-stride loop once): the dispatcher
// balances the XCDs dynamically.  For kernels without per-workgroup set-up (regular grids).
inline unsigned one_pass_blocks(size_t npts, int points_per_thread) {
  size_t per_block = (size_t)kBlock * points_per_thread;
  size_t want = (npts + per_block - 1) / per_block;
  if (want < 1) want = 1;
  // at most 2^23 workgroups: the dispatch packet counts work-items in 32 bits; the grid-stride
  // loop of the kernel covers anything beyond
  return (unsigned)(want < ((size_t)1 << 23) ? want : ((size_t)1 << 23));
}

}  // namespace interpn
