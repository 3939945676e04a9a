// 1-D multilinear on a rectilinear axis (multilinear::rectilinear::interpn with one dimension,
// src/multilinear/rectilinear.rs:49-83, :244-370) from one record per search bucket.
//
// The general rectilinear kernel finds the cell with a bucket table plus a short scan, then
// fetches the two bracketing coordinates and (1-D) the two values.  While the axis image fits its
// LDS budget (60 KiB: up to ~5000 coordinates) that runs at the stream rate (0.33 ms per 1e8
// points); beyond it the five dependent gathers go through L1/L2 and are all a 1-D point costs
// (4096-point axis: 1.25 ms, 65536 points: 1.57 ms, against 0.52-0.68 ms on a regular axis).  For
// those axes: when M uniform buckets over the axis span hold at most ONE coordinate each (sorted
// finite axes; M is doubled from 2n until that holds, up to a 128 MiB table), everything a point
// in bucket b can need fits one 64-byte record (32 B in f32):
//     idx0   = number of coordinates in buckets < b          (what tab[b] holds in the LDS search)
//     gprobe = g[min(idx0, n-1)]     the only coordinate that can share bucket b with x
//     cell A = clamp(idx0 - 1, 0, n-2)   if !(idx0 < n && gprobe < x)      -> (ga, gb, va, vb)
//     cell B = clamp(idx0,     0, n-2)   otherwise; B = A or A + 1         -> (gb, gc, vb, vc)
// bucket_of() is monotone and is the function the records are built with, so coordinates in
// earlier buckets are < x and those in later buckets are not, for any rounding: idx0 + (gprobe < x)
// is exactly core::slice::partition_point(|g| g < x) (rectilinear.rs:363).  One gather of one
// record per point, then the reference's arithmetic: t = (x - x0) / (x1 - x0) (rectilinear.rs:
// 310-313), y0 + t (y1 - y0) with the fma flavour's single fused step (:339-344).
#include "interpn_kernels.h"

namespace interpn {

template <typename T>
struct __attribute__((aligned(8 * sizeof(T)))) Rec1 {
  T gprobe, ga, gb, gc, va, vb, vc, flags;  // flags: 1 = idx0 < n (compare allowed), 2 = cell B is A + 1
};

template <typename T>
struct Rec1Args {
  const Rec1<T>* recs;
  const T* obs;
  T* out;
  size_t npts;
  T g0, scale;
  int M;
};

template <typename T, bool FMA>
__device__ __forceinline__ T rec1_eval(const Rec1Args<T>& a, T x) {
  const int b = bucket_of<T>(x, a.g0, a.scale, a.M);  // NaN -> bucket 0 -> cell 0, like partition_point
  const Rec1<T> r = a.recs[b];
  const int fl = (int)r.flags;
  const bool up = (fl & 1) && (r.gprobe < x);
  const bool sel = up && (fl & 2);
  const T x0 = sel ? r.gb : r.ga;
  const T x1 = sel ? r.gc : r.gb;
  const T v0 = sel ? r.vb : r.va;
  const T v1 = sel ? r.vc : r.vb;
  const T step = x1 - x0;
  const T t = (x - x0) / step;            // rectilinear.rs:310-313
  return mul_add<FMA>(t, v1 - v0, v0);    // rectilinear.rs:339-344
}

// One pass of 256-lane workgroups, PPL points per lane (PPL = 2: 2*sizeof(T)-byte vector streams,
// needs obs and out aligned to that).
template <typename T, bool FMA, int PPL>
__global__ void __launch_bounds__(kBlock) k_linear1_records(const Rec1Args<T> a) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  const size_t nslots = (a.npts + PPL - 1) / PPL;
  const size_t stride = (size_t)gridDim.x * kBlock;
  for (size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x; s < nslots; s += stride) {
    const size_t i0 = s * PPL;
    if constexpr (PPL == 2) {
      if (i0 + 1 < a.npts) {
        const T2 x = stream_load(reinterpret_cast<const T2*>(a.obs + i0));
        T2 r;
        r.x = rec1_eval<T, FMA>(a, x.x);
        r.y = rec1_eval<T, FMA>(a, x.y);
        stream_store(reinterpret_cast<T2*>(a.out + i0), r);
      } else {
        stream_store(a.out + i0, rec1_eval<T, FMA>(a, stream_load(a.obs + i0)));
      }
    } else {
      stream_store(a.out + i0, rec1_eval<T, FMA>(a, stream_load(a.obs + i0)));
    }
  }
}

// Records builder: one thread per bucket.  idx0 = first k with bucket_of(g[k]) >= b (the axis is
// sorted, so bucket_of(g[k]) is non-decreasing in k): a binary search; the bucket's population
// (first k of bucket b+1 minus idx0) goes into *maxpop.
template <typename T>
__device__ __forceinline__ int first_in_bucket(const T* __restrict__ g, int n, int b, T g0, T scale, int M) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (bucket_of<T>(g[mid], g0, scale, M) < b) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_records1(const T* __restrict__ g, const T* __restrict__ vals, int n, int M, T g0,
                                                           T scale, Rec1<T>* __restrict__ recs, unsigned* __restrict__ maxpop) {
  for (int b = blockIdx.x * kBlock + threadIdx.x; b < M; b += gridDim.x * kBlock) {
    const int idx0 = first_in_bucket<T>(g, n, b, g0, scale, M);
    const int next = b + 1 < M ? first_in_bucket<T>(g, n, b + 1, g0, scale, M) : n;
    atomicMax(maxpop, (unsigned)(next - idx0));
    const int c = idx0 < n ? idx0 : n - 1;
    int la = idx0 - 1;
    la = la > 0 ? la : 0;
    la = la < n - 2 ? la : n - 2;
    int lb = idx0;
    lb = lb < n - 2 ? lb : n - 2;  // idx0 >= 0 always
    Rec1<T> r;
    r.gprobe = g[c];
    r.ga = g[la];
    r.gb = g[la + 1];
    r.gc = g[lb + 1];
    r.va = vals[la];
    r.vb = vals[la + 1];
    r.vc = vals[lb + 1];
    r.flags = (T)((idx0 < n ? 1 : 0) | (lb != la ? 2 : 0));
    recs[b] = r;
  }
}

size_t records1_bytes(const GridDesc& g, int M) { return (size_t)M * 8 * (g.dtype == kF64 ? 8 : 4); }

hipError_t build_records1(const GridDesc& g, int M, double scale, void* recs, unsigned* maxpop_dev, hipStream_t stream) {
  const unsigned blocks = (unsigned)((M + kBlock - 1) / kBlock);
  hipError_t e = hipMemsetAsync(maxpop_dev, 0, sizeof(unsigned), stream);
  if (e != hipSuccess) return e;
  if (g.dtype == kF64)
    hipLaunchKernelGGL(k_build_records1<double>, dim3(blocks), dim3(kBlock), 0, stream, static_cast<const double*>(g.grid[0]),
                       static_cast<const double*>(g.vals), g.n[0], M, g.axis_g0[0], scale, static_cast<Rec1<double>*>(recs),
                       maxpop_dev);
  else
    hipLaunchKernelGGL(k_build_records1<float>, dim3(blocks), dim3(kBlock), 0, stream, static_cast<const float*>(g.grid[0]),
                       static_cast<const float*>(g.vals), g.n[0], M, (float)g.axis_g0[0], (float)scale,
                       static_cast<Rec1<float>*>(recs), maxpop_dev);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_linear1_records(const GridDesc& g, const T* const* obs, T* out, size_t npts, hipStream_t stream) {
  Rec1Args<T> a;
  a.recs = static_cast<const Rec1<T>*>(g.bricks);
  a.obs = obs[0];
  a.out = out;
  a.npts = npts;
  a.g0 = (T)g.axis_g0[0];
  a.scale = (T)g.rec1_scale;
  a.M = g.rec1_buckets;
  const bool aligned = (reinterpret_cast<uintptr_t>(out) % (2 * sizeof(T))) == 0 &&
                       (reinterpret_cast<uintptr_t>(obs[0]) % (2 * sizeof(T))) == 0;
  const int ppl = (aligned && g.cfg.ppl != 1) ? 2 : 1;
  const unsigned blocks = one_pass_blocks(npts, ppl);
#define GO(FMA, PPL) do { g.tag.set("k_linear1_records", {FMA, PPL}, 0b01u); hipLaunchKernelGGL((k_linear1_records<T, FMA, PPL>), dim3(blocks), dim3(kBlock), 0, stream, a); } while (0)
  if (g.fma) { if (ppl == 2) GO(true, 2); else GO(true, 1); }
  else { if (ppl == 2) GO(false, 2); else GO(false, 1); }
#undef GO
  return hipGetLastError();
}

template hipError_t launch_linear1_records<double>(const GridDesc&, const double* const*, double*, size_t, hipStream_t);
template hipError_t launch_linear1_records<float>(const GridDesc&, const float* const*, float*, size_t, hipStream_t);

}  // namespace interpn
