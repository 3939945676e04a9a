// nearest::regular / nearest::rectilinear (reference: src/nearest/regular.rs:41-101, :234-317;
// src/nearest/rectilinear.rs:36-60, :193-262).  The same per-dimension index stage as the
// multilinear kernels, then ONE gather per point: the node at origin + (dt <= 0.5 ? 0 : 1).
// 64-bit indexing throughout (a single gather, nothing to save with 32-bit offsets).
#include "lane_axes.h"
#include "rect_args.h"

namespace interpn {

template <typename T, int N>
struct NearestArgs {
  const T* vals;
  const T* obs[N];
  T* out;
  unsigned long long* first_bad;
  size_t npts;
  T start[N];
  T step[N];
  int n[N];
  unsigned long long stride[N];
  AxisArgs<T, N> ax;
};

template <typename T, int N, bool RECT, bool FMA, bool LDS, int AXR, int PPL>
__device__ __forceinline__ void nearest_body(const NearestArgs<T, N>& a, const unsigned char* lds) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  const unsigned char* axbase = LDS ? lds : a.ax.image;
  const T half = (T)1 / ((T)1 + (T)1);
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  LaneAxes<T, N> la;
  if constexpr (RECT && AXR != 0) la = load_lane_axes<T, N, AXR>(a.ax);
  // Wave-uniform trip count: with the axes in lanes every lane of a wave must stay active for the
  // cross-lane reads, so the loop runs over the wave's first slot and dead lanes are masked at
  // the store only (they search for coordinate 0).  A slot is PPL consecutive points of one lane.
  const size_t nslots = (a.npts + PPL - 1) / PPL;
  const size_t wave0 = (size_t)blockIdx.x * kBlock + (threadIdx.x & ~63u);
  for (size_t w = wave0; w < nslots; w += nthreads) {
    const size_t i0 = (w + (threadIdx.x & 63u)) * PPL;
    bool live[PPL];
#pragma unroll
    for (int h = 0; h < PPL; ++h) live[h] = i0 + h < a.npts;
    T xin[PPL][N];
    if (PPL == 2) {
#pragma unroll
      for (int d = 0; d < N; ++d) {
        T2 v;
        v.x = RECT ? (T)0 : a.start[d];
        v.y = v.x;
        if (live[PPL - 1]) v = stream_load(reinterpret_cast<const T2*>(a.obs[d] + i0));
        else if (live[0]) v.x = stream_load(a.obs[d] + i0);
        xin[0][d] = v.x;
        xin[PPL - 1][d] = v.y;
      }
    } else {
#pragma unroll
      for (int d = 0; d < N; ++d) xin[0][d] = live[0] ? stream_load(a.obs[d] + i0) : (RECT ? (T)0 : a.start[d]);
    }
    int cell[PPL][N];
    T x0r[PPL][N], x1r[PPL][N];
    if constexpr (RECT && AXR != 0) lane_axes_locate<T, N, PPL, AXR>(a.ax, la, xin, cell, x0r, x1r);
    T resv[PPL];
#pragma unroll
    for (int h = 0; h < PPL; ++h) {
      unsigned long long idx = 0;
      bool ok = true;
#pragma unroll
      for (int d = 0; d < N; ++d) {
        const T x = xin[h][d];
        int loc;
        T dt;
        if (RECT) {
          T x0, x1;
          if constexpr (AXR != 0) {
            loc = cell[h][d];
            x0 = x0r[h][d];
            x1 = x1r[h][d];
          } else {
            const Axis<T> ax = make_axis<T, N>(a.ax, axbase, d);
            loc = axis_cell<T>(ax, x, &x0, &x1);  // nearest/rectilinear.rs:248-262, :223-224
          }
          const T step = x1 - x0;
          dt = (x - x0) / step;  // rectilinear.rs:223-227
        } else {
          T floc;
          ok &= regular_floc<T>(x, a.start[d], a.step[d], &floc);  // nearest/regular.rs:306-309
          loc = clamp_loc<T>(floc, a.n[d] - 2);
          const T izl = mul_add<FMA>(a.step[d], (T)loc, a.start[d]);  // regular.rs:272-275
          dt = (x - izl) / a.step[d];
        }
        const int offset = (dt <= half) ? 0 : 1;  // regular.rs:283-287 (NaN compares false => 1)
        idx += (unsigned long long)(loc + offset) * a.stride[d];
      }
      if (!RECT && !ok && live[h]) atomicMin(a.first_bad, (unsigned long long)(i0 + h));
      resv[h] = live[h] ? a.vals[idx] : (T)0;
    }
    if (PPL == 2) {
      if (live[PPL - 1]) {
        T2 v;
        v.x = resv[0];
        v.y = resv[PPL - 1];
        stream_store(reinterpret_cast<T2*>(a.out + i0), v);
      } else if (live[0]) {
        stream_store(a.out + i0, resv[0]);
      }
    } else if (live[0]) {
      stream_store(a.out + i0, resv[0]);
    }
  }
}

// AXR != 0 (rectilinear, every axis <= 64 coordinates): axes in lanes, see lane_axes.h.
// PPL = 2: two consecutive points per lane, vector coordinate / result accesses (needs obs / out
// aligned to 2*sizeof(T); the launcher checks).
template <typename T, int N, bool RECT, bool FMA, int AXR = 0, int PPL = 1>
__global__ void __launch_bounds__(kBlock) k_nearest(const NearestArgs<T, N> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  if constexpr (RECT && AXR != 0) {
    nearest_body<T, N, RECT, FMA, false, AXR, PPL>(a, nullptr);
  } else if (RECT && a.ax.use_lds) {
    stage_axes<T, N>(a.ax, smem_raw);
    nearest_body<T, N, RECT, FMA, true, 0, PPL>(a, smem_raw);
  } else {
    nearest_body<T, N, RECT, FMA, false, 0, PPL>(a, nullptr);
  }
}

template <typename T, int N>
static hipError_t launch_n(const GridDesc& g, const T* const* obs, T* out, size_t npts, unsigned long long* first_bad,
                           hipStream_t stream) {
  NearestArgs<T, N> a;
  a.vals = static_cast<const T*>(g.vals);
  a.out = out;
  a.first_bad = first_bad;
  a.npts = npts;
  unsigned long long acc = 1;
  for (int d = N - 1; d >= 0; --d) {
    a.obs[d] = obs[d];
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
    a.stride[d] = acc;
    acc *= (unsigned long long)g.n[d];
  }
  size_t lds = 0;
  a.ax.use_lds = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  if (g.kind == kRectilinear) lds = fill_axis_args<T, N>(g, a.ax, /*big_lds=*/true);  // (per-bucket records measured slower here: 3-D 80^3 1.32 vs 1.17 ms)
  const int axr = lane_axes_mode(g);  // rectilinear axes of <= 64 coordinates: searched across lanes, no LDS image
  if (axr) lds = 0;
  // Two points per lane when every stream is aligned to 2*sizeof(T) (`ppl` option = 1: scalar form).
  bool aligned = (reinterpret_cast<uintptr_t>(out) % (2 * sizeof(T))) == 0;
  for (int d = 0; d < N; ++d) aligned = aligned && (reinterpret_cast<uintptr_t>(obs[d]) % (2 * sizeof(T))) == 0;
  const int ppl = (aligned && g.cfg.ppl != 1) ? 2 : 1;
  // lane-resident axes cost six small loads per wave: four rows per wave amortise them
  const unsigned blocks = ((g.kind == kRegular || axr) && !g.cfg.persistent) ? one_pass_blocks(npts, ppl * (axr ? 4 : 1))
                                                                            : grid_blocks(npts, ppl, g.cfg);
#define GO2(RECT, FMA, AXR, PPL) do { g.tag.set("k_nearest", {N, RECT, FMA, AXR, PPL}, 0b00110u); hipLaunchKernelGGL((k_nearest<T, N, RECT, FMA, AXR, PPL>), dim3(blocks), dim3(kBlock), lds, stream, a); } while (0)
#define GO(RECT, FMA, AXR) do { if (ppl == 2) GO2(RECT, FMA, AXR, 2); else GO2(RECT, FMA, AXR, 1); } while (0)
  if (g.kind == kRegular) { if (g.fma) GO(false, true, 0); else GO(false, false, 0); }
  else if (axr == 2) GO(true, true, 2);  // no FMA site in the rectilinear path
  else if (axr == 3) GO(true, true, 3);
  else if (axr == 1) GO(true, true, 1);
  else GO(true, true, 0);
#undef GO
#undef GO2
  return hipGetLastError();
}

template <typename T>
hipError_t launch_nearest(const GridDesc& g, const T* const* obs, T* out, size_t npts, unsigned long long* first_bad,
                          hipStream_t stream) {
  switch (g.ndims) {
    case 1: return launch_n<T, 1>(g, obs, out, npts, first_bad, stream);
    case 2: return launch_n<T, 2>(g, obs, out, npts, first_bad, stream);
    case 3: return launch_n<T, 3>(g, obs, out, npts, first_bad, stream);
    case 4: return launch_n<T, 4>(g, obs, out, npts, first_bad, stream);
    case 5: return launch_n<T, 5>(g, obs, out, npts, first_bad, stream);
    case 6: return launch_n<T, 6>(g, obs, out, npts, first_bad, stream);
    default: return hipErrorInvalidValue;
  }
}

template hipError_t launch_nearest<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t);
template hipError_t launch_nearest<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t);

}  // namespace interpn
