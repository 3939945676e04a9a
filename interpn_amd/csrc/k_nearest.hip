// nearest::regular / nearest::rectilinear (reference: src/nearest/regular.rs:41-101, :234-317;
// src/nearest/rectilinear.rs:36-60, :193-262).  The same per-dimension index stage as the
// multilinear kernels, then ONE gather per point: the node at origin + (dt <= 0.5 ? 0 : 1).
// 64-bit indexing throughout (a single gather, nothing to save with 32-bit offsets).
#include <atomic>
#include <type_traits>

#include "lane_axes.h"
#include "sweep_rounds.h"
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
  const unsigned* gate;  // gated launch (GridDesc::launch_gate): null, or a word that must be non-zero for this launch to do anything
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
  if (a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // (launch-uniform)
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

// ---- Sweep evaluation of large 2-D / 3-D batches on regular grids (round 5, third session) -------------------------------
// linear_sweep.h's scheme (points ordered on chip by leading cell index inside the window the chip holds, waves walking that
// index in step with a clock, rounds on demand) around this file's row: one gather per point from the C-ordered grid.  The
// index stage is the multilinear one; on regular grids it runs without divide sequences (interpn_device.h::step_cell_fast).
template <typename T, int N>
struct NearestSweepArgs {
  SweepRounds<T, N> r;   // sweep_rounds.h: the streams, the sort key, the rounds
  const T* vals;
  unsigned long long* first_bad;
  T start[N], step[N], rstep[N];  // rstep: RN(1 / step), a division in T on the host (interpn_device.h::step_cell_fast)
  int n[N];
  unsigned long long stride[N];
  unsigned fastdiv;      // != 0: every step lies where step_cell_fast is the reference's value
};

template <typename T, int N, bool FMA, int K, int KL, int THREADS>
__global__ void __launch_bounds__(THREADS) k_nearest_sweep(const NearestSweepArgs<T, N> s) {
  typedef SweepRoundsLds<T, N, K, KL> L;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  sweep_rounds<T, N, K, KL, THREADS, 0, L>(s.r, smem_raw, [&](auto, const T (&xr)[N], size_t gi) -> T {
    T t[N];
    int loc[N];
    bool ok = true;
    bool exact = s.fastdiv != 0;
#pragma unroll
    for (int d = 0; d < N; ++d) {
      const StepCell<T> sc = step_cell_fast<FMA>(xr[d], s.start[d], s.step[d], s.rstep[d], s.n[d] - 2);
      t[d] = sc.t;
      loc[d] = sc.loc;
      exact = exact && sc.exact;
    }
    if (__any(!exact)) {  // a lane on a grid plane, far outside, not finite ...: the reference's operations as they stand, for the wave
#pragma unroll
      for (int d = 0; d < N; ++d) {
        T floc;
        ok &= regular_floc<T>(xr[d], s.start[d], s.step[d], &floc);  // nearest/regular.rs:306-309
        const int l = clamp_loc<T>(floc, s.n[d] - 2);
        const T izl = mul_add<FMA>(s.step[d], (T)l, s.start[d]);     // regular.rs:272-275
        t[d] = (xr[d] - izl) / s.step[d];
        loc[d] = l;
      }
    }
    if (!ok && gi < s.r.npts) atomicMin(s.first_bad, (unsigned long long)gi);
    // nearest/regular.rs:283-287: the node at origin + (dt <= 0.5 ? 0 : 1) per dimension (NaN compares false => 1); one gather
    const T half = (T)1 / ((T)1 + (T)1);
    unsigned long long idx = 0;
#pragma unroll
    for (int d = 0; d < N; ++d) idx += (unsigned long long)(loc[d] + ((t[d] <= half) ? 0 : 1)) * s.stride[d];
    return s.vals[idx];
  });
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
  unsigned blocks = ((g.kind == kRegular || axr) && !g.cfg.persistent) ? one_pass_blocks(npts, ppl * (axr ? 4 : 1))
                                                                      : grid_blocks(npts, ppl, g.cfg);
  a.gate = g.launch_gate;
  // fewer, fatter workgroups: mostly they return at once (option gated_iters: 4 — lattices walked along the table's slowest
  // dimension lose with fatter ones, 16: -11 %, while those along its fastest gain little beyond 4: tools/gated_iters_lattices.py)
  // (this kernel: half that factor — 128^3 on a lattice 0.566 ms at 2, 0.59 at 4, 0.61 at 8)
  if ((a.gate || g.launch_fat) && g.cfg.gated_iters > 3) blocks = (blocks + (unsigned)(g.cfg.gated_iters / 2) - 1) / (unsigned)(g.cfg.gated_iters / 2);
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

// ---- host side of the nearest-neighbour sweep evaluation ---------------------------------------------------------------
namespace {
constexpr int kNearestSweepThreads = 768;
// rows of 64 points per wave and round: in registers + parked in LDS
template <typename T, int N> constexpr int nearest_sweep_rows() { return sizeof(T) == 8 ? (N == 2 ? 16 : 12) : (N == 2 ? 32 : 24); }
template <typename T, int N> constexpr int nearest_sweep_parked() { return sizeof(T) == 8 ? 4 : 0; }
}  // namespace

// 0 = never for this handle, 1 = not for this batch, 2 = yes.
int nearest_sweep_applies(const GridDesc& g, size_t npts) {
  if (g.method != kNearest || (g.ndims != 2 && g.ndims != 3) || g.kind != kRegular || g.cfg.sweep == 0 || g.cfg.force_generic) return 0;
  const size_t lds = (size_t)SweepRoundsLds<double, 3, 12, 4>::kWave * (kNearestSweepThreads / 64) + SweepRoundsLds<double, 3, 12, 4>::kWorkgroup;
  if ((long long)lds > g.cfg.lds_per_cu) return 0;
  if (g.cfg.sweep > 0) return 2;
  const size_t cus = (size_t)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
  const size_t rows = g.dtype == kF64 ? (g.ndims == 2 ? 20 : 16) : (g.ndims == 2 ? 32 : 24);
  // (profiles/r05_nearest_sweep.jsonl, f64: 3-D 64^3 0.89 -> 0.73 ms per 1e8 points and 0.275 -> 0.260 per 3e7, 128^3 1.65 -> 1.03 and
  //  0.50 -> 0.33; 2-D 64^2 0.61 -> 0.51, 1000^2 1.08 -> 0.74: eight rounds per wave where the L2 holds the grid, four beyond)
  size_t grid_bytes = g.dtype == kF64 ? 8 : 4;
  for (int d = 0; d < g.ndims; ++d) grid_bytes *= (size_t)g.n[d];
  if (npts < (grid_bytes > thresholds(g.cfg).table_l2_sized ? 4 : 8) * rows * kNearestSweepThreads * cus) return 1;
  return 2;
}

template <typename T, int N>
static hipError_t launch_sweep_n(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad, void* work, hipStream_t stream) {
  constexpr int K = nearest_sweep_rows<T, N>(), KL = nearest_sweep_parked<T, N>(), TH = kNearestSweepThreads;
  NearestSweepArgs<T, N> s;
  SweepRounds<T, N>& r = s.r;
  s.vals = static_cast<const T*>(g.vals);
  r.out = static_cast<T*>(out);
  s.first_bad = first_bad;
  r.npts = npts;
  s.fastdiv = 1;
  unsigned long long acc = 1;
  for (int d = N - 1; d >= 0; --d) {
    r.obs[d] = static_cast<const T*>(obs[d]);
    s.start[d] = (T)g.start[d];
    r.absent[d] = s.start[d];
    s.step[d] = (T)g.step[d];
    s.n[d] = g.n[d];
    s.stride[d] = acc;
    acc *= (unsigned long long)g.n[d];
    const volatile T one = (T)1;
    s.rstep[d] = one / (T)g.step[d];
    const double mag = g.step[d] < 0 ? -g.step[d] : g.step[d];
    if (!(mag >= StepCellRange<T>::lo && mag <= StepCellRange<T>::hi)) s.fastdiv = 0;
  }
  r.key_start = (T)g.start[0];
  r.key_scale = (T)(1.0 / g.step[0]);
  if (!(r.key_scale > 0) || !(r.key_scale < (T)1e30)) r.key_scale = 0;
  r.key_cells = g.n[0] - 2;
  r.key_shift = 0;
  while (((g.n[0] - 2) >> r.key_shift) >= 64) ++r.key_shift;
  const size_t chunk = (size_t)64 * (K + KL);
  const size_t rounds = (npts + chunk - 1) / chunk;
  if (rounds > 0xFFFFFFF0ull) return hipErrorInvalidValue;
  r.rounds = (unsigned)rounds;
  r.per_shard = (r.rounds + 7u) / 8u;
  r.period = g.cfg.sweep_period > 0 ? (unsigned)g.cfg.sweep_period : 0u;
  r.period_default = 2000;
  r.gated = g.sweep_gated ? 1u : 0u;
  r.stamps = nullptr;
  r.work = static_cast<SweepWork*>(work);
  const unsigned cus = (unsigned)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
  unsigned blocks = cus;
  const unsigned need = (r.rounds + (TH / 64) - 1) / (TH / 64);
  if (blocks > need) blocks = need;
  const size_t lds = (size_t)SweepRoundsLds<T, N, K, KL>::kWave * (TH / 64) + SweepRoundsLds<T, N, K, KL>::kWorkgroup;
  auto launch = [&](auto kern, bool fma) -> hipError_t {
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    g.tag.set("k_nearest_sweep", {N, fma, K, KL, TH}, 0b00010u);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(TH), lds, stream, s);
    return hipGetLastError();
  };
  return g.fma ? launch(k_nearest_sweep<T, N, true, K, KL, TH>, true) : launch(k_nearest_sweep<T, N, false, K, KL, TH>, false);
}

hipError_t launch_nearest_sweep(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad, void* work, hipStream_t stream) {
  if (g.method != kNearest || !work || npts == 0) return hipErrorInvalidValue;
  for (int d = 0; d < g.ndims; ++d)
    if (reinterpret_cast<uintptr_t>(obs[d]) % 16) return hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(out) % 16) return hipErrorInvalidValue;
  if (g.ndims == 2) return g.dtype == kF64 ? launch_sweep_n<double, 2>(g, obs, out, npts, first_bad, work, stream) : launch_sweep_n<float, 2>(g, obs, out, npts, first_bad, work, stream);
  if (g.ndims == 3) return g.dtype == kF64 ? launch_sweep_n<double, 3>(g, obs, out, npts, first_bad, work, stream) : launch_sweep_n<float, 3>(g, obs, out, npts, first_bad, work, stream);
  return hipErrorInvalidValue;
}

}  // namespace interpn
