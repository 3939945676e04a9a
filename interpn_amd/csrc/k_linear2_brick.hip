// 2-D multilinear (regular and rectilinear, f64 and f32) on a bricked copy of the grid with a
// lane-pair cooperative gather.
//
// In C order the four corners of a 2-D cell lie on two rows = two 128-B lines.  Here the grid is
// re-laid in bricks of 2 rows x KW2 columns (KW2 = 8 for f64, 16 for f32; one 128-B line), stepped
// 1 along i (every row pair is stored) and KW2-1 along j (a j-pair never leaves a brick row), i.e.
// 2 * KW2/(KW2-1) = 2.3x the grid, so that a cell's corners are one line.  The two lanes of a pair
// fetch the two row pieces of ONE point per load instruction (same line => one L2 request), then
// swap the piece they hold for the other point with a quad-permute DPP move — no LDS.  Arithmetic
// and operation order are the reference's (src/multilinear/regular.rs:296-404), results are
// bit-identical to the C-order kernel.
#include <atomic>
#include <type_traits>

#include "lane_axes.h"
#include "sweep_rounds.h"
#include "rect_args.h"

namespace interpn {

template <typename T> struct Brick2Geom {
  static constexpr int KW = 64 / (int)sizeof(T);  // columns per brick row (64 B)
  static constexpr int SJ = KW - 1;
  static constexpr int ELEMS = 2 * KW;            // 128 B
};

template <typename T>
struct Brick2Args {
  const T* bricks;
  const T* obs[2];
  T* out;
  unsigned long long* first_bad;
  size_t npts;
  T start[2];
  T step[2];
  int n[2];
  AxisArgs<T, 2> ax;
  unsigned nbj;
  const unsigned* gate;  // gated launch (GridDesc::launch_gate): null, or a word that must be non-zero for this launch to do anything
};

// swap with the neighbouring lane (lane ^ 1): quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned dpp_swap1(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ float dpp_swap1(float v) { return __uint_as_float(dpp_swap1(__float_as_uint(v))); }
__device__ __forceinline__ double dpp_swap1(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = dpp_swap1((unsigned)b), hi = dpp_swap1((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// AXR != 0 (rectilinear, both axes <= 64 coordinates): axes in lanes, see lane_axes.h.
// PPL = 2: a lane owns two consecutive points, coordinates and results move as 2*sizeof(T)-byte
// vectors (as in the 3-D kernel; needs obs / out aligned to 2*sizeof(T), the launcher checks).
template <typename T, bool RECT, bool FMA, int AXR = 0, int PPL = 1>
__global__ void __launch_bounds__(kBlock) k_linear2_brick(const Brick2Args<T> a) {
  typedef typename LeafVec<T, 2>::type P;
  typedef T T2 __attribute__((ext_vector_type(2)));
  constexpr int KW = Brick2Geom<T>::KW;
  constexpr int SJ = Brick2Geom<T>::SJ;
  if (a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // (launch-uniform)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  LaneAxes<T, 2> la;
  if constexpr (RECT && AXR != 0) la = load_lane_axes<T, 2, AXR>(a.ax);
  else if (RECT && a.ax.use_lds) stage_axes<T, 2>(a.ax, smem_raw);
  const unsigned char* axis_base = (RECT && AXR == 0 && a.ax.use_lds) ? smem_raw : a.ax.image;
  const unsigned lane = threadIdx.x;
  const unsigned q = lane & 1;  // which row piece this lane fetches
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  const size_t nslots = (a.npts + PPL - 1) / PPL;
  const size_t niter = (nslots + nthreads - 1) / nthreads;
  for (size_t it = 0; it < niter; ++it) {
    // every lane runs every iteration (its pair partner may be live)
    const size_t i0 = (it * nthreads + (size_t)blockIdx.x * kBlock + lane) * PPL;
    bool live[PPL];
#pragma unroll
    for (int h = 0; h < PPL; ++h) live[h] = i0 + h < a.npts;
    T xin[PPL][2];
    if (PPL == 2) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
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
      for (int d = 0; d < 2; ++d) xin[0][d] = live[0] ? stream_load(a.obs[d] + i0) : (RECT ? (T)0 : a.start[d]);
    }
    int cell[PPL][2];
    T x0r[PPL][2], x1r[PPL][2];
    if constexpr (RECT && AXR != 0) lane_axes_locate<T, 2, PPL, AXR>(a.ax, la, xin, cell, x0r, x1r);
    T resv[PPL];
#pragma unroll
    for (int h = 0; h < PPL; ++h) {
      T t[2];
      int loc[2];
      bool ok = true;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const T x = xin[h][d];
        if (RECT) {
          T x0, x1;
          int l;
          if constexpr (AXR != 0) {
            l = cell[h][d];
            x0 = x0r[h][d];
            x1 = x1r[h][d];
          } else {
            const Axis<T> ax = make_axis<T, 2>(a.ax, axis_base, d);
            l = axis_cell<T>(ax, x, &x0, &x1);  // multilinear/rectilinear.rs:353-370, :310-311
          }
          const T step = x1 - x0;
          t[d] = (x - x0) / step;
          loc[d] = l;
        } else {
          T floc;
          ok &= regular_floc<T>(x, a.start[d], a.step[d], &floc);  // multilinear/regular.rs:415-418
          const int l = clamp_loc<T>(floc, a.n[d] - 2);
          const T izl = mul_add<FMA>(a.step[d], (T)l, a.start[d]);
          t[d] = (x - izl) / a.step[d];
          loc[d] = l;
        }
      }
      if (!RECT && !ok && live[h]) atomicMin(a.first_bad, (unsigned long long)(i0 + h));
      // brick (bi = i, bj = j / SJ); my point's pair starts at column j - bj*SJ of both rows
      const unsigned bj = (unsigned)loc[1] / (unsigned)SJ;
      const unsigned mine = ((unsigned)loc[0] * a.nbj + bj) * (unsigned)Brick2Geom<T>::ELEMS + ((unsigned)loc[1] - bj * (unsigned)SJ);
      const unsigned theirs = dpp_swap1(mine);
      // instruction r fetches point r of the pair (r = 0: even lane's point, r = 1: odd lane's)
      const unsigned off0 = (q == 0 ? mine : theirs) + q * (unsigned)KW;
      const unsigned off1 = (q == 0 ? theirs : mine) + q * (unsigned)KW;
      const P p0 = *reinterpret_cast<const P*>(a.bricks + off0);
      const P p1 = *reinterpret_cast<const P*>(a.bricks + off1);
      // I keep the piece of my own point and trade the other one
      const P keep = q == 0 ? p0 : p1;
      const P send = q == 0 ? p1 : p0;
      P recv;
      recv.x = dpp_swap1(send.x);
      recv.y = dpp_swap1(send.y);
      const P row0 = q == 0 ? keep : recv;  // row i   : v(i, j), v(i, j+1)
      const P row1 = q == 0 ? recv : keep;  // row i+1
      // reference tree: dim 0 first for each j, then dim 1 (multilinear/regular.rs:347-403)
      const T c0 = mul_add<FMA>(t[0], row1.x - row0.x, row0.x);
      const T c1 = mul_add<FMA>(t[0], row1.y - row0.y, row0.y);
      resv[h] = mul_add<FMA>(t[1], c1 - c0, c0);
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

// ---- Sweep evaluation of large 2-D batches (round 5, third session) ---------------------------------------------
// linear_sweep.h's scheme (read its head: points ordered on chip by leading cell index inside the window the chip
// holds, all waves walking that index in step with a clock, rounds dealt on demand) around this file's rows: a 2-D
// table of more than a few MiB (512^2: 4.7 MiB, 1000^2: 18 MiB) is missed by unordered points once per point; ordered,
// an XCD fetches every brick row once per window.  Regular grids; cell index and t without divide sequences
// (interpn_device.h::step_cell_fast).  A point is 16 bytes of coordinates: 16 rows per wave in registers + 4 parked.
template <typename T>
struct Linear2SweepArgs {
  SweepRounds<T, 2> r;   // sweep_rounds.h: the streams, the sort key, the rounds
  const T* bricks;
  unsigned long long* first_bad;
  T start[2], step[2], rstep[2];  // rstep: RN(1 / step), a division in T on the host (interpn_device.h::step_cell_fast)
  int n[2];
  unsigned nbj;
  unsigned fastdiv;
};

template <typename T, bool FMA, int K, int KL, int THREADS>
__global__ void __launch_bounds__(THREADS) k_linear2_sweep(const Linear2SweepArgs<T> s) {
  typedef typename LeafVec<T, 2>::type P;
  typedef SweepRoundsLds<T, 2, K, KL> L;
  constexpr int KW = Brick2Geom<T>::KW;
  constexpr int SJ = Brick2Geom<T>::SJ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const unsigned q = threadIdx.x & 1u;  // which row piece this lane fetches
  sweep_rounds<T, 2, K, KL, THREADS, 0, L>(s.r, smem_raw, [&](auto, const T (&xr)[2], size_t gi) -> T {
    T t[2];
    int loc[2];
    bool ok = true;
    bool exact = s.fastdiv != 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const StepCell<T> sc = step_cell_fast<FMA>(xr[d], s.start[d], s.step[d], s.rstep[d], s.n[d] - 2);
      t[d] = sc.t;
      loc[d] = sc.loc;
      exact = exact && sc.exact;
    }
    if (__any(!exact)) {  // a lane on a grid line, far outside, not finite ...: the reference's operations as they stand, for the wave
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        T floc;
        ok &= regular_floc<T>(xr[d], s.start[d], s.step[d], &floc);  // multilinear/regular.rs:415-418
        const int l = clamp_loc<T>(floc, s.n[d] - 2);
        const T izl = mul_add<FMA>(s.step[d], (T)l, s.start[d]);
        t[d] = (xr[d] - izl) / s.step[d];
        loc[d] = l;
      }
    }
    if (!ok && gi < s.r.npts) atomicMin(s.first_bad, (unsigned long long)gi);
    // the rows of k_linear2_brick: brick (bi = i, bj = j / SJ), lane pairs fetch the two row pieces of one point per load
    const unsigned bj = (unsigned)loc[1] / (unsigned)SJ;
    const unsigned mine_e = ((unsigned)loc[0] * s.nbj + bj) * (unsigned)Brick2Geom<T>::ELEMS + ((unsigned)loc[1] - bj * (unsigned)SJ);
    const unsigned theirs = dpp_swap1(mine_e);
    const unsigned off0 = (q == 0 ? mine_e : theirs) + q * (unsigned)KW;
    const unsigned off1 = (q == 0 ? theirs : mine_e) + q * (unsigned)KW;
    const P p0 = *reinterpret_cast<const P*>(s.bricks + off0);
    const P p1 = *reinterpret_cast<const P*>(s.bricks + off1);
    const P keep = q == 0 ? p0 : p1;
    const P send = q == 0 ? p1 : p0;
    P recv;
    recv.x = dpp_swap1(send.x);
    recv.y = dpp_swap1(send.y);
    const P row0 = q == 0 ? keep : recv;
    const P row1 = q == 0 ? recv : keep;
    const T c0 = mul_add<FMA>(t[0], row1.x - row0.x, row0.x);  // multilinear/regular.rs:347-403
    const T c1 = mul_add<FMA>(t[0], row1.y - row0.y, row0.y);
    return mul_add<FMA>(t[1], c1 - c0, c0);
  });
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_bricks2(const T* __restrict__ vals, T* __restrict__ bricks, int n0, int n1,
                                                          unsigned nbi, unsigned nbj) {
  constexpr int KW = Brick2Geom<T>::KW;
  constexpr int EL = Brick2Geom<T>::ELEMS;
  const size_t total = (size_t)nbi * nbj * EL;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const unsigned within = (unsigned)(e % EL);
    const size_t b = e / EL;
    const unsigned bj = (unsigned)(b % nbj);
    const unsigned bi = (unsigned)(b / nbj);
    const int i = (int)bi + (int)(within / KW);
    const int j = (int)bj * (KW - 1) + (int)(within % KW);
    T v = (T)0;
    if (i < n0 && j < n1) v = vals[(size_t)i * n1 + j];
    bricks[e] = v;
  }
}

void brick2_geometry(const GridDesc& g, unsigned nb[2], size_t* bytes) {
  const int kw = g.dtype == kF64 ? 8 : 16;
  nb[0] = (unsigned)(g.n[0] - 1);
  nb[1] = (unsigned)((g.n[1] - 2) / (kw - 1) + 1);
  *bytes = (size_t)nb[0] * nb[1] * 128;
}

hipError_t build_bricks2(const GridDesc& g, void* bricks, hipStream_t stream) {
  const size_t elems = (size_t)g.brick_nb[0] * g.brick_nb[1] * (g.dtype == kF64 ? 16 : 32);
  size_t blocks = (elems + kBlock - 1) / kBlock;
  if (blocks > 65535) blocks = 65535;
  if (g.dtype == kF64)
    hipLaunchKernelGGL(k_build_bricks2<double>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, static_cast<const double*>(g.vals),
                       static_cast<double*>(bricks), g.n[0], g.n[1], g.brick_nb[0], g.brick_nb[1]);
  else
    hipLaunchKernelGGL(k_build_bricks2<float>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, static_cast<const float*>(g.vals),
                       static_cast<float*>(bricks), g.n[0], g.n[1], g.brick_nb[0], g.brick_nb[1]);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_linear2_brick(const GridDesc& g, const T* const* obs, T* out, size_t npts, unsigned long long* first_bad,
                                hipStream_t stream) {
  Brick2Args<T> a;
  a.bricks = static_cast<const T*>(g.bricks);
  a.out = out;
  a.first_bad = first_bad;
  a.npts = npts;
  for (int d = 0; d < 2; ++d) {
    a.obs[d] = obs[d];
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
  }
  a.nbj = g.brick_nb[1];
  size_t lds = 0;
  a.ax.use_lds = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  if (g.kind == kRectilinear) lds = fill_axis_args<T, 2>(g, a.ax, /*big_lds=*/true, /*records=*/true);
  const int axr = lane_axes_mode(g);  // both axes <= 64 coordinates: searched across lanes, no LDS image
  if (axr) lds = 0;
  // Two points per lane when every stream is aligned to 2*sizeof(T) (the handle's `ppl` option = 1
  // forces the scalar form).
  bool aligned = (reinterpret_cast<uintptr_t>(out) % (2 * sizeof(T))) == 0;
  for (int d = 0; d < 2; ++d) aligned = aligned && (reinterpret_cast<uintptr_t>(obs[d]) % (2 * sizeof(T))) == 0;
  const int ppl = (aligned && g.cfg.ppl != 1) ? 2 : 1;
  unsigned blocks = (g.kind == kRegular || axr) ? one_pass_blocks(npts, ppl * (axr ? 4 : 1)) : grid_blocks(npts, ppl, g.cfg);
  a.gate = g.launch_gate;
  if ((a.gate || g.launch_fat) && g.cfg.gated_iters > 1) blocks = (blocks + (unsigned)g.cfg.gated_iters - 1) / (unsigned)g.cfg.gated_iters;  // few, fat workgroups: mostly they return at once
#define GO2(RECT, FMA, AXR, PPL) do { g.tag.set("k_linear2_brick", {RECT, FMA, AXR, PPL}, 0b0011u); hipLaunchKernelGGL((k_linear2_brick<T, RECT, FMA, AXR, PPL>), dim3(blocks), dim3(kBlock), lds, stream, a); } while (0)
#define GO(RECT, FMA, AXR) do { if (ppl == 2) GO2(RECT, FMA, AXR, 2); else GO2(RECT, FMA, AXR, 1); } while (0)
  if (g.kind == kRegular) { if (g.fma) GO(false, true, 0); else GO(false, false, 0); }
  else if (axr == 2) { if (g.fma) GO(true, true, 2); else GO(true, false, 2); }
  else if (axr == 3) { if (g.fma) GO(true, true, 3); else GO(true, false, 3); }
  else if (axr == 1) { if (g.fma) GO(true, true, 1); else GO(true, false, 1); }
  else { if (g.fma) GO(true, true, 0); else GO(true, false, 0); }
#undef GO
#undef GO2
  return hipGetLastError();
}

template hipError_t launch_linear2_brick<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t);
template hipError_t launch_linear2_brick<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t);

// ---- host side of the 2-D sweep evaluation ------------------------------------------------------------------------
namespace {
constexpr int kSweep2Threads = 768;
template <typename T> constexpr int sweep2_rows() { return sizeof(T) == 8 ? 16 : 28; }  // (f32: 32 rows in registers spill 21 in the fma flavour)
template <typename T> constexpr int sweep2_parked() { return 4; }
}  // namespace

// 0 = never for this handle, 1 = not for this batch, 2 = yes.
int linear2_sweep_applies(const GridDesc& g, size_t npts) {
  if (g.method != kLinear || g.ndims != 2 || g.kind != kRegular || !g.bricks || g.cfg.sweep == 0 || g.cfg.force_generic) return 0;
  unsigned nb[2];
  size_t bytes = 0;
  brick2_geometry(g, nb, &bytes);
  if (bytes >= (1ull << 32) / 2) return 0;  // 32-bit element offsets
  const size_t lds = (size_t)SweepRoundsLds<double, 2, 16, 4>::kWave * (kSweep2Threads / 64) + SweepRoundsLds<double, 2, 16, 4>::kWorkgroup;
  if ((long long)lds > g.cfg.lds_per_cu) return 0;
  if (g.cfg.sweep > 0) return 2;
  // automatic (profiles/r05_linear2_sweep.jsonl, 1e8 points): f64 1000^2 1.52 -> 0.87 ms, 2000^2 1.93 -> 1.52, and on tables the
  // L2 holds 512^2 0.80 -> 0.68, 64^2 0.73 -> 0.56 (3e7 points: 0.22 -> 0.21); f32 (28 rows in registers + 4 parked: 32 in registers
  // spilled and gained nothing) 512^2 0.58 -> 0.48, 1000^2 0.96 -> 0.57, 2000^2 1.59 -> 0.80.  Crossover: 1000^2 below 1e7 points (0.131 against 0.161 ms there), L2-resident
  // tables at 2e7 (64^2 .. 512^2: 0.96-0.99 there, 0.84-0.90 at 4e7): from 2.5 rounds per wave, six where the L2 holds the table.
  const bool beyond_l2 = bytes > thresholds(g.cfg).table_l2_sized;
  const size_t cus = (size_t)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
  const size_t per_round = (size_t)(g.dtype == kF64 ? 20 : 32) * kSweep2Threads * cus;
  if (npts < (beyond_l2 ? per_round * 5 / 2 : per_round * 6)) return 1;
  return 2;
}

template <typename T>
static hipError_t launch2_t(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad, void* work, hipStream_t stream) {
  constexpr int K = sweep2_rows<T>(), KL = sweep2_parked<T>(), TH = kSweep2Threads;
  Linear2SweepArgs<T> s;
  SweepRounds<T, 2>& r = s.r;
  s.bricks = static_cast<const T*>(g.bricks);
  r.out = static_cast<T*>(out);
  s.first_bad = first_bad;
  r.npts = npts;
  s.fastdiv = 1;
  for (int d = 0; d < 2; ++d) {
    r.obs[d] = static_cast<const T*>(obs[d]);
    s.start[d] = (T)g.start[d];
    r.absent[d] = s.start[d];
    s.step[d] = (T)g.step[d];
    s.n[d] = g.n[d];
    const volatile T one = (T)1;
    s.rstep[d] = one / (T)g.step[d];
    const double mag = g.step[d] < 0 ? -g.step[d] : g.step[d];
    if (!(mag >= StepCellRange<T>::lo && mag <= StepCellRange<T>::hi)) s.fastdiv = 0;
  }
  s.nbj = g.brick_nb[1];
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
  r.period_default = 2500;
  r.gated = g.sweep_gated ? 1u : 0u;
  r.stamps = nullptr;
  r.work = static_cast<SweepWork*>(work);
  const unsigned cus = (unsigned)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
  unsigned blocks = cus;
  const unsigned need = (r.rounds + (TH / 64) - 1) / (TH / 64);
  if (blocks > need) blocks = need;
  const size_t lds = (size_t)SweepRoundsLds<T, 2, K, KL>::kWave * (TH / 64) + SweepRoundsLds<T, 2, K, KL>::kWorkgroup;
  auto launch = [&](auto kern, bool fma) -> hipError_t {
    if (lds > 64 * 1024) {  // (per launch: the two flavours are two kernels, and the call is cheap beside a batch of this size)
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    g.tag.set("k_linear2_sweep", {fma, K, KL, TH}, 0b0001u);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(TH), lds, stream, s);
    return hipGetLastError();
  };
  return g.fma ? launch(k_linear2_sweep<T, true, K, KL, TH>, true) : launch(k_linear2_sweep<T, false, K, KL, TH>, false);
}

hipError_t launch_linear2_sweep(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad, void* work, hipStream_t stream) {
  if (g.method != kLinear || g.ndims != 2 || !g.bricks || !work || npts == 0) return hipErrorInvalidValue;
  for (int d = 0; d < 2; ++d)
    if (reinterpret_cast<uintptr_t>(obs[d]) % 16) return hipErrorInvalidValue;
  if (reinterpret_cast<uintptr_t>(out) % 16) return hipErrorInvalidValue;
  if (g.dtype == kF64) return launch2_t<double>(g, obs, out, npts, first_bad, work, stream);
  return launch2_t<float>(g, obs, out, npts, first_bad, work, stream);
}

}  // namespace interpn
