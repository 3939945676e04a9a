// Host side of the sweep evaluation of 2-D and 3-D multicubic batches (cubic_sweep.h): when it applies and the launcher.
#include <atomic>

#include "cubic_sweep.h"

namespace interpn {

namespace {

constexpr int kCubicSweepThreads = 768;  // one workgroup per CU, three waves per SIMD
// rows of 64 points per wave and round: in registers + parked in LDS (cubic_sweep.h K, KL)
template <typename T, bool RECT, int N = 3> constexpr int cubic_sweep_rows() { return N == 2 ? (sizeof(T) == 8 ? 10 : 24) : (sizeof(T) == 8 ? 6 : 16); }
template <typename T, bool RECT, int N = 3> constexpr int cubic_sweep_parked() { return N == 2 ? 4 : (sizeof(T) == 8 ? 2 : 4); }
// threads per workgroup: rectilinear f64 rows (three CubicDimRect + the tile: ~130 registers beside the rows' coordinates) get
// the 256 registers of two waves per SIMD
template <typename T, bool RECT, int N = 3> constexpr int cubic_sweep_threads() { return RECT && sizeof(T) == 8 ? 512 : kCubicSweepThreads; }
// LDS the axis image of a rectilinear grid may take beside the waves' regions
constexpr size_t kCubicSweepAxisLds = 16 * 1024;

// the fully overlapped tile table of a 3-D multicubic handle (steps 1,1): `bricks` itself or the second table
const void* tiles11(const GridDesc& g, unsigned nb[2]) {
  if (g.bricks && g.brick_step[0] == 1 && g.brick_step[1] == 1) { nb[0] = g.brick_nb[0]; nb[1] = g.brick_nb[1]; return g.bricks; }
  if (g.bricks11) { nb[0] = g.bricks11_nb[0]; nb[1] = g.bricks11_nb[1]; return g.bricks11; }
  return nullptr;
}

// LDS of the waves' regions of the shape a handle's batches run with
size_t cubic_sweep_wave_lds(const GridDesc& g) {
  const unsigned waves = kCubicSweepThreads / 64;
  if (g.ndims == 2)
    return g.dtype == kF64 ? (size_t)CubicSweepLds<double, cubic_sweep_rows<double, false, 2>(), cubic_sweep_parked<double, false, 2>(), 2>::kWave * waves + 16
                           : (size_t)CubicSweepLds<float, cubic_sweep_rows<float, false, 2>(), cubic_sweep_parked<float, false, 2>(), 2>::kWave * waves + 16;
  return g.dtype == kF64 ? (size_t)CubicSweepLds<double, cubic_sweep_rows<double, false, 3>(), cubic_sweep_parked<double, false, 3>(), 3>::kWave * waves + 16
                         : (size_t)CubicSweepLds<float, cubic_sweep_rows<float, false, 3>(), cubic_sweep_parked<float, false, 3>(), 3>::kWave * waves + 16;
}

}  // namespace

// 0 = never for this handle, 1 = not for this batch, 2 = yes.
int cubic_sweep_applies(const GridDesc& g, size_t npts) {
  if (g.method != kCubic || (g.ndims != 3 && g.ndims != 2) || g.cfg.sweep == 0 || g.cfg.force_generic) return 0;
  unsigned nb[2];
  if (!tiles11(g, nb)) return 0;
  if ((long long)cubic_sweep_wave_lds(g) > g.cfg.lds_per_cu) return 0;  // (a rectilinear grid's axes: in LDS where room is left, else read through the caches)
  if (g.cfg.sweep > 0) return 2;
  // automatic (profiles/r05_cubic_sweep.jsonl, 1e7 points): regular grids whose one-tile-per-footprint table unordered points
  // miss — beyond the L2 (32^3: 0.38 ms either way) and not so large that a window no longer re-uses its planes while the
  // tiled kernel has a smaller layout to fall back on (f32 96^3: 53 MiB, 0.53 against 0.45 ms on its 4 x 4 tiles; f64 96^3,
  // 106 MiB: 0.64 against 0.75) —, batches from two rounds per wave (f64 64^3: 0.207 against 0.249 ms at 4e6 points, 0.142
  // against 0.129 at 2e6; f32: 0.187 against 0.250 at 6e6, 0.168 against 0.177 at 4e6).  Rectilinear grids: on request only
  // (their rows are bound by the cell search and the nodes' divisions: 0.97 against 0.95 ms at 64^3).
  if (g.kind == kRectilinear) return 1;
  if (g.ndims == 2) {  // 2-D (profiles/r05_cubic_sweep.jsonl): f64 128^2 .. 1000^2 8-20 % ahead at 3e7 points, behind at 1e7; f32: no gain
    if (g.dtype != kF64) return 1;
    const size_t cus2 = (size_t)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
    return npts >= (size_t)8 * 14 * kCubicSweepThreads * cus2 ? 2 : 1;
  }
  unsigned nbb[2];
  size_t bytes = 0;
  cubic_tile_geometry(g, 1, 1, nbb, &bytes);
  if (bytes <= thresholds(g.cfg).table_l2_sized) return 0;
  if (bytes > (g.dtype == kF64 ? (size_t)128 << 20 : (size_t)32 << 20)) return 0;
  const size_t cus = (size_t)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
  const size_t per_round = (size_t)(g.ndims == 2 ? (g.dtype == kF64 ? 14 : 28) : (g.dtype == kF64 ? 8 : 20)) * kCubicSweepThreads * cus;
  if (npts < (g.dtype == kF64 ? 2 * per_round : per_round + per_round / 2)) return 1;
  return 2;
}

template <typename T, int N, bool RECT, bool FMA, int K = cubic_sweep_rows<T, RECT, N>(), int KL = cubic_sweep_parked<T, RECT, N>(), int TH = cubic_sweep_threads<T, RECT, N>()>
static hipError_t go(const GridDesc& g, CubicSweepArgs<T, N> s, unsigned cus, hipStream_t stream) {
  {
    const size_t chunk = (size_t)64 * (K + KL);
    const size_t rounds = (s.r.npts + chunk - 1) / chunk;
    if (rounds > 0xFFFFFFF0ull) return hipErrorInvalidValue;
    s.r.rounds = (unsigned)rounds;
    s.r.per_shard = (s.r.rounds + 7u) / 8u;
  }
  unsigned blocks = cus;
  {
    const unsigned need = (s.r.rounds + (TH / 64) - 1) / (TH / 64);
    if (blocks > need) blocks = need;
  }
  auto kern = k_cubic_sweep<T, RECT, FMA, K, KL, TH, N>;
  const size_t lds = (size_t)CubicSweepLds<T, K, KL, N>::kWave * (TH / 64) + CubicSweepLds<T, K, KL, N>::kWorkgroup + ((RECT && s.c.ax.use_lds) ? (size_t)s.c.ax.image_bytes : 0);
  static std::atomic<unsigned long long> opted{0};  // bit per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
  if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || !((opted.load() >> dev) & 1ull))) {
    size_t most = (size_t)CubicSweepLds<T, K, KL, N>::kWave * (TH / 64) + CubicSweepLds<T, K, KL, N>::kWorkgroup + (RECT ? kCubicSweepAxisLds : 0);
    if (g.cfg.lds_per_cu > 0 && most > (size_t)g.cfg.lds_per_cu) most = (size_t)g.cfg.lds_per_cu;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) opted.fetch_or(1ull << dev);
  }
  if (N == 3) g.tag.set("k_cubic_sweep", {RECT, FMA, K, KL, TH}, 0b00011u);
  else g.tag.set("k_cubic_sweep", {RECT, FMA, K, KL, TH, N}, 0b000011u);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(TH), lds, stream, s);
  return hipGetLastError();
}

template <typename T, int N>
static hipError_t launch_t(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad,
                           void* work, hipStream_t stream) {
  constexpr int KD = N == 3 ? 2 : 0;  // the dimension the points are ordered by: the tile table's slowest index
  CubicSweepArgs<T, N> s;
  CubicBrickArgs<T, N>& a = s.c;
  unsigned nb[2];
  a.bricks = static_cast<const T*>(tiles11(g, nb));
  if (!a.bricks) return hipErrorInvalidValue;
  {
    unsigned nbb[2];
    size_t bytes = 0;
    cubic_tile_geometry(g, 1, 1, nbb, &bytes);
    if (bytes >= 0xFFFFF000ull) return hipErrorInvalidValue;
    a.table_bytes = (unsigned)bytes;
  }
  SweepRounds<T, N>& r = s.r;
  a.out = static_cast<T*>(out);
  a.first_bad = first_bad;
  a.npts = npts;
  r.out = a.out;
  r.npts = npts;
  a.scatter = nullptr;
  a.gate = nullptr;
  a.index_base = 0;
  a.eighth = 0;
  a.linearize = g.linearize;
  for (int d = 0; d < N; ++d) {
    a.obs[d] = static_cast<const T*>(obs[d]);
    r.obs[d] = a.obs[d];
    r.absent[d] = g.kind == kRectilinear ? (T)0 : (T)g.start[d];
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
    a.plane_stride[d] = 0;
  }
  a.nbj = nb[1];
  if constexpr (N == 3) a.plane_stride[2] = nb[0] * nb[1] * 16u;  // table[plane (dim 2)][bi][bj][16]
  s.fastdiv = g.kind == kRectilinear ? 0u : 1u;
  for (int d = 0; d < N; ++d) {
    const volatile T one = (T)1;  // one IEEE division in T, at run time
    s.rstep[d] = g.kind == kRectilinear ? (T)0 : one / (T)g.step[d];
    const double mag = g.step[d] < 0 ? -g.step[d] : g.step[d];
    if (!(mag >= StepCellRange<T>::lo && mag <= StepCellRange<T>::hi)) s.fastdiv = 0;
  }
  a.ax.use_lds = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  for (int d = 0; d < N; ++d)
    a.crec[d] = (g.kind == kRectilinear && g.axis_crec_bytes)
                    ? reinterpret_cast<const CubicCellRecord<T>*>(static_cast<const unsigned char*>(g.axis_image) + g.axis_crec_off[d]) : nullptr;
  if (g.kind == kRectilinear) {
    (void)fill_axis_args<T, N>(g, a.ax);
    if (a.ax.use_lds && (a.ax.image_bytes > kCubicSweepAxisLds || (long long)(cubic_sweep_wave_lds(g) + a.ax.image_bytes) > g.cfg.lds_per_cu)) a.ax.use_lds = 0;
    const double span = g.bound_hi[KD] - g.bound_lo[KD];
    r.key_start = (T)g.bound_lo[KD];
    r.key_scale = span > 0 ? (T)((double)(g.n[KD] - 1) / span) : (T)0;
  } else {
    r.key_start = (T)g.start[KD];
    r.key_scale = (T)(1.0 / g.step[KD]);
  }
  if (!(r.key_scale > 0) || !(r.key_scale < (T)1e30)) r.key_scale = 0;  // every point in bin 0: still correct
  r.key_cells = g.n[KD] - 2;
  r.key_shift = 0;
  while (((g.n[KD] - 2) >> r.key_shift) >= 64) ++r.key_shift;
  r.period = g.cfg.sweep_period > 0 ? (unsigned)g.cfg.sweep_period : 0u;
  r.period_default = 4000;  // 40 us: a round of ten cubic rows (the kernel measures from its first launch on)
  r.gated = g.sweep_gated ? 1u : 0u;
  r.stamps = nullptr;
  r.work = static_cast<SweepWork*>(work);
  const unsigned cus = (unsigned)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
  if (g.kind == kRegular) return g.fma ? go<T, N, false, true>(g, s, cus, stream) : go<T, N, false, false>(g, s, cus, stream);
  return g.fma ? go<T, N, true, true>(g, s, cus, stream) : go<T, N, true, false>(g, s, cus, stream);
}

hipError_t launch_cubic_sweep(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad,
                              void* work, hipStream_t stream) {
  if (g.method != kCubic || (g.ndims != 3 && g.ndims != 2) || !work || npts == 0) return hipErrorInvalidValue;
  for (int d = 0; d < g.ndims; ++d)
    if (reinterpret_cast<uintptr_t>(obs[d]) % 16) return hipErrorInvalidValue;  // the caller checked (abi_sweep.hip)
  if (reinterpret_cast<uintptr_t>(out) % 16) return hipErrorInvalidValue;
  if (g.ndims == 2) return g.dtype == kF64 ? launch_t<double, 2>(g, obs, out, npts, first_bad, work, stream) : launch_t<float, 2>(g, obs, out, npts, first_bad, work, stream);
  return g.dtype == kF64 ? launch_t<double, 3>(g, obs, out, npts, first_bad, work, stream) : launch_t<float, 3>(g, obs, out, npts, first_bad, work, stream);
}

}  // namespace interpn
