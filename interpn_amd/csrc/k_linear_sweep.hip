// Host side of the sweep evaluation of 3-D multilinear batches (linear_sweep.h): when it applies,
// which table it runs on, and the launcher.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "linear_sweep.h"

namespace interpn {

namespace {

constexpr int kSweepRows = 12;       // f64: rows of 64 points per wave and round in registers: three waves per SIMD at 168 VGPRs ...
constexpr int kSweepParked = 4;      // ... + rows parked in LDS between the sort and their turn (linear_sweep.h KL): 156 of a CU's 160 KiB; 64^3: 0.925 (two rows) -> 0.90 ms
#ifndef INTERPN_SWEEP_RECT_PARKED
#define INTERPN_SWEEP_RECT_PARKED 2
#endif
#ifndef INTERPN_SWEEP_RECT_ROWS
#define INTERPN_SWEEP_RECT_ROWS 12
#endif
constexpr int kSweepParkedRect = INTERPN_SWEEP_RECT_PARKED;  // rectilinear grids, axes in lanes: the cell search's registers leave room for two (four spill 7)
#ifndef INTERPN_SWEEP_F32_ROWS
#define INTERPN_SWEEP_F32_ROWS 24
#endif
#ifndef INTERPN_SWEEP_F32_PARKED
#define INTERPN_SWEEP_F32_PARKED 0
#endif
constexpr int kSweepRowsF32 = INTERPN_SWEEP_F32_ROWS;    // f32: half the registers per point
constexpr int kSweepThreads = 768;   // one workgroup per CU
constexpr int kSweepRowsF32Rect = 20;  // ... less the registers of the cell search (24 rows spill 6-29 VGPRs there)
#ifndef INTERPN_SWEEP_LANES_ROWS   // f64 rectilinear with the axes in lanes (AXR 1..3): rows in registers + parked (the same 14 rows per round)
#define INTERPN_SWEEP_LANES_ROWS INTERPN_SWEEP_RECT_ROWS
#define INTERPN_SWEEP_LANES_PARKED INTERPN_SWEEP_RECT_PARKED
#endif
static_assert(INTERPN_SWEEP_LANES_ROWS + INTERPN_SWEEP_LANES_PARKED == INTERPN_SWEEP_RECT_ROWS + INTERPN_SWEEP_RECT_PARKED, "one round size per grid kind");
template <typename T, bool RECT = false, int AXR = 0> constexpr int sweep_rows() { return sizeof(T) == 8 ? (RECT ? (AXR >= 1 && AXR <= 3 ? INTERPN_SWEEP_LANES_ROWS : INTERPN_SWEEP_RECT_ROWS) : kSweepRows) : (RECT ? kSweepRowsF32Rect : kSweepRowsF32); }
template <typename T, bool RECT = false, int AXR = 0> constexpr int sweep_parked() { return sizeof(T) == 8 ? (RECT ? (AXR >= 1 && AXR <= 3 ? INTERPN_SWEEP_LANES_PARKED : kSweepParkedRect) : kSweepParked) : (RECT ? 0 : INTERPN_SWEEP_F32_PARKED); }
// points the chip holds at a time, per CU (the sweep's window, linear_sweep.h)
constexpr size_t kSweepPointsPerCu = (size_t)(kSweepRows + kSweepParked) * kSweepThreads;
constexpr size_t kSweepPointsPerCuF32 = (size_t)(kSweepRowsF32 + INTERPN_SWEEP_F32_PARKED) * kSweepThreads;
// LDS the axis image of a rectilinear grid may take beside the waves' regions (92 KiB of a CU's 160): what
// fill_axis_args allows the brick kernels (20 KiB) — per-bucket records of up to ~500 coordinates per axis
constexpr size_t kSweepAxisLds = 20 * 1024;

size_t brick_lines(const GridDesc& g, int si, int sj) {
  unsigned nb[3];
  size_t bytes = 0;
  brick_geometry(g, si, sj, nb, &bytes);
  return bytes / 128;
}

}  // namespace

size_t sweep_work_bytes() { return sizeof(SweepWork); }

// Which brick layout the sweep evaluation of this grid wants: (1,1) while a sweep's window still
// re-uses its lines (table lines fetched per point and XCD = 8 x lines / window <= 1/2), else
// (1,2) — half the lines at 1.5 lines per point — while THAT re-uses them, else (1,1) again (no
// re-use either way: fewest lines per point).  Measured (tools/sweep_clock_probe.py, 1e8 points,
// f64): 64^3 (1,1) 1.07 against (1,2) 1.19 ms; 80^3 (1,2) 1.25 against (1,1) 1.35.
// Returns false where the sweep does not apply to the handle at all.
bool sweep_layout(const GridDesc& g, int* si, int* sj, int* cell) {
  if (g.method != kLinear || g.ndims != 3 || g.cfg.sweep == 0) return false;
  *cell = 0;
  if (g.dtype == kF32) {  // f32: the 2 x 4 x 4 bricks (one line per cell at 3.56x the grid; linear_brick.h CELL == 2)
    *si = 1; *sj = 1; *cell = 2;
    return true;
  }
  if (const char* env = getenv("INTERPN_HIP_SWEEP_LAYOUT")) {  // tuning / tests: "11" or "12"
    if (!strcmp(env, "11") || !strcmp(env, "12")) { *si = 1; *sj = env[1] - '0'; return true; }
  }
  const size_t window = kSweepPointsPerCu * (size_t)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256) * 85 / 100;  // ~ the share of a round spent in rows
  const size_t l11 = brick_lines(g, 1, 1), l12 = brick_lines(g, 1, 2);
  const int xcds = g.cfg.num_xcds > 0 ? g.cfg.num_xcds : 8;
  if ((size_t)xcds * l11 * 2 <= window) { *si = 1; *sj = 1; }
  else if ((size_t)xcds * l12 * 10 <= window * 7) { *si = 1; *sj = 2; }
  else { *si = 1; *sj = 1; }
  return true;
}

// 0 = never for this handle, 1 = not for this batch, 2 = yes.
int sweep_applies(const GridDesc& g, size_t npts) {
  if (g.method == kNearest) return nearest_sweep_applies(g, npts);  // 2-D / 3-D nearest neighbour: k_nearest.hip
  if (g.method == kCubic) return cubic_sweep_applies(g, npts);  // 3-D multicubic: cubic_sweep.h
  if (g.method == kLinear && g.ndims == 2) return linear2_sweep_applies(g, npts);  // 2-D multilinear: k_linear2_brick.hip
  if (!g.sweep_bricks || g.cfg.sweep == 0 || g.cfg.force_generic) return 0;
  if (g.sweep_table_bytes >= (1ull << 32)) return 0;
  // the rows' index arithmetic is 24-bit (linear_brick.h::brick_line_bytes24, div_small): axes shorter than 2^20, fewer than
  // 2^24 (i, j) bricks and bricks along k — true of every table under 4 GiB that is not a needle
  for (int d = 0; d < 3; ++d)
    if (g.n[d] >= (1 << 20)) return 0;
  if ((unsigned long long)g.sweep_nb[0] * g.sweep_nb[1] >= (1ull << 24) || g.sweep_nb[2] >= (1u << 24)) return 0;  // the kernel addresses the table with 32-bit byte offsets (and a table that size is re-used by nobody)
  // the workgroup's LDS (its waves' regions, + the axis image budget on rectilinear grids) must exist on this device
  const size_t lds = g.kind == kRectilinear
                         ? (lane_axes_mode(g) != 0  // axes in lanes: no axis image in LDS
                                ? (size_t)SweepLds<double, INTERPN_SWEEP_LANES_ROWS, INTERPN_SWEEP_LANES_PARKED>::kWave * (kSweepThreads / 64) + SweepLds<double, INTERPN_SWEEP_LANES_ROWS, INTERPN_SWEEP_LANES_PARKED>::kWorkgroup
                                : (size_t)SweepLds<double, INTERPN_SWEEP_RECT_ROWS, kSweepParkedRect>::kWave * (kSweepThreads / 64) + SweepLds<double, INTERPN_SWEEP_RECT_ROWS, kSweepParkedRect>::kWorkgroup + kSweepAxisLds)
                         : (size_t)SweepLds<double, kSweepRows, kSweepParked>::kWave * (kSweepThreads / 64) + SweepLds<double, kSweepRows, kSweepParked>::kWorkgroup;  // (the f32 shapes need no more)
  if ((long long)lds > g.cfg.lds_per_cu) return 0;
  if (g.cfg.sweep > 0) return 2;
  // automatic: a batch must give every wave a few rounds (the period is a round's duration; the launch's start and end
  // cost ~35 us more than the brick kernel's).  Measured crossover (profiles/r05_sweep_threshold.jsonl, third
  // session: the division-free kernel): f64 64^3 at 8e6 points, 80^3 at 6e6, 128^3 at 1.3e7; four rounds per
  // wave = 1.26e7 points (regular f64) is at most 2 % slower there and 6-12 % faster at 64^3 / 80^3.  f32: six rounds.
  // Tables the L2 holds (third session): on regular grids the division-free rows make the sweep kernel the
  // faster one there too for large batches (1e8 points: f64 24^3 .. 48^3 0.77-0.83 against 0.91-0.96 ms, f32 64^3
  // 0.59 against 0.68; 3.2e7 points: 7-10 % ahead; 12^3: equal) — from eight rounds per wave; rectilinear grids too
  // (f64 32^3 / 48^3: 0.92-0.95 against 0.99-1.01 ms per 1e8 points, f32 48^3 / 64^3: 0.65-0.66 against 0.78-0.79).
  const bool beyond_l2 = g.sweep_table_bytes > thresholds(g.cfg).table_l2_sized;
  const size_t cus = (size_t)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
  const size_t per_cu = g.dtype == kF64 ? (size_t)(g.kind == kRectilinear ? INTERPN_SWEEP_RECT_ROWS + kSweepParkedRect : kSweepRows + kSweepParked) * kSweepThreads : kSweepPointsPerCuF32;
  const size_t rounds = g.dtype == kF64 ? (beyond_l2 ? 4 : 8) : (beyond_l2 ? 3 : 6);  // (f32, third session: 80^3 from 9e6 points, 128^3 from 6e6; 64^3 — L2-resident — from 2.4e7)
  if (npts < rounds * per_cu * cus) return 1;
  return 2;
}

template <typename T, bool RECT, bool FMA, int SI, int SJ, int AXR, int CELL = 0>
static hipError_t go(const GridDesc& g, const SweepArgs<T>& s, unsigned blocks, hipStream_t stream) {
  constexpr int K = sweep_rows<T, RECT, AXR>(), KL = sweep_parked<T, RECT, AXR>(), TH = kSweepThreads;
  auto kern = k_linear_sweep<T, RECT, FMA, SI, SJ, K, TH, AXR, false, CELL, KL>;
  const size_t lds = (size_t)SweepLds<T, K, KL>::kWave * (TH / 64) + SweepLds<T, K, KL>::kWorkgroup + ((RECT && AXR == 4 && s.b.ax.use_lds) ? (size_t)s.b.ax.image_bytes : 0);
  static std::atomic<unsigned long long> opted{0};  // bit per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
  if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || !((opted.load() >> dev) & 1ull))) {
    // (the largest this instantiation ever asks for: its waves' regions + the axis image budget)
    const size_t most = (size_t)SweepLds<T, K, KL>::kWave * (TH / 64) + SweepLds<T, K, KL>::kWorkgroup + (RECT && AXR == 4 ? kSweepAxisLds : 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)most);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) opted.fetch_or(1ull << dev);
  }
  g.tag.set("k_linear_sweep", {RECT, FMA, SI, SJ, K, TH, AXR, 0, CELL, KL, 0}, 0b00010000011u);  // (the last: ABL, measurement builds only)
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(TH), lds, stream, s);
  return hipGetLastError();
}

template <typename T, bool RECT, bool FMA, int AXR>
static hipError_t go_layout(const GridDesc& g, const SweepArgs<T>& s, unsigned blocks, hipStream_t stream) {
  if constexpr (sizeof(T) == 4) {
    if (g.sweep_cell == 2) return go<T, RECT, FMA, 1, 1, AXR, 2>(g, s, blocks, stream);
    return hipErrorInvalidValue;
  } else {
    if (g.sweep_cell != 0) return hipErrorInvalidValue;
    if (g.sweep_step[0] == 1 && g.sweep_step[1] == 1) return go<T, RECT, FMA, 1, 1, AXR>(g, s, blocks, stream);
    if (g.sweep_step[0] == 1 && g.sweep_step[1] == 2) return go<T, RECT, FMA, 1, 2, AXR>(g, s, blocks, stream);
    return hipErrorInvalidValue;
  }
}

// ---- which kernel for a large batch: decided on the device from a sample -----------------------------------------------
// The sweep kernels are the faster ones on points in no particular order (they make the locality such points lack); on
// batches that are coherent as they stand — re-gridding onto a finer lattice with the last dimension fastest, clustered
// points — the one-pass kernels are (neighbouring lanes already share lines and they run eight waves per SIMD: 64^3 f64
// multilinear, 1e8 points: lattice 0.72 against 0.88 ms, one cell 0.66 against 0.80; nearest 128^3 0.54 against 0.69, 1000^2
// 0.41 against 0.56; 2-D multilinear 0.52 against 0.57; 2-D multicubic 0.34 against 0.38; profiles/r06_obs_*.jsonl).  The
// host cannot look at the points without a synchronisation, so an automatic launch is three: this kernel samples kProbeRows
// rows of 64 consecutive points spread over the batch, counts how often a point's table line differs from its
// predecessor's, and leaves the verdict in the scratch block; the sweep kernel and the one-pass kernel behind it are both
// enqueued, and the one the verdict is against returns at once (+0.8 % on unordered points).  Unordered points change
// line at every step (63 of 63), a fine lattice at a few per row.
constexpr unsigned kProbeRows = 256;
template <typename T, int N>
struct ProbeArgs {
  const T* obs[N];
  size_t npts;
  T start[N], scale[N];  // ~ cell index = (x - start) * scale (a hint, like the sweep's sort key)
  int top[N];            // n - 2
  int sk;                // cells along the last dimension that share a line of the one-pass kernel's table (at least)
  SweepWork* work;
  // the verdict for the HOST as well (abi_sweep.hip: it thins the sampling out once the batches of a handle keep coming out
  // unordered): (seq << 1 | coherent) into a pinned word, or null
  unsigned* host_word;
  unsigned seq;
};

template <typename T, int N>
__global__ void __launch_bounds__(256) k_sweep_probe(const ProbeArgs<T, N> p) {
  const unsigned lane = threadIdx.x & 63u;
  const unsigned row = blockIdx.x * 4u + (threadIdx.x >> 6);
  const unsigned rows = gridDim.x * 4u;
  // row r: 64 consecutive points starting at a multiple of 64 near r / rows of the batch
  const size_t slots = p.npts / 64u;  // >= rows (the host launches this for large batches only)
  const size_t at = (size_t)((unsigned long long)row * (slots - 1u) / (rows > 1u ? rows - 1u : 1u)) * 64u + lane;
  unsigned id = 0;
#pragma unroll
  for (int d = 0; d < N; ++d) {
    const T u = (p.obs[d][at] - p.start[d]) * p.scale[d];
    const int c = u >= (T)1 ? (u < (T)p.top[d] ? (int)u : p.top[d]) : 0;  // (NaN: 0)
    id = id * 0x9E3779B1u + (unsigned)(d == N - 1 ? c / p.sk : c);
  }
  const unsigned prev = (unsigned)__shfl_up((int)id, 1);
  const unsigned changes = (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane > 0 && id != prev));
  if (lane == 0) {
    const unsigned r1 = atomicAdd(&p.work->probe_changes, changes);
    asm volatile("" ::"v"(r1));  // (returning form, answer consumed: performed before `probe_done` is touched)
    if (atomicAdd(&p.work->probe_done, 1u) == rows - 1u) {
      const unsigned total = atomicAdd(&p.work->probe_changes, 0u);
      // coherent: fewer than a quarter of the sampled points start a new line
      const unsigned coherent = total * 4u < rows * 63u ? 1u : 0u;
      atomicExch(&p.work->take_brick, coherent);
      if (p.host_word) __hip_atomic_store(p.host_word, (p.seq << 1) | coherent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      atomicExch(&p.work->probe_changes, 0u);
      atomicExch(&p.work->probe_done, 0u);
    }
  }
}

template <typename T, int N>
static hipError_t probe_t(const GridDesc& g, const void* const* obs, size_t npts, void* work, hipStream_t stream, unsigned* host_word, unsigned seq) {
  ProbeArgs<T, N> p;
  p.host_word = host_word;
  p.seq = seq;
  for (int d = 0; d < N; ++d) {
    p.obs[d] = static_cast<const T*>(obs[d]);
    if (g.kind == kRectilinear) {
      const double span = g.bound_hi[d] - g.bound_lo[d];
      p.start[d] = (T)g.bound_lo[d];
      p.scale[d] = span > 0 ? (T)((double)(g.n[d] - 1) / span) : (T)0;
    } else {
      p.start[d] = (T)g.start[d];
      p.scale[d] = (T)(1.0 / g.step[d]);
    }
    if (!(p.scale[d] > 0) || !(p.scale[d] < (T)1e30)) p.scale[d] = 0;
    p.top[d] = g.n[d] - 2;
  }
  const int per_line = (int)(128 / sizeof(T));
  if (g.method == kLinear && N == 3) p.sk = g.dtype == kF64 ? BrickGeom<double, 0>::SK : (g.brick_cell == 2 ? BrickGeom<float, 2>::SK : BrickGeom<float, 0>::SK);
  else if (g.method == kLinear) p.sk = per_line / 2 - 1;  // 2 x KW bricks stepped KW - 1 (k_linear2_brick.hip)
  else if (g.method == kNearest) p.sk = per_line;         // the C-ordered grid
  else p.sk = 1;                                          // tiles: a footprint per point (cubic_brick.h)
  p.npts = npts;
  p.work = static_cast<SweepWork*>(work);
  hipLaunchKernelGGL((k_sweep_probe<T, N>), dim3(kProbeRows / 4), dim3(256), 0, stream, p);
  return hipGetLastError();
}

// Which handles' automatic sweep launches are gated by a sample: those whose one-pass kernel wins on coherent batches
// (measured: 3-D / 2-D multilinear, nearest-neighbour, 2-D multicubic; the 3-D multicubic sweep wins on lattices too).
bool sweep_probe_applies(const GridDesc& g) {
  // option sweep: -1 automatic (thresholds, then the sample), 0 never, 1 the sweep kernel whatever the batch, 2 the sample
  // decides whatever the batch's size (tests, the fuzzer); sweep_probe: 0 no sample, 1 every launch, 2 thinned out by the host
  if (g.cfg.sweep == 0 || g.cfg.sweep == 1 || g.cfg.sweep_probe == 0) return false;
  if (g.ndims != 2 && g.ndims != 3) return false;
  if (g.method == kCubic) return g.ndims == 2;
  return g.method == kLinear ? g.bricks != nullptr : g.method == kNearest;
}

hipError_t launch_sweep_probe(const GridDesc& g, const void* const* obs, size_t npts, void* work, hipStream_t stream, unsigned* host_word, unsigned seq) {
  if ((g.ndims != 2 && g.ndims != 3) || !work || npts < (size_t)kProbeRows * 64) return hipErrorInvalidValue;
  if (g.ndims == 3) return g.dtype == kF64 ? probe_t<double, 3>(g, obs, npts, work, stream, host_word, seq) : probe_t<float, 3>(g, obs, npts, work, stream, host_word, seq);
  return g.dtype == kF64 ? probe_t<double, 2>(g, obs, npts, work, stream, host_word, seq) : probe_t<float, 2>(g, obs, npts, work, stream, host_word, seq);
}

size_t sweep_probe_word_offset() { return offsetof(SweepWork, take_brick); }

// `work`: a zeroed SweepWork block that no other launch in flight uses (abi_sweep.hip).
template <typename T>
static hipError_t launch_t(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad,
                           void* work, hipStream_t stream) {
  SweepArgs<T> s;
  s.gated = g.sweep_gated ? 1u : 0u;
  BrickArgs<T, 3>& a = s.b;
  a.gate = nullptr;
  a.bricks = static_cast<const T*>(g.sweep_bricks);
  a.out = static_cast<T*>(out);
  a.first_bad = first_bad;
  a.npts = npts;
  for (int d = 0; d < 3; ++d) {
    a.obs[d] = static_cast<const T*>(obs[d]);
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
  }
  a.nbj = g.sweep_nb[1];
  a.nbk = g.sweep_nb[2];
  a.lead_stride[0] = 0;
  a.iters = 1;
  a.ax.use_lds = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  int axr = 0;
  if (g.kind == kRectilinear) {
    axr = lane_axes_mode(g);
    if (axr == 0) axr = 4;  // axes longer than a wave: searched in the (LDS-staged) axis image
    (void)fill_axis_args<T, 3>(g, a.ax, false, /*records=*/axr == 4);
    if (a.ax.use_lds && a.ax.image_bytes > kSweepAxisLds) a.ax.use_lds = 0;
    // the uniform grid over the leading axis' span (bound_lo / bound_hi are g[0] and g[n-1])
    const double span = g.bound_hi[0] - g.bound_lo[0];
    s.key_start = (T)g.bound_lo[0];
    s.key_scale = span > 0 ? (T)((double)(g.n[0] - 1) / span) : (T)0;
  } else {
    s.key_start = (T)g.start[0];
    s.key_scale = (T)(1.0 / g.step[0]);
  }
  if (!(s.key_scale > 0) || !(s.key_scale < (T)1e30)) { s.key_scale = 0; }  // every point in bin 0: still correct
  // regular grids: the correctly rounded reciprocals of the steps, and whether every step lies where the
  // division-free forms of interpn_device.h::step_cell_fast are the reference's values
  s.fastdiv = g.kind == kRectilinear ? 0u : 1u;
  for (int d = 0; d < 3; ++d) {
    const T st = (T)g.step[d];
    const volatile T one = (T)1;   // (one IEEE division in T, at run time)
    s.rstep[d] = g.kind == kRectilinear ? (T)0 : one / st;
    const double mag = st < 0 ? -(double)st : (double)st;
    if (!(mag >= StepCellRange<T>::lo && mag <= StepCellRange<T>::hi)) s.fastdiv = 0;  // (NaN, 0, infinities, far-out steps: the divide sequences)
  }
  s.key_shift = 0;
  while (((g.n[0] - 2) >> s.key_shift) >= 64) ++s.key_shift;
  const size_t chunk = (size_t)64 * (g.kind == kRectilinear ? sweep_rows<T, true>() + sweep_parked<T, true>() : sweep_rows<T, false>() + sweep_parked<T, false>());
  const size_t rounds = (npts + chunk - 1) / chunk;
  if (rounds > 0xFFFFFFF0ull) return hipErrorInvalidValue;
  s.rounds = (unsigned)rounds;
  s.per_shard = (s.rounds + 7u) / 8u;
  s.period = g.cfg.sweep_period > 0 ? (unsigned)g.cfg.sweep_period : 0u;
  s.period_default = 2600;  // 26 us: 0.9 x a round of 16 rows on a 64^3 f64 grid (the first launch through a scratch block; the kernel measures from then on)
  s.work = static_cast<SweepWork*>(work);
  s.stamps = nullptr;
  const unsigned cus = (unsigned)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256);
  unsigned blocks = cus;
  const unsigned need = (s.rounds + (kSweepThreads / 64) - 1) / (kSweepThreads / 64);
  if (blocks > need) blocks = need;
#define SWEEP_KIND(RECT_, AXR_) (g.fma ? go_layout<T, RECT_, true, AXR_>(g, s, blocks, stream) : go_layout<T, RECT_, false, AXR_>(g, s, blocks, stream))
  switch (axr) {
    case 0: return SWEEP_KIND(false, 0);
    case 1: return SWEEP_KIND(true, 1);
    case 2: return SWEEP_KIND(true, 2);
    case 3: return SWEEP_KIND(true, 3);
    case 4: return SWEEP_KIND(true, 4);
  }
#undef SWEEP_KIND
  return hipErrorInvalidValue;
}

hipError_t launch_linear_sweep(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad,
                               void* work, hipStream_t stream) {
  if (g.method == kNearest) return launch_nearest_sweep(g, obs, out, npts, first_bad, work, stream);
  if (g.method == kCubic) return launch_cubic_sweep(g, obs, out, npts, first_bad, work, stream);
  if (g.method == kLinear && g.ndims == 2) return launch_linear2_sweep(g, obs, out, npts, first_bad, work, stream);
  if (g.ndims != 3 || !g.sweep_bricks || !work || npts == 0) return hipErrorInvalidValue;
  for (int d = 0; d < 3; ++d)
    if (reinterpret_cast<uintptr_t>(obs[d]) % 16) return hipErrorInvalidValue;  // the caller checked (abi_sweep.hip)
  if (reinterpret_cast<uintptr_t>(out) % 16) return hipErrorInvalidValue;
  if (g.dtype == kF64) return launch_t<double>(g, obs, out, npts, first_bad, work, stream);
  return launch_t<float>(g, obs, out, npts, first_bad, work, stream);
}

}  // namespace interpn
