// Counting sort of a batch of observation points by the tile position of their multicubic
// footprint (interpn_host.h: "Binned evaluation").  Three launches per slice of at most 2^25
// points: histogram of the bin keys, exclusive scan of the bin totals (<= 1024 tile-position bins,
// <= 4096 class-pair bins), scatter of the
// coordinates (all N dimensions) and of the original indices into bin order.  Every workgroup owns
// a contiguous chunk of 4096 points: it counts its chunk in LDS, sorts the chunk's local indices
// by bin in LDS, reserves one run per non-empty bin with a single global atomic, and copies its
// points into those runs in sorted order, so the copies are made of runs of chunk/bins points
// written by consecutive lanes instead of single scattered elements.  The order of points inside a bin is not
// deterministic; results do not depend on it (a point's result depends on its coordinates only,
// src/multicubic/regular.rs:297-313).
#include <atomic>

#include "interpn_kernels.h"

namespace interpn {

namespace {

constexpr int kHistIters = 32;                             // histogram: 256-lane rows per workgroup
constexpr size_t kHistChunk = (size_t)kBlock * kHistIters;
constexpr size_t kBinChunk = 4096;                         // scatter: points per workgroup (sorted in LDS)
constexpr size_t kRecChunk = 4096;                         // ... when it writes records for the column kernel (measured: 2048 0.234, 4096 0.239, 8192 0.26, 12288 0.32 ms per 1e7 points)
// totals[kMaxBins] | cursor[kMaxBins] | part_prefix[kMaxBins + 1] | ... | work counter of the column kernel at [3 kMaxBins + 16]
constexpr size_t kCounterBytes = (size_t)4 * kMaxBins * sizeof(unsigned);

struct BinParams {
  double start[2];
  double scale[2];
  int ncell[2];
  int shift[2];
  int nb1;
  int nbins;
  int mult;
  int classes;  // 1: bins are saturation classes (BinPlan::classes), 0: footprint cells >> shift
  // the column kernel's local sort key (cubic_column.h; regular grids): class of dim 2 * key_q3 +
  // (class of dim 3 >> key_sh3) < 256, classes as col_class_hint estimates them; stored in the upper
  // eight bits of the point's index word.  key_q3 = 0: off.
  int key_q3, key_sh3;
  double kstart[2], kscale[2];
  int kclasses[2];
  int flag_outside;  // column evaluation with linearised extrapolation: note per bin whether any of its points lies outside the grid along dim 0 (ScatterArgs::flags)
  int tail_den, tail_div;  // column evaluation: the last 1 / tail_den of the bins are cut tail_div times finer (k_bin_scan)
  int scramble; // testing: every 5th point is put into the NEXT bin (the key is only a locality hint: results must not change)
  // rectilinear classes (BinPlan::rect): axes 0, 1 as the kernels search them
  int rect;
  const void* axis_g[2];
  const unsigned* axis_tab[2];
  int axis_n[2], axis_M[2];
  double axis_g0[2], axis_scale[2];
};

// ~ footprint origin: clamp(floor((x - start) / step) - 1, 0, n - 4); NaN -> 0.  A locality hint
// only: it does not have to agree with the kernel's own (exact) cell computation.
__device__ __forceinline__ int bin_cell(double x, double start, double scale, int ncell) {
  const double u = (x - start) * scale;
  return u >= 1.0 ? (u < (double)ncell ? (int)u - 1 : ncell - 1) : 0;
}

// ~ class of a coordinate on a regular axis of n points (`nclass` = n - 1 classes):
// 0 = floc <= 0 (saturated low, inside or outside), c = floc for 1 <= floc <= n - 3 (interior),
// n - 2 = floc >= n - 2 (saturated high); floc = floor((x - start) / step).  Footprint cell =
// clamp(class - 1, 0, n - 4).  The column kernel (cubic_column.h) re-derives the exact class.
__device__ __forceinline__ int bin_class(double x, double start, double scale, int nclass) {
  const double u = (x - start) * scale;
  return u >= 1.0 ? (u < (double)(nclass - 1) ? (int)u : nclass - 1) : 0;
}

// exact class on a rectilinear axis (cubic_column.h::col_rect_class)
template <typename T>
__device__ __forceinline__ int bin_rect_class(const BinParams& p, int d, T x) {
  Axis<T> ax;
  ax.g = static_cast<const T*>(p.axis_g[d]);
  ax.tab = p.axis_tab[d];
  ax.n = p.axis_n[d];
  ax.M = p.axis_M[d];
  ax.g0 = (T)p.axis_g0[d];
  ax.scale = (T)p.axis_scale[d];
  int iloc = axis_partition_point<T>(ax, x) - 2;
  iloc = iloc < -1 ? -1 : (iloc > ax.n - 3 ? ax.n - 3 : iloc);
  return iloc + 1;
}

template <typename T>
__device__ __forceinline__ int bin_key(const BinParams& p, T x0, T x1) {
  int c0, c1;
  if (p.rect) {
    c0 = bin_rect_class<T>(p, 0, x0);
    c1 = bin_rect_class<T>(p, 1, x1);
  } else if (p.classes) {
    c0 = bin_class((double)x0, p.start[0], p.scale[0], p.ncell[0]);
    c1 = bin_class((double)x1, p.start[1], p.scale[1], p.ncell[1]);
  } else {
    c0 = bin_cell((double)x0, p.start[0], p.scale[0], p.ncell[0]) >> p.shift[0];
    c1 = bin_cell((double)x1, p.start[1], p.scale[1], p.ncell[1]) >> p.shift[1];
  }
  // Bins are visited in a scrambled order (key * mult mod nbins, mult coprime with nbins): each
  // bin is its own unit of locality (a tile position's planes), so any order of bins serves the
  // caches equally, but contiguous stretches of the sorted points — what one XCD gets when they
  // are dealt out (cubic_brick.h `eighth`) — then mix boundary and interior cells, whose
  // evaluation costs differ on rectilinear grids (saturation branches): without the scramble the
  // XCDs holding the first and last rows of cells finished 15 % late.
  return (c0 * p.nb1 + c1) * p.mult % p.nbins;
}

// local sort key of the column kernel (BinParams::key_q3), < 256
template <typename T>
__device__ __forceinline__ unsigned column_key(const BinParams& p, T x2, T x3) {
  const int h2 = bin_class((double)x2, p.kstart[0], p.kscale[0], p.kclasses[0]);
  const int h3 = bin_class((double)x3, p.kstart[1], p.kscale[1], p.kclasses[1]);
  return (unsigned)(h2 * p.key_q3 + (h3 >> p.key_sh3)) & 255u;
}

// Does x lie outside the grid along dim 0?  A hint (the uniform grid over the axis' span, rounding
// at the ends): a bin flagged without need keeps the table values in the column kernel, a point
// missed here goes to the table in global memory there — either way the same bits.  NaN: outside.
__device__ __forceinline__ bool bin_outside0(const BinParams& p, double x0) {
  const double u = (x0 - p.start[0]) * p.scale[0];
  return !(u >= 0.0 && u <= (double)p.ncell[0]);
}

// the key the sort uses for point i (testing option `scramble`: see BinParams)
template <typename T>
__device__ __forceinline__ int bin_key_at(const BinParams& p, T x0, T x1, size_t i) {
  int k = bin_key<T>(p, x0, x1);
  if (p.scramble && i % 5 == 0) k = k + 1 < p.nbins ? k + 1 : 0;
  return k;
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_bin_hist(const T* __restrict__ x0, const T* __restrict__ x1, size_t npts,
                                                     const BinParams p, unsigned* __restrict__ totals) {
  __shared__ unsigned hist[kMaxBins];
  for (int b = threadIdx.x; b < p.nbins; b += kBlock) hist[b] = 0;
  __syncthreads();
  // A workgroup walks chunks blockIdx.x, + gridDim.x, ... and flushes its counters once: the flush is one
  // global atomic per non-empty bin and workgroup, all workgroups onto the same `nbins` words.
  for (size_t first = (size_t)blockIdx.x * kHistChunk; first < npts; first += (size_t)gridDim.x * kHistChunk)
  // eight rows at a time, all sixteen loads issued before the first key is computed
  for (int it0 = 0; it0 < kHistIters; it0 += 8) {
    T a0[8], a1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = first + (size_t)(it0 + u) * kBlock + threadIdx.x;
      a0[u] = i < npts ? stream_load(x0 + i) : (T)0;
      a1[u] = i < npts ? stream_load(x1 + i) : (T)0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t i = first + (size_t)(it0 + u) * kBlock + threadIdx.x;
      if (i < npts) atomicAdd(&hist[bin_key_at<T>(p, a0[u], a1[u], i)], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < p.nbins; b += kBlock)
    if (hist[b]) atomicAdd(&totals[b], hist[b]);
}

// cursor[b] = sum of totals[0..b); one workgroup of 1024 threads (kMaxBins <= 1024).  Also the
// work list of the column kernel (cubic_column.h): a bin of c points is cut into
// ceil(c / part_points) parts, part_prefix[b] = parts in front of bin b, part_prefix[nbins] = all.
__global__ void __launch_bounds__(1024) k_bin_scan(unsigned* __restrict__ totals, unsigned* __restrict__ cursor, int nbins,
                                                   unsigned* __restrict__ part_prefix, unsigned part_points, int tail_den, int tail_div) {
  constexpr int BPT = kMaxBins / 1024;  // consecutive bins per thread
  __shared__ unsigned s[1024];
  __shared__ unsigned sp[1024];
  const int t = threadIdx.x;
  unsigned mine[BPT], parts[BPT];
  unsigned sum = 0, sump = 0;
#pragma unroll
  for (int k = 0; k < BPT; ++k) {
    const int b = t * BPT + k;
    mine[k] = b < nbins ? totals[b] : 0u;
    totals[b] = 0;  // ready for the next sort through this scratch block (no separate reset launch)
    // The column kernel's persistent workgroups draw parts in bin order: the last fifth of the bins
    // is cut four times finer (never below 1024 points), so that the launch ends in small pieces
    // (a whole bin is 1/4 of a workgroup's share of cfg4: the end of the launch idled 8 % of the CU time).
    unsigned pp = part_points;
    if (pp && tail_den > 0 && b >= nbins - nbins / tail_den) pp = pp / (unsigned)tail_div > 1024u ? pp / (unsigned)tail_div : (pp < 1024u ? pp : 1024u);
    parts[k] = pp ? (mine[k] + pp - 1u) / pp : 0u;
    sum += mine[k];
    sump += parts[k];
  }
  if (t == 0) totals[3 * kMaxBins + 16] = 0;  // the column kernel's part counter (cubic_column.h)
  if (t < kMaxBins / 32) totals[3 * kMaxBins + 64 + t] = 0;  // one bit per bin: points outside the grid along dim 0 (set by the records scatter)
  s[t] = sum;
  sp[t] = sump;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned add = t >= off ? s[t - off] : 0u;
    const unsigned addp = t >= off ? sp[t - off] : 0u;
    __syncthreads();
    s[t] += add;
    sp[t] += addp;
    __syncthreads();
  }
  unsigned run = s[t] - sum, runp = sp[t] - sump;
#pragma unroll
  for (int k = 0; k < BPT; ++k) {
    const int b = t * BPT + k;
    if (b < nbins) {
      cursor[b] = run;
      part_prefix[b] = runp;
      if (b == nbins - 1) part_prefix[nbins] = runp + parts[k];
    }
    run += mine[k];
    runp += parts[k];
  }
}

template <typename T, int N>
struct ScatterArgs {
  const T* obs[N];
  T* binned[N];
  unsigned* index;
  T* records;      // optional: the sorted points as N-element records (array of structures) instead of `binned`
  unsigned* cursor;
  unsigned* flags;  // one bit per bin (records forms, BinParams::flag_outside), zeroed by k_bin_scan
  size_t npts;
  BinParams p;
};

// Scatter with the chunk sorted in LDS first, so that consecutive lanes write consecutive
// positions of a bin's run.  1024 threads per workgroup and all of a thread's loads issued before
// the first is used: with 256 threads and one load in flight per lane the kernel was bound by
// memory latency (0.57..0.73 ms per 1e7 4-D points).
constexpr int kScatThreads = 1024;
constexpr size_t kStagedLdsMax = 160 * 1024 - 1024;  // dynamic LDS of the staged records scatter at most (a CU's LDS less its static words: wave sums, outside flags)

// CH = points per workgroup.
template <typename T, int N, int CH>
__global__ void __launch_bounds__(kScatThreads) k_bin_scatter(const ScatterArgs<T, N> a) {
  static_assert(kMaxTiledBins <= kScatThreads && kMaxTiledBins <= 65536 && CH <= 65536, "one bin per thread in the scan; keys and local indices are 16-bit");
  constexpr int kIters = CH / kScatThreads;
  __shared__ unsigned fill[kMaxTiledBins];
  __shared__ unsigned lstart[kMaxTiledBins];   // first local position of a bin inside this chunk
  __shared__ unsigned base[kMaxTiledBins];     // first global position of this chunk's run in a bin
  __shared__ unsigned short keys[CH];       // key of local point l
  __shared__ unsigned short sorted_src[CH]; // local point at sorted position j
  __shared__ unsigned short sorted_key[CH];
  const int nbins = a.p.nbins;
  const unsigned tid = threadIdx.x;
  if (tid < kMaxTiledBins) fill[tid] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * CH;
  const unsigned count = (unsigned)((a.npts - first) < (size_t)CH ? (a.npts - first) : (size_t)CH);
  {
    T x0[kIters], x1[kIters];
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const unsigned l = (unsigned)it * kScatThreads + tid;
      x0[it] = l < count ? a.obs[0][first + l] : (T)0;
      x1[it] = l < count ? a.obs[1][first + l] : (T)0;
    }
#pragma unroll
    for (int it = 0; it < kIters; ++it) {
      const unsigned l = (unsigned)it * kScatThreads + tid;
      if (l < count) {
        const int key = bin_key_at<T>(a.p, x0[it], x1[it], first + l);
        keys[l] = (unsigned short)key;
        atomicAdd(&fill[key], 1u);
      }
    }
  }
  __syncthreads();
  // exclusive scan of the bin counts: one bin per thread, Hillis-Steele
  const unsigned mine = tid < kMaxTiledBins ? fill[tid] : 0u;
  if (tid < kMaxTiledBins) lstart[tid] = mine;
  __syncthreads();
  for (int off = 1; off < kMaxTiledBins; off <<= 1) {
    const unsigned add = (tid < kMaxTiledBins && tid >= (unsigned)off) ? lstart[tid - off] : 0u;
    __syncthreads();
    if (tid < kMaxTiledBins) lstart[tid] += add;
    __syncthreads();
  }
  const unsigned excl = tid < kMaxTiledBins ? lstart[tid] - mine : 0u;
  __syncthreads();
  if (tid < kMaxTiledBins) {
    lstart[tid] = excl;
    base[tid] = (mine && (int)tid < nbins) ? atomicAdd(&a.cursor[tid], mine) : 0u;  // one run per non-empty bin
    fill[tid] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned l = (unsigned)it * kScatThreads + tid;
    if (l < count) {
      const unsigned key = keys[l];
      const unsigned lpos = lstart[key] + atomicAdd(&fill[key], 1u);
      sorted_src[lpos] = (unsigned short)l;
      sorted_key[lpos] = (unsigned short)key;
    }
  }
  __syncthreads();
  unsigned pos[kIters];
  size_t src[kIters];
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned j = (unsigned)it * kScatThreads + tid;
    const unsigned jj = j < count ? j : 0u;
    const unsigned key = sorted_key[jj];
    src[it] = first + sorted_src[jj];
    pos[it] = base[key] + (jj - lstart[key]);
  }
  {
#pragma unroll
    for (int d = 0; d < N; ++d) {
      T v[kIters];
#pragma unroll
      for (int it = 0; it < kIters; ++it) v[it] = ((unsigned)it * kScatThreads + tid) < count ? a.obs[d][src[it]] : (T)0;
#pragma unroll
      for (int it = 0; it < kIters; ++it)
        if (((unsigned)it * kScatThreads + tid) < count) a.binned[d][pos[it]] = v[it];
    }
  }
#pragma unroll
  for (int it = 0; it < kIters; ++it)
    if (((unsigned)it * kScatThreads + tid) < count) a.index[pos[it]] = (unsigned)src[it];
}

// Records form of the scatter (column evaluation): no sorting inside the workgroup at all.  Every
// thread keeps its points in registers (read coalesced, all N coordinates), takes a rank inside
// (workgroup, bin) from the LDS histogram's returning atomic, one thread per bin reserves the
// workgroup's run in that bin with one global atomic, and every point is then stored as ONE
// N-element record at run start + rank.  The 32-byte records of a run are written by different
// waves of the same workgroup within microseconds of each other, so the L2 (one XCD's: a workgroup
// lives on one XCD) merges them into whole lines before they leave.  Two barriers, no gathers.
template <typename T, int N, int CH>
__global__ void __launch_bounds__(kScatThreads) k_bin_scatter_records(const ScatterArgs<T, N> a) {
  constexpr int kIters = CH / kScatThreads;
  __shared__ unsigned fill[kMaxBins];
  __shared__ unsigned base[kMaxBins];
  __shared__ unsigned s_flags[kMaxBins / 32];
  const int nbins = a.p.nbins;
  const unsigned tid = threadIdx.x;
  for (int b = (int)tid; b < nbins; b += kScatThreads) fill[b] = 0;
  if (tid < kMaxBins / 32) s_flags[tid] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * CH;
  const unsigned count = (unsigned)((a.npts - first) < (size_t)CH ? (a.npts - first) : (size_t)CH);
  T x[kIters][N];
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned l = (unsigned)it * kScatThreads + tid;
#pragma unroll
    for (int d = 0; d < N; ++d) x[it][d] = l < count ? stream_load(a.obs[d] + first + l) : (T)0;
  }
  unsigned short key[kIters], rank[kIters];
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned l = (unsigned)it * kScatThreads + tid;
    key[it] = 0;
    rank[it] = 0;
    if (l < count) {
      const int k = bin_key_at<T>(a.p, x[it][0], x[it][1], first + l);
      key[it] = (unsigned short)k;
      rank[it] = (unsigned short)atomicAdd(&fill[k], 1u);
      if (a.p.flag_outside && bin_outside0(a.p, (double)x[it][0])) {
        const unsigned w = (unsigned)k >> 5, bit = 1u << ((unsigned)k & 31u);
        if (!(s_flags[w] & bit)) atomicOr(&s_flags[w], bit);  // the workgroup's own bits first: outside points crowd into a few bins
      }
    }
  }
  __syncthreads();
  if (a.p.flag_outside && tid < kMaxBins / 32 && s_flags[tid]) atomicOr(&a.flags[tid], s_flags[tid]);
  for (int b = (int)tid; b < nbins; b += kScatThreads) {
    const unsigned mine = fill[b];
    base[b] = mine ? atomicAdd(&a.cursor[b], mine) : 0u;  // one run per non-empty bin
  }
  __syncthreads();
  typedef T RV __attribute__((ext_vector_type(N)));
  RV* __restrict__ recs = reinterpret_cast<RV*>(a.records);
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned l = (unsigned)it * kScatThreads + tid;
    if (l < count) {
      const unsigned pos = base[key[it]] + rank[it];
      RV r;
#pragma unroll
      for (int d = 0; d < N; ++d) r[d] = x[it][d];
      recs[pos] = r;
      unsigned ix = (unsigned)(first + l);
      if constexpr (N == 4) {
        if (a.p.key_q3) ix |= column_key<T>(a.p, x[it][2], x[it][3]) << 24;
      }
      a.index[pos] = ix;
    }
  }
}

// Records scatter with the chunk's records staged in LDS in bin order (round 4).  The direct form
// above issues one 32-byte store per point to a position nobody else in the wave is near: 1e7
// record stores + 1e7 index stores reach the fabric as 1.5e7 write requests (profiles/
// r04_traffic.json: WRITE_SIZE 679 MB for 360 MB of payload), and the kernel runs at that request
// rate.  Here every point's record goes to LDS at (first local slot of its bin) + (its rank), and
// the workgroup then copies the staged chunk out linearly: consecutive lanes hold consecutive
// records of a run, so a run of k records leaves as ~k / 2 + 1 64-byte requests instead of k, and
// its k index words as one.  One returning LDS atomic per point, one global atomic per non-empty
// bin, a 1024-counter scan, four barriers; no coordinate is read twice.  LDS: CH x (record + 4)
// bytes + three counters per bin — for f64 and 4096 points 144 KiB + 12 nbins bytes, one workgroup
// per CU (opt-in); taken for nbins <= 1024 (the scan is one counter per thread), else the direct form.
template <typename T, int N, int CH>
__global__ void __launch_bounds__(kScatThreads) k_bin_scatter_records_staged(const ScatterArgs<T, N> a) {
  constexpr int kIters = CH / kScatThreads;
  typedef T RV __attribute__((ext_vector_type(N)));
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_sc[];
  static_assert(CH <= 4096 && kScatThreads <= 1024, "meta word: 10 bits of bin, 8 of column key, 12 of local index");
  RV* const lrec = reinterpret_cast<RV*>(smem_sc);                                        // [CH] records in bin order
  unsigned* const lmeta = reinterpret_cast<unsigned*>(smem_sc + (size_t)CH * sizeof(RV)); // [CH] bin | column key << 10 | local index << 20
  unsigned* const fill = lmeta + CH;                                                       // [nbins] points per bin
  const int nbins = a.p.nbins;  // <= kScatThreads
  unsigned* const lstart = fill + nbins;                                                   // first local slot of a bin
  unsigned* const base = lstart + nbins;                                                   // first global slot of this chunk's run
  __shared__ unsigned s_wsum[kScatThreads / 64];
  __shared__ unsigned s_flags[kMaxBins / 32];
  const unsigned tid = threadIdx.x;
  if ((int)tid < nbins) fill[tid] = 0;
  if (tid < kMaxBins / 32) s_flags[tid] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * CH;
  const unsigned count = (unsigned)((a.npts - first) < (size_t)CH ? (a.npts - first) : (size_t)CH);
  T x[kIters][N];
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned l = (unsigned)it * kScatThreads + tid;
#pragma unroll
    for (int d = 0; d < N; ++d) x[it][d] = l < count ? stream_load(a.obs[d] + first + l) : (T)0;
  }
  unsigned short key[kIters], rank[kIters];
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned l = (unsigned)it * kScatThreads + tid;
    key[it] = 0;
    rank[it] = 0;
    if (l < count) {
      const int k = bin_key_at<T>(a.p, x[it][0], x[it][1], first + l);
      key[it] = (unsigned short)k;
      rank[it] = (unsigned short)atomicAdd(&fill[k], 1u);
      if (a.p.flag_outside && bin_outside0(a.p, (double)x[it][0])) {
        const unsigned w = (unsigned)k >> 5, bit = 1u << ((unsigned)k & 31u);
        if (!(s_flags[w] & bit)) atomicOr(&s_flags[w], bit);  // the workgroup's own bits first: outside points crowd into a few bins
      }
    }
  }
  __syncthreads();
  if (a.p.flag_outside && tid < kMaxBins / 32 && s_flags[tid]) atomicOr(&a.flags[tid], s_flags[tid]);
  // exclusive scan of the bin counts (one bin per thread): wave scan + wave totals; one run per non-empty bin
  {
    const unsigned mine = (int)tid < nbins ? fill[tid] : 0u;
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned up = (unsigned)__shfl_up((int)incl, off);
      if ((tid & 63u) >= (unsigned)off) incl += up;
    }
    if ((tid & 63u) == 63u) s_wsum[tid >> 6] = incl;
    __syncthreads();
    unsigned run = incl - mine;
    for (unsigned w = 0; w < (tid >> 6); ++w) run += s_wsum[w];
    if ((int)tid < nbins) {
      lstart[tid] = run;
      base[tid] = mine ? atomicAdd(&a.cursor[tid], mine) : 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned l = (unsigned)it * kScatThreads + tid;
    if (l < count) {
      const unsigned lp = lstart[key[it]] + rank[it];
      RV r;
#pragma unroll
      for (int d = 0; d < N; ++d) r[d] = x[it][d];
      lrec[lp] = r;
      unsigned ck = 0;
      if constexpr (N == 4) {
        if (a.p.key_q3) ck = column_key<T>(a.p, x[it][2], x[it][3]);
      }
      lmeta[lp] = (unsigned)key[it] | (ck << 10) | (l << 20);
    }
  }
  __syncthreads();
  RV* __restrict__ recs = reinterpret_cast<RV*>(a.records);
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const unsigned j = (unsigned)it * kScatThreads + tid;
    if (j < count) {
      const unsigned m = lmeta[j];
      const unsigned k = m & 1023u;
      const unsigned pos = base[k] + (j - lstart[k]);
      recs[pos] = lrec[j];
      a.index[pos] = (unsigned)(first + (m >> 20)) | (((m >> 10) & 255u) << 24);  // key bits are zero without column keys
    }
  }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T, int N>
hipError_t bin_points_n(const BinParams& p, const void* const* obs, size_t npts, void* scratch, const void** binned_obs,
                        const unsigned** index, BinExtras* extras, unsigned part_points, hipStream_t stream, hipEvent_t* stage,
                        bool totals_clean, bool staged, unsigned hist_wgs, size_t lds_per_cu) {
  unsigned char* base = static_cast<unsigned char*>(scratch);
  unsigned* totals = reinterpret_cast<unsigned*>(base);
  unsigned* cursor = totals + kMaxBins;
  unsigned* part_prefix = cursor + kMaxBins;
  size_t off = align_up(kCounterBytes, 256);
  unsigned* idx = reinterpret_cast<unsigned*>(base + off);
  off += align_up(npts * sizeof(unsigned), 256);
  ScatterArgs<T, N> a;
  a.records = nullptr;
  for (int d = 0; d < N; ++d) {
    a.obs[d] = static_cast<const T*>(obs[d]);
    a.binned[d] = reinterpret_cast<T*>(base + off);
    binned_obs[d] = a.binned[d];
    off += align_up(npts * sizeof(T), 256);
  }
  if (extras) {  // column evaluation: the same room holds the points as N-element records
    a.records = a.binned[0];
    extras->records = a.records;
    extras->bin_end = cursor;  // after the scatter every cursor stands at the end of its bin
    extras->part_prefix = part_prefix;
    extras->work = totals + 3 * kMaxBins + 16;
    extras->bin_flags = totals + 3 * kMaxBins + 64;
  }
  a.index = idx;
  a.flags = totals + 3 * kMaxBins + 64;
  a.cursor = cursor;
  a.npts = npts;
  a.p = p;
  *index = idx;
  auto mark = [&](int k) { if (stage) (void)hipEventRecord(stage[k], stream); };  // per-stage timing on request (bench.py)
  mark(0);
  hipError_t e = hipSuccess;
  if (!totals_clean) {  // first use of the block, or a sequence through it was cut short: reset the counters
    e = hipMemsetAsync(totals, 0, kMaxBins * sizeof(unsigned), stream);
    if (e != hipSuccess) return e;
  }
  unsigned hblocks = (unsigned)((npts + kHistChunk - 1) / kHistChunk);
  if (hist_wgs > 0 && hblocks > hist_wgs) hblocks = hist_wgs;
  hipLaunchKernelGGL(k_bin_hist<T>, dim3(hblocks), dim3(kBlock), 0, stream, a.obs[0], a.obs[1], npts, p, totals);
  mark(1);
  hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, stream, totals, cursor, p.nbins, part_prefix, part_points, p.tail_den, p.tail_div);
  mark(2);
  if (extras) {
    if constexpr (N == 4) {
      const unsigned blocks = (unsigned)((npts + kRecChunk - 1) / kRecChunk);
      const size_t staged_lds = kRecChunk * (N * sizeof(T) + 4) + (size_t)3 * (size_t)p.nbins * sizeof(unsigned);
      // The staged form needs its chunk in LDS beside the kernel's static words (wave sums, outside
      // flags: under 1 KiB): only where the device's CU has that much, and opted in to what is
      // needed, not to the maximum.  A device (or an opt-in) that does not give it takes the direct form.
      bool use_staged = p.nbins <= kScatThreads && staged && staged_lds <= kStagedLdsMax && staged_lds + 1024 <= lds_per_cu;
      auto kern = k_bin_scatter_records_staged<T, N, (int)kRecChunk>;
      if (use_staged && staged_lds > 64 * 1024) {
        static std::atomic<unsigned long long> opted_bytes[64];  // per device: the largest size opted in to so far
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
        if (dev < 0 || dev >= 64 || opted_bytes[dev].load() < staged_lds) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)staged_lds) != hipSuccess) {
            (void)hipGetLastError();
            use_staged = false;
          } else if (dev >= 0 && dev < 64) {
            unsigned long long cur = opted_bytes[dev].load();
            while (cur < staged_lds && !opted_bytes[dev].compare_exchange_weak(cur, staged_lds)) {}
          }
        }
      }
      if (use_staged) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(kScatThreads), staged_lds, stream, a);
      } else {
        hipLaunchKernelGGL((k_bin_scatter_records<T, N, (int)kRecChunk>), dim3(blocks), dim3(kScatThreads), 0, stream, a);
      }
    } else {
      return hipErrorInvalidValue;
    }
  } else {
    const unsigned blocks = (unsigned)((npts + kBinChunk - 1) / kBinChunk);
    hipLaunchKernelGGL((k_bin_scatter<T, N, (int)kBinChunk>), dim3(blocks), dim3(kScatThreads), 0, stream, a);
  }
  mark(3);
  return hipGetLastError();
}

}  // namespace

bool make_bin_plan(const GridDesc& g, size_t table_bytes, BinPlan* plan, bool classes) {
  if (g.method != kCubic || g.ndims < 2) return false;
  BinPlan p;
  p.classes = classes ? 1 : 0;
  if (classes && g.kind != kRegular) {  // rectilinear classes: searched exactly, on axes with a bucket table
    if (g.axis_buckets[0] <= 0 || g.axis_buckets[1] <= 0 || !g.axis_image) return false;
    p.rect = 1;
    for (int d = 0; d < 2; ++d) {
      p.axis_g[d] = g.grid[d];
      p.axis_tab[d] = reinterpret_cast<const unsigned*>(static_cast<const unsigned char*>(g.axis_image) + g.axis_tab_off[d]);
      p.axis_n[d] = g.n[d];
      p.axis_M[d] = g.axis_buckets[d];
      p.axis_g0[d] = g.axis_g0[d];
      p.axis_scale[d] = g.axis_scale[d];
    }
  }
  for (int d = 0; d < 2; ++d) {
    p.ncell[d] = classes ? g.n[d] - 1 : g.n[d] - 3;
    if (p.ncell[d] < 1) return false;
    if (g.kind == kRegular) {
      p.start[d] = g.start[d];
      p.scale[d] = 1.0 / g.step[d];
    } else {
      // rectilinear: the uniform grid over the same span (bound_lo / bound_hi are g[0] and g[n-1])
      const double span = g.bound_hi[d] - g.bound_lo[d];
      p.start[d] = g.bound_lo[d];
      p.scale[d] = span > 0 ? (double)(g.n[d] - 1) / span : 0.0;
    }
    if (!(p.scale[d] > 0) || !(p.scale[d] < 1e300)) return false;
  }
  // Bins sized so that one bin's share of the table is about half a MiB (the workgroups an XCD has
  // in flight then share an L2-sized piece of it): 32^4 f64 (113 MiB) -> 225 bins, 48^4 (597 MiB)
  // -> 529; never fewer than 64 (balance across the XCDs), never more than kMaxBins.
  const size_t share = thresholds(g.cfg).bin_table_share;  // L2 / 8 = 512 KiB on MI355X
  long long target = (long long)(table_bytes / (share ? share : 1));
  target = target < 64 ? 64 : (target > kMaxTiledBins ? kMaxTiledBins : target);
  if (classes) {  // one bin per pair of saturation classes (column evaluation): only if they all fit
    if ((long long)p.ncell[0] * p.ncell[1] > kMaxBins) return false;
    target = kMaxBins;
  }
  auto bins = [&](int d) { return ((p.ncell[d] - 1) >> p.shift[d]) + 1; };
  while ((long long)bins(0) * bins(1) > target) {
    if (bins(0) >= bins(1)) ++p.shift[0];
    else ++p.shift[1];
  }
  p.nb1 = bins(1);
  p.nbins = bins(0) * bins(1);
  // a multiplier near the golden-ratio fraction of nbins, coprime with it
  auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
  p.mult = (int)(0.6180339887 * p.nbins);
  if (p.mult < 1) p.mult = 1;
  while (gcd(p.mult, p.nbins) != 1) ++p.mult;
  p.inv_mult = 1;  // (b * inv_mult) % nbins undoes (key * mult) % nbins
  for (int v = 1; v < p.nbins; ++v)
    if ((long long)v * p.mult % p.nbins == 1) { p.inv_mult = v; break; }
  *plan = p;
  return true;
}

size_t bin_scratch_bytes(const GridDesc& g, size_t slice_points) {
  const size_t elem = g.dtype == kF64 ? 8 : 4;
  const size_t arrays = g.ndims == 3 ? 4 : (size_t)g.ndims;  // 3-D records take the 4-D form (bin_points)
  size_t b = align_up(kCounterBytes, 256) + align_up(slice_points * sizeof(unsigned), 256) +
             arrays * align_up(slice_points * elem, 256);
  return b;
}

hipError_t bin_points(const GridDesc& g, const BinPlan& plan, const void* const* obs, size_t npts, void* scratch,
                      const void** binned_obs, const unsigned** index, hipStream_t stream, BinExtras* extras,
                      unsigned part_points, hipEvent_t* stage, bool totals_clean) {
  if (extras && g.ndims != 4 && g.ndims != 3) return hipErrorInvalidValue;
  // 3-D points sorted into records (cubic3_column.h) take the 4-D record form with dim 2 twice: (x0, x1, x2, x2)
  const void* obs4[4] = {obs[0], obs[1], g.ndims > 2 ? obs[2] : nullptr, g.ndims > 3 ? obs[3] : (g.ndims > 2 ? obs[2] : nullptr)};
  const bool as4 = extras && g.ndims == 3;
  if (npts == 0 || npts > kBinSlicePoints) return hipErrorInvalidValue;
  BinParams p;
  for (int d = 0; d < 2; ++d) {
    p.start[d] = plan.start[d];
    p.scale[d] = plan.scale[d];
    p.ncell[d] = plan.ncell[d];
    p.shift[d] = plan.shift[d];
  }
  p.nb1 = plan.nb1;
  p.nbins = plan.nbins;
  p.mult = plan.mult;
  p.classes = plan.classes;
  p.scramble = g.cfg.bin_scramble;
  p.flag_outside = (extras && g.linearize) ? 1 : 0;
  p.tail_den = g.cfg.column_tail >> 4;
  p.tail_div = (g.cfg.column_tail & 15) ? (g.cfg.column_tail & 15) : 1;
  p.rect = plan.rect;
  for (int d = 0; d < 2; ++d) {
    p.axis_g[d] = plan.axis_g[d];
    p.axis_tab[d] = plan.axis_tab[d];
    p.axis_n[d] = plan.axis_n[d];
    p.axis_M[d] = plan.axis_M[d];
    p.axis_g0[d] = plan.axis_g0[d];
    p.axis_scale[d] = plan.axis_scale[d];
  }
  p.key_q3 = 0;
  p.key_sh3 = 0;
  for (int d = 0; d < 2; ++d) { p.kstart[d] = 0; p.kscale[d] = 0; p.kclasses[d] = 1; }
  if (extras && extras->key_q3 > 0 && g.ndims == 4 && g.kind == kRegular && npts <= kColumnKeySlicePoints) {
    p.key_q3 = extras->key_q3;
    p.key_sh3 = extras->key_sh3;
    for (int d = 2; d < 4; ++d) {
      p.kstart[d - 2] = g.start[d];
      p.kscale[d - 2] = 1.0 / g.step[d];
      p.kclasses[d - 2] = g.n[d] - 1;
    }
  } else if (extras) {
    extras->key_q3 = 0;  // tells the launcher: plain indices
  }
#define GO(T, N) return bin_points_n<T, N>(p, (as4 ? obs4 : obs), npts, scratch, binned_obs, index, extras, part_points, stream, stage, totals_clean, g.cfg.scatter_staged != 0, (unsigned)(g.cfg.hist_wgs_per_cu > 0 ? g.cfg.hist_wgs_per_cu * (g.cfg.num_cus > 0 ? g.cfg.num_cus : 256) : 0), (size_t)g.cfg.lds_per_cu)
  if (g.dtype == kF64) {
    switch (as4 ? 4 : g.ndims) {
      case 2: GO(double, 2);
      case 3: GO(double, 3);
      case 4: GO(double, 4);
    }
  } else {
    switch (as4 ? 4 : g.ndims) {
      case 2: GO(float, 2);
      case 3: GO(float, 3);
      case 4: GO(float, 4);
    }
  }
#undef GO
  return hipErrorInvalidValue;
}

}  // namespace interpn
