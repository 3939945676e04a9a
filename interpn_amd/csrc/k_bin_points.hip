// Counting sort of a batch of observation points by the tile position of their multicubic
// footprint (interpn_host.h: "Binned evaluation").  Three launches per slice of at most 2^25
// points: histogram of the bin keys, exclusive scan of the <= 256 bin totals, scatter of the
// coordinates (all N dimensions) and of the original indices into bin order.  Every workgroup owns
// a contiguous chunk of 4096 points: it counts its chunk in LDS, sorts the chunk's local indices
// by bin in LDS, reserves one run per non-empty bin with a single global atomic, and copies its
// points into those runs in sorted order, so the copies are made of runs of chunk/bins points
// written by consecutive lanes instead of single scattered elements.  The order of points inside a bin is not
// deterministic; results do not depend on it (a point's result depends on its coordinates only,
// src/multicubic/regular.rs:297-313).
#include "interpn_kernels.h"

namespace interpn {

namespace {

constexpr int kHistIters = 32;                             // histogram: 256-lane rows per workgroup
constexpr size_t kHistChunk = (size_t)kBlock * kHistIters;
constexpr size_t kBinChunk = 4096;                         // scatter: points per workgroup (sorted in LDS)
// totals[kMaxBins] | cursor[kMaxBins] | part_prefix[kMaxBins + 1] (+ padding)
constexpr size_t kCounterBytes = (size_t)4 * kMaxBins * sizeof(unsigned);

struct BinParams {
  double start[2];
  double scale[2];
  int ncell[2];
  int shift[2];
  int nb1;
  int nbins;
  int mult;
};

// ~ footprint origin: clamp(floor((x - start) / step) - 1, 0, n - 4); NaN -> 0.  A locality hint
// only: it does not have to agree with the kernel's own (exact) cell computation.
__device__ __forceinline__ int bin_cell(double x, double start, double scale, int ncell) {
  const double u = (x - start) * scale;
  return u >= 1.0 ? (u < (double)ncell ? (int)u - 1 : ncell - 1) : 0;
}

template <typename T>
__device__ __forceinline__ int bin_key(const BinParams& p, T x0, T x1) {
  const int c0 = bin_cell((double)x0, p.start[0], p.scale[0], p.ncell[0]) >> p.shift[0];
  const int c1 = bin_cell((double)x1, p.start[1], p.scale[1], p.ncell[1]) >> p.shift[1];
  // Bins are visited in a scrambled order (key * mult mod nbins, mult coprime with nbins): each
  // bin is its own unit of locality (a tile position's planes), so any order of bins serves the
  // caches equally, but contiguous stretches of the sorted points — what one XCD gets when they
  // are dealt out (cubic_brick.h `eighth`) — then mix boundary and interior cells, whose
  // evaluation costs differ on rectilinear grids (saturation branches): without the scramble the
  // XCDs holding the first and last rows of cells finished 15 % late.
  return (c0 * p.nb1 + c1) * p.mult % p.nbins;
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_bin_hist(const T* __restrict__ x0, const T* __restrict__ x1, size_t npts,
                                                     const BinParams p, unsigned* __restrict__ totals) {
  __shared__ unsigned hist[kMaxBins];
  for (int b = threadIdx.x; b < p.nbins; b += kBlock) hist[b] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * kHistChunk;
  for (int it = 0; it < kHistIters; ++it) {
    const size_t i = first + (size_t)it * kBlock + threadIdx.x;
    if (i < npts) atomicAdd(&hist[bin_key<T>(p, x0[i], x1[i])], 1u);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < p.nbins; b += kBlock)
    if (hist[b]) atomicAdd(&totals[b], hist[b]);
}

// cursor[b] = sum of totals[0..b); one workgroup of 1024 threads (kMaxBins <= 1024).  Also the
// work list of the column kernel (cubic_column.h): a bin of c points is cut into
// ceil(c / part_points) parts, part_prefix[b] = parts in front of bin b, part_prefix[nbins] = all.
__global__ void __launch_bounds__(1024) k_bin_scan(const unsigned* __restrict__ totals, unsigned* __restrict__ cursor, int nbins,
                                                   unsigned* __restrict__ part_prefix, unsigned part_points) {
  __shared__ unsigned s[1024];
  __shared__ unsigned sp[1024];
  const int t = threadIdx.x;
  const unsigned mine = t < nbins ? totals[t] : 0u;
  const unsigned parts = part_points ? (mine + part_points - 1u) / part_points : 0u;
  s[t] = mine;
  sp[t] = parts;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const unsigned add = t >= off ? s[t - off] : 0u;
    const unsigned addp = t >= off ? sp[t - off] : 0u;
    __syncthreads();
    s[t] += add;
    sp[t] += addp;
    __syncthreads();
  }
  if (t < nbins) {
    cursor[t] = s[t] - mine;
    part_prefix[t] = sp[t] - parts;
    if (t == nbins - 1) part_prefix[nbins] = sp[t];
  }
}

template <typename T, int N>
struct ScatterArgs {
  const T* obs[N];
  T* binned[N];
  unsigned* index;
  unsigned* rank;  // optional: rank[i] = sorted position of point i (the inverse of `index`)
  unsigned* cursor;
  size_t npts;
  BinParams p;
};

// Scatter with the chunk sorted in LDS first, so that consecutive lanes write consecutive
// positions of a bin's run.  1024 threads per workgroup and all of a thread's loads issued before
// the first is used: with 256 threads and one load in flight per lane the kernel was bound by
// memory latency (0.57..0.73 ms per 1e7 4-D points).
constexpr int kScatThreads = 1024;
constexpr int kScatIters = (int)(kBinChunk / kScatThreads);

template <typename T, int N>
__global__ void __launch_bounds__(kScatThreads) k_bin_scatter(const ScatterArgs<T, N> a) {
  static_assert(kMaxBins <= kScatThreads && kMaxBins <= 65536 && kBinChunk <= 65536, "one bin per thread in the scan; keys and local indices are 16-bit");
  __shared__ unsigned fill[kMaxBins];
  __shared__ unsigned lstart[kMaxBins];   // first local position of a bin inside this chunk
  __shared__ unsigned base[kMaxBins];     // first global position of this chunk's run in a bin
  __shared__ unsigned short keys[kBinChunk];       // key of local point l
  __shared__ unsigned short sorted_src[kBinChunk]; // local point at sorted position j
  __shared__ unsigned short sorted_key[kBinChunk];
  __shared__ unsigned lpos32[kBinChunk];            // sorted position of local point l (rank output)
  const int nbins = a.p.nbins;
  const unsigned tid = threadIdx.x;
  if (tid < kMaxBins) fill[tid] = 0;
  __syncthreads();
  const size_t first = (size_t)blockIdx.x * kBinChunk;
  const unsigned count = (unsigned)((a.npts - first) < kBinChunk ? (a.npts - first) : kBinChunk);
  {
    T x0[kScatIters], x1[kScatIters];
#pragma unroll
    for (int it = 0; it < kScatIters; ++it) {
      const unsigned l = (unsigned)it * kScatThreads + tid;
      x0[it] = l < count ? a.obs[0][first + l] : (T)0;
      x1[it] = l < count ? a.obs[1][first + l] : (T)0;
    }
#pragma unroll
    for (int it = 0; it < kScatIters; ++it) {
      const unsigned l = (unsigned)it * kScatThreads + tid;
      if (l < count) {
        const int key = bin_key<T>(a.p, x0[it], x1[it]);
        keys[l] = (unsigned short)key;
        atomicAdd(&fill[key], 1u);
      }
    }
  }
  __syncthreads();
  // exclusive scan of the bin counts: one bin per thread, Hillis-Steele
  const unsigned mine = tid < kMaxBins ? fill[tid] : 0u;
  if (tid < kMaxBins) lstart[tid] = mine;
  __syncthreads();
  for (int off = 1; off < kMaxBins; off <<= 1) {
    const unsigned add = (tid < kMaxBins && tid >= (unsigned)off) ? lstart[tid - off] : 0u;
    __syncthreads();
    if (tid < kMaxBins) lstart[tid] += add;
    __syncthreads();
  }
  const unsigned excl = tid < kMaxBins ? lstart[tid] - mine : 0u;
  __syncthreads();
  if (tid < kMaxBins) {
    lstart[tid] = excl;
    base[tid] = (mine && (int)tid < nbins) ? atomicAdd(&a.cursor[tid], mine) : 0u;  // one run per non-empty bin
    fill[tid] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kScatIters; ++it) {
    const unsigned l = (unsigned)it * kScatThreads + tid;
    if (l < count) {
      const unsigned key = keys[l];
      const unsigned lpos = lstart[key] + atomicAdd(&fill[key], 1u);
      sorted_src[lpos] = (unsigned short)l;
      sorted_key[lpos] = (unsigned short)key;
    }
  }
  __syncthreads();
  unsigned pos[kScatIters];
  size_t src[kScatIters];
#pragma unroll
  for (int it = 0; it < kScatIters; ++it) {
    const unsigned j = (unsigned)it * kScatThreads + tid;
    const unsigned jj = j < count ? j : 0u;
    const unsigned key = sorted_key[jj];
    src[it] = first + sorted_src[jj];
    pos[it] = base[key] + (jj - lstart[key]);
  }
#pragma unroll
  for (int d = 0; d < N; ++d) {
    T v[kScatIters];
#pragma unroll
    for (int it = 0; it < kScatIters; ++it) v[it] = ((unsigned)it * kScatThreads + tid) < count ? a.obs[d][src[it]] : (T)0;
#pragma unroll
    for (int it = 0; it < kScatIters; ++it)
      if (((unsigned)it * kScatThreads + tid) < count) a.binned[d][pos[it]] = v[it];
  }
#pragma unroll
  for (int it = 0; it < kScatIters; ++it)
    if (((unsigned)it * kScatThreads + tid) < count) a.index[pos[it]] = (unsigned)src[it];
  if (a.rank) {
    // rank[] is indexed by ORIGINAL position: hand the sorted positions over through LDS so that
    // consecutive lanes write consecutive words.
#pragma unroll
    for (int it = 0; it < kScatIters; ++it) {
      const unsigned j = (unsigned)it * kScatThreads + tid;
      if (j < count) lpos32[sorted_src[j]] = pos[it];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kScatIters; ++it) {
      const unsigned l = (unsigned)it * kScatThreads + tid;
      if (l < count) a.rank[first + l] = lpos32[l];
    }
  }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

template <typename T, int N>
hipError_t bin_points_n(const BinParams& p, const void* const* obs, size_t npts, void* scratch, const void** binned_obs,
                        const unsigned** index, BinExtras* extras, unsigned part_points, hipStream_t stream) {
  unsigned char* base = static_cast<unsigned char*>(scratch);
  unsigned* totals = reinterpret_cast<unsigned*>(base);
  unsigned* cursor = totals + kMaxBins;
  unsigned* part_prefix = cursor + kMaxBins;
  size_t off = align_up(kCounterBytes, 256);
  unsigned* idx = reinterpret_cast<unsigned*>(base + off);
  off += align_up(npts * sizeof(unsigned), 256);
  ScatterArgs<T, N> a;
  for (int d = 0; d < N; ++d) {
    a.obs[d] = static_cast<const T*>(obs[d]);
    a.binned[d] = reinterpret_cast<T*>(base + off);
    binned_obs[d] = a.binned[d];
    off += align_up(npts * sizeof(T), 256);
  }
  a.rank = nullptr;
  if (extras) {  // column evaluation: rank (original -> sorted) and a sorted-order result array
    a.rank = reinterpret_cast<unsigned*>(base + off);
    off += align_up(npts * sizeof(unsigned), 256);
    extras->rank = a.rank;
    extras->res_sorted = base + off;
    off += align_up(npts * sizeof(T), 256);
    extras->bin_end = cursor;  // after the scatter every cursor stands at the end of its bin
    extras->part_prefix = part_prefix;
  }
  a.index = idx;
  a.cursor = cursor;
  a.npts = npts;
  a.p = p;
  *index = idx;
  hipError_t e = hipMemsetAsync(totals, 0, kMaxBins * sizeof(unsigned), stream);
  if (e != hipSuccess) return e;
  const unsigned blocks = (unsigned)((npts + kBinChunk - 1) / kBinChunk);
  const unsigned hblocks = (unsigned)((npts + kHistChunk - 1) / kHistChunk);
  hipLaunchKernelGGL(k_bin_hist<T>, dim3(hblocks), dim3(kBlock), 0, stream, a.obs[0], a.obs[1], npts, p, totals);
  hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, stream, totals, cursor, p.nbins, part_prefix, part_points);
  hipLaunchKernelGGL((k_bin_scatter<T, N>), dim3(blocks), dim3(kScatThreads), 0, stream, a);
  return hipGetLastError();
}

}  // namespace

bool make_bin_plan(const GridDesc& g, size_t table_bytes, BinPlan* plan, bool exact_cells) {
  if (g.method != kCubic || g.ndims < 2) return false;
  BinPlan p;
  for (int d = 0; d < 2; ++d) {
    p.ncell[d] = g.n[d] - 3;
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
  long long target = (long long)(table_bytes >> 19);
  target = target < 64 ? 64 : (target > kMaxBins ? kMaxBins : target);
  if (exact_cells) {  // one bin per (i, j) cell (column evaluation): only if they all fit
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
  size_t b = align_up(kCounterBytes, 256) + align_up(slice_points * sizeof(unsigned), 256) +
             (size_t)g.ndims * align_up(slice_points * elem, 256);
  // column evaluation (4-D): rank + results in sorted order
  if (g.ndims == 4) b += align_up(slice_points * sizeof(unsigned), 256) + align_up(slice_points * elem, 256);
  return b;
}

hipError_t bin_points(const GridDesc& g, const BinPlan& plan, const void* const* obs, size_t npts, void* scratch,
                      const void** binned_obs, const unsigned** index, hipStream_t stream, BinExtras* extras,
                      unsigned part_points) {
  if (extras && g.ndims != 4) return hipErrorInvalidValue;
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
#define GO(T, N) return bin_points_n<T, N>(p, obs, npts, scratch, binned_obs, index, extras, part_points, stream)
  if (g.dtype == kF64) {
    switch (g.ndims) {
      case 2: GO(double, 2);
      case 3: GO(double, 3);
      case 4: GO(double, 4);
    }
  } else {
    switch (g.ndims) {
      case 2: GO(float, 2);
      case 3: GO(float, 3);
      case 4: GO(float, 4);
    }
  }
#undef GO
  return hipErrorInvalidValue;
}

}  // namespace interpn
