// 4-D multicubic on SORTED points with the table column of a cell resident in LDS.
//
// After the counting sort of k_bin_points.hip with one bin per (i, j) cell of dims 0 and 1, all
// points of a bin read the SAME 4 x 4 (i, j) footprint of every (k, l) plane: n2 x n3 tiles of the
// fully overlapped tile table (cubic_brick.h), 128 B each in f64 — 128 KiB for cfg4's 32 x 32
// planes, which fits the 160 KiB LDS of a CU.  A 1024-thread workgroup therefore loads that column
// ONCE (LDS-DMA, one 1-KiB instruction per 8 tiles) and then evaluates its share of the bin's
// points entirely out of LDS: a point still reads its 16 tiles = 2 KiB, but from the CU's own
// LDS (256 B/clk) instead of 16 L2 lines (the tiled kernel on sorted points spends 1.04 ms per
// 1e7 points on 1.6e8 L2-hit line requests; here the L2 sees 1024 lines per ~6000 points).
//
// LDS layout: tiles are interleaved sixteen at a time — piece c (16 bytes) of tile T sits at
//   (T >> 4) * (16 * TILE) + c * 256 + (T & 15) * 16,          TILE = 16 * sizeof(T) bytes
// so a `ds_read_b128` of piece c by 16 lanes (one LDS lane group) with different tiles hits the
// 16-byte slot (T & 15): different slots for different (l + dl) & 15, no fixed conflict pattern
// (tile-major storage would put all 16 lanes of a group on two slots: 8-way conflicts), and the
// eight pieces of a tile are reached from ONE address register with immediate offsets c * 256.
//
// Arithmetic, plane order and reduction tree are those of cubic_brick.h / the reference
// (src/multicubic/regular.rs:325-623): bit-identical results.  A point whose exact cell is not
// the workgroup's (the sort's cell estimate multiplies by a reciprocal, the kernel divides like
// the reference; they can disagree on a cell boundary) is evaluated from the table in global
// memory by the same code path as the unsorted kernel's gather.
#pragma once
#include "cubic_brick.h"

namespace interpn {

template <typename T>
struct CubicColumnArgs {
  const T* tiles;            // fully overlapped tile table [plane (k, l)][bi][bj][16]
  unsigned table_bytes;
  const T* obs[4];           // the slice's points in bin order
  const unsigned* index;     // sorted position -> index within the slice
  T* out;                    // caller's output of the slice (scattered write) ...
  T* res_sorted;             // ... or, if non-null, results in sorted order (un-permuted by k_unpermute)
  unsigned long long* first_bad;
  size_t index_base;
  size_t npts;
  const unsigned* bin_end;      // end of bin b in sorted order (the scatter's cursors after the scatter)
  const unsigned* part_prefix;  // workgroups (parts) in front of bin b; [nbins] = total
  int nbins;
  int nb1;                   // cells along dim 1
  unsigned inv_mult;         // sorted bin b holds cell key (b * inv_mult) % nbins
  unsigned part_points;      // a bin of c points is cut into ceil(c / part_points) equal parts
  T start[4];
  T step[4];
  int n[4];
  int linearize;
  unsigned plane_stride[4];  // table elements per unit index of dims 2, 3
  unsigned nbj;
};

constexpr int kColThreads = 1024;

template <typename T> constexpr unsigned col_tile_bytes() { return 16u * (unsigned)sizeof(T); }
// LDS bytes of a column of `ntiles` tiles (whole 16-tile groups)
template <typename T> inline size_t col_lds_bytes(unsigned ntiles) { return (size_t)((ntiles + 15u) / 16u) * 16u * col_tile_bytes<T>(); }

template <typename T>
__device__ __forceinline__ unsigned col_tile_base(unsigned tile) {  // LDS byte offset of piece 0 of `tile`
  return (tile >> 4) * (16u * col_tile_bytes<T>()) + (tile & 15u) * 16u;
}

// my tile (16 elements, e = ei * 4 + ej) out of the LDS column
template <typename T>
__device__ __forceinline__ void col_take_tile(unsigned lds_col, unsigned tile, T (&v)[16]) {
  constexpr int PP = (int)sizeof(T);       // 16-byte pieces per tile
  constexpr int EP = 16 / (int)sizeof(T);  // elements per piece
  typedef T TP __attribute__((ext_vector_type(EP), may_alias));
  typedef __attribute__((address_space(3))) const TP lds_TP;
  const unsigned base = lds_col + col_tile_base<T>(tile);
#pragma unroll
  for (int c = 0; c < PP; ++c) {
    const TP w = *(lds_TP*)(size_t)(base + (unsigned)c * 256u);
#pragma unroll
    for (int k = 0; k < EP; ++k) v[EP * c + k] = w[k];
  }
}

// One point from the table in global memory (points that are not in this workgroup's cell):
// same tree as reduce_planes_dma, plane (k2, k3) at k2 * stride2 + k3 * stride3.
template <typename T, bool FMA>
__device__ __noinline__ T col_slow_point(__amdgpu_buffer_rsrc_t rsrc, unsigned tile_off_bytes, unsigned ps2_bytes, unsigned ps3_bytes,
                                         const CubicDimRegular<T>* dim) {
  T s3[4];
#pragma unroll 1
  for (int k3 = 0; k3 < 4; ++k3) {
    T s2[4];
#pragma unroll 1
    for (int k2 = 0; k2 < 4; ++k2) {
      T v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e)
        v[e] = table_load<T>(rsrc, tile_off_bytes + (unsigned)k2 * ps2_bytes + (unsigned)k3 * ps3_bytes + (unsigned)e * (unsigned)sizeof(T), 0u);
      s2[k2] = reduce_tile<T, false, FMA>(v, dim, 0u);
    }
    s3[k3] = cubic_regular_node<FMA, T>(s2[0], s2[1], s2[2], s2[3], dim[2]);
  }
  return cubic_regular_node<FMA, T>(s3[0], s3[1], s3[2], s3[3], dim[3]);
}

template <typename T, bool FMA>
__global__ void __launch_bounds__(kColThreads) k_cubic_column(const CubicColumnArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_col[];
  __shared__ int s_bin;
  __shared__ unsigned s_begin, s_end;
  const unsigned tid = threadIdx.x;
  // Which (bin, part) is this workgroup?  part_prefix is non-decreasing; [nbins] = number of parts.
  if (tid == 0) {
    const unsigned w = blockIdx.x;
    int bin = -1;
    if (w < a.part_prefix[a.nbins]) {
      int lo = 0, hi = a.nbins;  // largest b with part_prefix[b] <= w
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.part_prefix[mid] <= w) lo = mid; else hi = mid;
      }
      bin = lo;
      const unsigned b0 = bin ? a.bin_end[bin - 1] : 0u;
      const unsigned b1 = a.bin_end[bin];
      const unsigned count = b1 - b0;
      const unsigned nparts = a.part_prefix[bin + 1] - a.part_prefix[bin];
      const unsigned j = w - a.part_prefix[bin];
      // equal parts, each a multiple of 64 points except the last
      unsigned per = (count + nparts - 1) / nparts;
      per = (per + 63u) & ~63u;
      const unsigned lo_p = b0 + j * per;
      unsigned hi_p = lo_p + per;
      if (hi_p > b1) hi_p = b1;
      s_begin = lo_p < b1 ? lo_p : b1;
      s_end = hi_p;
    }
    s_bin = bin;
  }
  __syncthreads();
  const int bin = s_bin;
  if (bin < 0) return;
  const unsigned begin = s_begin, end = s_end;
  if (begin >= end) return;
  const unsigned key = (unsigned)(((unsigned long long)(unsigned)bin * a.inv_mult) % (unsigned)a.nbins);
  const int ci = (int)(key / (unsigned)a.nb1), cj = (int)(key % (unsigned)a.nb1);
  const unsigned ntiles = (unsigned)a.n[2] * (unsigned)a.n[3];
  const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(a.tiles, a.table_bytes);
  const unsigned cell_off = (unsigned)(ci * (int)a.nbj + cj) * 16u * (unsigned)sizeof(T);  // my cell's tile inside a plane, bytes
  const unsigned ps2 = a.plane_stride[2] * (unsigned)sizeof(T), ps3 = a.plane_stride[3] * (unsigned)sizeof(T);
  const unsigned lds_col = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_col;
  // ---- column fill: one LDS-DMA instruction = 1 KiB = RPI 256-byte rows of one 16-tile group;
  // lane L delivers piece (row0 + (L >> 4)) of tile 16 g + (L & 15).
  {
    constexpr unsigned PP = (unsigned)sizeof(T);  // pieces (rows) per group
    constexpr unsigned IPG = PP / 4u;             // DMA instructions per group (4 rows each)
    const unsigned ngroups = (ntiles + 15u) / 16u;
    const unsigned ninstr = ngroups * IPG;
    const unsigned wave = tid >> 6, wl = tid & 63u;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    for (unsigned q = wave; q < ninstr; q += kColThreads / 64) {
      const unsigned g = q / IPG, r0 = (q % IPG) * 4u;
      const unsigned tile = g * 16u + (wl & 15u);
      const unsigned piece = r0 + (wl >> 4);
      unsigned src = 0xFFFFFFF0u;  // out of range: the descriptor's check turns it into zeros
      if (tile < ntiles) {
        const unsigned k = tile / (unsigned)a.n[3], l = tile - k * (unsigned)a.n[3];
        src = k * ps2 + l * ps3 + cell_off + piece * 16u;
      }
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_col + g * (16u * col_tile_bytes<T>()) + r0 * 256u));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_byte*)(size_t)dst, 16, src, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // ---- the part's points
  for (unsigned p0 = begin; p0 < end; p0 += kColThreads) {
    const unsigned p = p0 + tid;
    const bool live = p < end;
    CubicDimRegular<T> dim[4];
    int loc[4];
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const T x = live ? stream_load(a.obs[d] + p) : a.start[d];
      T floc;
      ok &= regular_floc<T>(x, a.start[d], a.step[d], &floc);   // multicubic/regular.rs:435-438
      ok &= floc != (T)-9223372036854775808.0;                  // `- 1` would overflow isize
      const T nn = (T)a.n[d];
      const int l = clamp_loc<T>(floc - (T)1, a.n[d] - 4);      // regular.rs:440-442
      int sat;
      bool outside;
      if (floc < (T)0) { sat = kSatLow; outside = true; }       // regular.rs:445-466 on floc = iloc + 1
      else if (floc == (T)0) { sat = kSatLow; outside = false; }
      else if (floc > nn - (T)2) { sat = kSatHigh; outside = true; }
      else if (floc == nn - (T)2) { sat = kSatHigh; outside = false; }
      else { sat = kSatNone; outside = false; }
      const T index_one_loc = mul_add<false>(a.step[d], (T)(l + 1), a.start[d]);  // regular.rs:356-360, never fused
      const T t = (x - index_one_loc) / a.step[d];
      dim[d].sat = sat;
      dim[d].linear = (outside && a.linearize) ? 1 : 0;
      dim[d].tt = sat == kSatLow ? -t : (sat == kSatHigh ? t - (T)1 : t);
      loc[d] = l;
    }
    // the original index is needed for a scattered store and for reporting a failing point
    const unsigned orig = (live && (!a.res_sorted || !ok)) ? a.index[p] : 0u;
    if (!ok && live) atomicMin(a.first_bad, (unsigned long long)(a.index_base + orig));
    unsigned interior = 0;
#pragma unroll
    for (int d = 0; d < 2; ++d)
      if (__builtin_amdgcn_ballot_w64(dim[d].sat != kSatNone || dim[d].linear != 0) == 0) interior |= 1u << d;
    // 16 planes out of LDS: dim 2 index = k & 3, dim 3 index = k >> 2 (the reference's order)
    const unsigned t0 = (unsigned)loc[2] * (unsigned)a.n[3] + (unsigned)loc[3];
    T s2[4], s3[4];
    T res = (T)0;
    // One tile in registers at a time: with four waves per SIMD the other waves' arithmetic covers
    // this wave's LDS latency (a second tile buffer costs 32 VGPRs, and at the 128 a 1024-thread
    // workgroup may use, spills).
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      T cur[16];
      col_take_tile<T>(lds_col, t0 + (unsigned)(k & 3) * (unsigned)a.n[3] + (unsigned)(k >> 2), cur);
      const T r01 = reduce_tile<T, false, FMA>(cur, dim, interior);
      s2[k & 3] = r01;
      if ((k & 3) == 3) {
        s3[k >> 2] = cubic_regular_node<FMA, T>(s2[0], s2[1], s2[2], s2[3], dim[2]);
        if (k == 15) res = cubic_regular_node<FMA, T>(s3[0], s3[1], s3[2], s3[3], dim[3]);
      }
    }
    // not my cell (the sort's estimate and the exact cell disagree on a boundary): from the table
    if (live && (loc[0] != ci || loc[1] != cj)) {
      const unsigned toff = ((unsigned)loc[2] * a.plane_stride[2] + (unsigned)loc[3] * a.plane_stride[3] +
                             (unsigned)(loc[0] * (int)a.nbj + loc[1]) * 16u) * (unsigned)sizeof(T);
      // a COPY goes to the out-of-line routine: taking the address of `dim` itself would keep it
      // in scratch memory for every point
      CubicDimRegular<T> dcopy[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) dcopy[d] = dim[d];
      res = col_slow_point<T, FMA>(rsrc, toff, ps2, ps3, dcopy);
    }
    if (live) {
      if (a.res_sorted) a.res_sorted[p] = res;
      else stream_store(a.out + orig, res);
    }
  }
}

// out[i] = res[rank[i]]: the un-permutation of results written in sorted order.  `rank` is read
// coalesced, `res` in short runs (the points of one 4096-point chunk of the sort sit in one run
// per bin), `out` is written coalesced — full lines instead of 8-byte scattered stores.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_unpermute(const T* __restrict__ res, const unsigned* __restrict__ rank, T* __restrict__ out, size_t npts) {
  const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < npts) stream_store(out + i, res[rank[i]]);
}

}  // namespace interpn
