// 4-D multicubic on SORTED points with (part of) the table column of a cell resident in LDS.
//
// After the counting sort of k_bin_points.hip with one bin per (i, j) cell of dims 0 and 1, all
// points of a bin read the SAME 4 x 4 (i, j) footprint of every (k, l) plane: n2 x n3 tiles of the
// fully overlapped tile table (cubic_brick.h), 128 B each in f64 — 128 KiB for cfg4's 32 x 32
// planes.  A workgroup takes one part of a bin (<= 16 points per thread), sorts it locally by the
// class pair of dims 2, 3 and evaluates it out of LDS: a point still reads its 16 tiles = 2 KiB,
// but from the CU's own LDS (256 B/clk) instead of 16 L2 lines.
//
// Round 4: the column is resident only a K-RANGE AT A TIME.  The part's local order is by the
// class of dim 2 first, so the points whose dim-2 class lies in [r * cpp, (r + 1) * cpp) are one
// contiguous stretch of it ("phase" r) and need only the tile rows k = loc2 .. loc2 + 3 of those
// classes: cpp + 3 rows of n3 tiles.  With 13-row sub-columns (52 KiB for cfg4) TWO workgroups fit
// a CU, and the set-up of one (part lookup, record loads, local sort, LDS-DMA fill: memory
// latency, no arithmetic) runs under the plane arithmetic of the other; with the whole column
// (round 3: 148 KiB, one workgroup per CU) those phases were serial on every CU.  Columns larger
// than the LDS (48^4: 288 KiB) are simply more phases.  A wave whose 64 slots of a phase are all
// beyond its end skips the row (no idle arithmetic in the tail rows), and the record of the next
// row is requested before this row's planes are evaluated.
//
// LDS layout: tiles are interleaved sixteen at a time — piece c (16 bytes) of tile T sits at
//   (T >> 4) * (16 * TILE) + c * 256 + (T & 15) * 16,          TILE = 16 * sizeof(T) bytes
// (T counted from the first row of the phase) so the eight pieces of a tile are reached from ONE
// address register with immediate offsets c * 256; lanes of a wave mostly read the SAME tile after
// the local sort (broadcast), so the layout never conflicts in a fixed pattern.
//
// Arithmetic, plane order and reduction tree are those of cubic_brick.h / the reference
// (src/multicubic/regular.rs:325-623): bit-identical results.  A point whose exact cell is not
// the workgroup's, or whose exact dim-2 rows are not in the phase's sub-column (the sorts estimate
// classes by multiplying with a reciprocal, the kernel divides like the reference; they can
// disagree on a cell boundary) is evaluated from the table in global memory by the same tree.
#pragma once
#include "cubic_brick.h"

namespace interpn {

template <typename T>
struct CubicColumnArgs {
  const T* tiles;            // fully overlapped tile table [plane (k, l)][bi][bj][16]
  unsigned table_bytes;
  const T* records;          // the slice's points in bin order, 4 coordinates each
  const unsigned* index;     // sorted position -> index within the slice
  T* out;                    // caller's output of the slice
  unsigned long long* first_bad;
  size_t index_base;
  size_t npts;
  const unsigned* bin_end;      // end of bin b in sorted order (the scatter's cursors after the scatter)
  const unsigned* part_prefix;  // workgroups (parts) in front of bin b; [nbins] = total
  int nbins;
  int nb1;                   // classes along dim 1 (n1 - 1)
  unsigned inv_mult;         // sorted bin b holds class pair (b * inv_mult) % nbins
  T start[4];
  T step[4];
  T rstep[4];                // ~1 / step: the local sort's class estimate (a hint; the exact class is re-derived)
  int n[4];
  int linearize;
  unsigned plane_stride[4];  // table elements per unit index of dims 2, 3
  unsigned nbj;
  // K-range phases (see the head of this file)
  int cpp;                   // classes of dim 2 per phase (>= 1); a phase's sub-column has at most cpp + 3 tile rows
  int nphase;                // ceil((n2 - 1) / cpp)
  int q3;                    // local sort key = class of dim 2 * q3 + (class of dim 3 >> sh3); (n2 - 1) * q3 <= 1024
  int sh3;
  unsigned sub_bytes;        // LDS bytes reserved for the sub-column; the part's local order (16-bit) sits behind them
  unsigned long long* stamps;  // measurement aid (option debug_stamps): 8 words per workgroup, or null
};

constexpr int kColPerThread = 16;  // points of a part per thread at most (register arrays of the local sort)


template <typename T> constexpr unsigned col_tile_bytes() { return 16u * (unsigned)sizeof(T); }
// LDS bytes of a column of `ntiles` tiles (whole 16-tile groups)
template <typename T> __host__ __device__ inline size_t col_lds_bytes(unsigned ntiles) { return (size_t)((ntiles + 15u) / 16u) * 16u * col_tile_bytes<T>(); }

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

// ---- nodes with a wave-uniform form -------------------------------------------------------------
// Within a workgroup all points share the class of dims 0 and 1, and the local sort below puts
// points of one (dim 2, dim 3) class pair next to each other, so almost every wave is uniform
// along every dimension: all lanes interior (the reference's Saturation::None arm), all saturated
// low, or all saturated high, none of them extrapolating linearly.  Each of these is the
// reference's arm for that case without the selects of the general form (same operations, same
// bits, multicubic/regular.rs:495-623); anything else takes the general form.
enum : int { kFormNone = 0, kFormLow = 1, kFormHigh = 2, kFormMixed = 3 };

template <bool FMA, int FORM, typename T>
__device__ __forceinline__ T col_node(T v0, T v1, T v2, T v3, const CubicDimRegular<T>& d) {
  const T two = (T)2;
  if constexpr (FORM == kFormNone) {
    return cubic_regular_node_interior<FMA, T>(v0, v1, v2, v3, d.tt);
  } else if constexpr (FORM == kFormLow) {   // InsideLow / OutsideLow without linearisation: regular.rs:507-527
    const T dy = v0 - v1;
    const T k0 = -((v2 - v0) / two);
    const T k1 = two * dy - k0;              // == two.mul_add(dy, -k0): 2 dy is exact
    return hermite<FMA>(d.tt, v1, dy, k0, k1);
  } else if constexpr (FORM == kFormHigh) {  // InsideHigh / OutsideHigh without linearisation: regular.rs:563-582
    const T dy = v3 - v2;
    const T k0 = (v3 - v1) / two;
    const T k1 = two * dy - k0;
    return hermite<FMA>(d.tt, v2, dy, k0, k1);
  } else {
    return cubic_regular_node<FMA, T>(v0, v1, v2, v3, d);
  }
}

template <bool FMA, typename T>
__device__ __forceinline__ T col_node_rt(int form, T v0, T v1, T v2, T v3, const CubicDimRegular<T>& d) {
  switch (form) {  // wave-uniform
    case kFormNone: return col_node<FMA, kFormNone, T>(v0, v1, v2, v3, d);
    case kFormLow: return col_node<FMA, kFormLow, T>(v0, v1, v2, v3, d);
    case kFormHigh: return col_node<FMA, kFormHigh, T>(v0, v1, v2, v3, d);
    default: return col_node<FMA, kFormMixed, T>(v0, v1, v2, v3, d);
  }
}

// the wave's form along one dimension (every lane must call this)
template <typename T>
__device__ __forceinline__ int col_wave_form(const CubicDimRegular<T>& d) {
  if (__builtin_amdgcn_ballot_w64(d.linear != 0) != 0) return kFormMixed;
  if (__builtin_amdgcn_ballot_w64(d.sat != kSatNone) == 0) return kFormNone;
  if (__builtin_amdgcn_ballot_w64(d.sat != kSatLow) == 0) return kFormLow;
  if (__builtin_amdgcn_ballot_w64(d.sat != kSatHigh) == 0) return kFormHigh;
  return kFormMixed;
}

// All 16 planes of a point out of the LDS column, dim 0 in form F0: dim 2 index = k & 3, dim 3
// index = k >> 2 (the reference's order, multicubic/regular.rs:368-421).  The tile of the next plane
// is requested before this plane's nodes are evaluated (PIPE: needs 32 more VGPRs).
// ALLNONE: every dimension of the wave is interior (the common case): no form tests at all.
template <typename T, bool FMA, int F0, bool PIPE, bool ALLNONE = false>
__device__ __forceinline__ T col_reduce(unsigned lds_col, unsigned t0, unsigned n3, const CubicDimRegular<T>* dim, int f1, int f2, int f3) {
  T s3[4];
  T cur[16];
  if constexpr (PIPE) col_take_tile<T>(lds_col, t0, cur);
#pragma unroll 1
  for (int k3 = 0; k3 < 4; ++k3) {
    T s2[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
      T nxt[16];
      if constexpr (PIPE) {
        // plane after (k2, k3): (k2 + 1, k3), or (0, k3 + 1); behind the last plane this reads one
        // tile too many (t0 + 4: still inside the column)
        const unsigned tn = k2 < 3 ? t0 + (unsigned)(k2 + 1) * n3 + (unsigned)k3 : t0 + (unsigned)(k3 + 1);
        col_take_tile<T>(lds_col, tn, nxt);
      } else {
        col_take_tile<T>(lds_col, t0 + (unsigned)k2 * n3 + (unsigned)k3, cur);
      }
      T w[4];
#pragma unroll
      for (int ej = 0; ej < 4; ++ej) w[ej] = col_node<FMA, F0, T>(cur[ej], cur[4 + ej], cur[8 + ej], cur[12 + ej], dim[0]);
      if constexpr (ALLNONE) s2[k2] = col_node<FMA, kFormNone, T>(w[0], w[1], w[2], w[3], dim[1]);
      else s2[k2] = col_node_rt<FMA, T>(f1, w[0], w[1], w[2], w[3], dim[1]);
      if constexpr (PIPE) {
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] = nxt[e];
      }
    }
    if constexpr (ALLNONE) s3[k3] = col_node<FMA, kFormNone, T>(s2[0], s2[1], s2[2], s2[3], dim[2]);
    else s3[k3] = col_node_rt<FMA, T>(f2, s2[0], s2[1], s2[2], s2[3], dim[2]);
  }
  if constexpr (ALLNONE) return col_node<FMA, kFormNone, T>(s3[0], s3[1], s3[2], s3[3], dim[3]);
  else return col_node_rt<FMA, T>(f3, s3[0], s3[1], s3[2], s3[3], dim[3]);
}

// class estimate of the local sort (a hint only): 0 = floc <= 0, c = floc, n - 2 = floc >= n - 2
template <typename T>
__device__ __forceinline__ unsigned col_class_hint(T x, T start, T rstep, int n) {
  const T u = (x - start) * rstep;
  return u >= (T)1 ? (u < (T)(n - 2) ? (unsigned)(int)u : (unsigned)(n - 2)) : 0u;
}

// Registers: 12 waves per CU = three per SIMD (168 VGPRs) whether they come as two 384-thread
// workgroups, three of 256 or one of 768; 512-thread workgroups pair up at four per SIMD (128).
template <typename T, bool FMA, int THREADS>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 512 ? 4 : 3)))
k_cubic_column(const CubicColumnArgs<T> a) {
  constexpr bool PIPE = THREADS <= 768;  // 168+ VGPRs per lane: room for a second tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_col[];
  __shared__ int s_bin;
  __shared__ unsigned s_begin, s_end;
  __shared__ unsigned s_hist[1024];   // local sort: points per key, then (after pass 2) the END of every key's stretch
  __shared__ unsigned s_wave[THREADS / 64];
  __shared__ unsigned s_row;          // next 64-slot row of the current phase (rows are dealt to the waves as they free up)
  const unsigned tid = threadIdx.x;
  const unsigned wave = tid >> 6, wl = tid & 63u;
  unsigned long long t_stamp[6] = {0, 0, 0, 0, 0, 0};
  if (a.stamps && tid == 0) t_stamp[0] = wall_clock64();
  // Which (bin, part) is this workgroup?  part_prefix is non-decreasing, [nbins] = number of parts:
  // the bin b with part_prefix[b] <= w < part_prefix[b + 1] — every thread tests bins (a single
  // thread bisecting costs ten dependent global loads, ~10 us).
  if (tid == 0) { s_bin = -1; s_row = 0; }
  for (unsigned c = tid; c < 1024u; c += THREADS) s_hist[c] = 0;
  __syncthreads();
  {
    const unsigned w = blockIdx.x;
    for (int b = (int)tid; b < a.nbins; b += THREADS) {
      const unsigned p0 = a.part_prefix[b], p1 = a.part_prefix[b + 1];
      if (p0 <= w && w < p1) {
        const unsigned b0 = b ? a.bin_end[b - 1] : 0u;
        const unsigned b1 = a.bin_end[b];
        const unsigned cnt = b1 - b0;
        const unsigned nparts = p1 - p0;
        const unsigned j = w - p0;
        const unsigned per = (cnt + nparts - 1) / nparts;  // equal parts (<= the scan's part_points)
        const unsigned lo_p = b0 + j * per;
        unsigned hi_p = lo_p + per;
        if (hi_p > b1) hi_p = b1;
        s_begin = lo_p < b1 ? lo_p : b1;
        s_end = hi_p;
        s_bin = b;
      }
    }
  }
  __syncthreads();
  const int bin = s_bin;
  if (bin < 0) return;
  const unsigned begin = s_begin, end = s_end;
  if (begin >= end) return;
  const unsigned count = end - begin;  // <= kColPerThread * THREADS (the scan cut the bin accordingly)
  if (a.stamps && tid == 0) t_stamp[1] = wall_clock64();
  const unsigned key = (unsigned)(((unsigned long long)(unsigned)bin * a.inv_mult) % (unsigned)a.nbins);
  const int c0 = (int)(key / (unsigned)a.nb1), c1 = (int)(key % (unsigned)a.nb1);  // nominal classes of dims 0, 1
  const int ci = c0 - 1 < 0 ? 0 : (c0 - 1 > a.n[0] - 4 ? a.n[0] - 4 : c0 - 1);      // their footprint cell
  const int cj = c1 - 1 < 0 ? 0 : (c1 - 1 > a.n[1] - 4 ? a.n[1] - 4 : c1 - 1);
  const unsigned n3 = (unsigned)a.n[3];
  const int ncls2 = a.n[2] - 1;
  const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(a.tiles, a.table_bytes);
  const unsigned cell_off = (unsigned)(ci * (int)a.nbj + cj) * 16u * (unsigned)sizeof(T);  // my cell's tile inside a plane, bytes
  const unsigned ps2 = a.plane_stride[2] * (unsigned)sizeof(T), ps3 = a.plane_stride[3] * (unsigned)sizeof(T);
  const unsigned lds_col = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_col;
  unsigned short* perm = reinterpret_cast<unsigned short*>(smem_col + a.sub_bytes);  // local order: slot -> point of the part
  typedef T RV __attribute__((ext_vector_type(4)));
  const RV* __restrict__ recs = reinterpret_cast<const RV*>(a.records) + begin;
  const unsigned* __restrict__ index = a.index + begin;

  // tile rows of phase r: the footprints loc2 .. loc2 + 3 of the dim-2 classes [r cpp, (r + 1) cpp)
  auto phase_rows = [&](int r, unsigned* row0, unsigned* nrows) {
    const int c_lo = r * a.cpp;
    int c_hi = c_lo + a.cpp;
    c_hi = (c_hi < ncls2 ? c_hi : ncls2) - 1;
    const int top = a.n[2] - 4;
    const int l_lo = c_lo - 1 < 0 ? 0 : (c_lo - 1 > top ? top : c_lo - 1);
    const int l_hi = c_hi - 1 < 0 ? 0 : (c_hi - 1 > top ? top : c_hi - 1);
    *row0 = (unsigned)l_lo;
    *nrows = (unsigned)(l_hi + 4 - l_lo);
  };
  // sub-column fill: one LDS-DMA instruction = 1 KiB = four 256-byte rows of one 16-tile group;
  // lane L delivers piece (r0 + (L >> 4)) of tile 16 g + (L & 15).
  auto fill = [&](unsigned row0, unsigned nrows) {
    constexpr unsigned PP = (unsigned)sizeof(T);  // pieces (256-byte LDS rows) per group
    constexpr unsigned IPG = PP / 4u;             // DMA instructions per group (4 rows each)
    const unsigned ntiles = nrows * n3;
    const unsigned ninstr = ((ntiles + 15u) / 16u) * IPG;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    for (unsigned q = wave; q < ninstr; q += THREADS / 64) {
      const unsigned g = q / IPG, r0 = (q % IPG) * 4u;
      const unsigned tile = g * 16u + (wl & 15u);
      const unsigned piece = r0 + (wl >> 4);
      unsigned src = 0xFFFFFFF0u;  // out of range: the descriptor's check turns it into zeros
      if (tile < ntiles) {
        const unsigned k = tile / n3, l = tile - k * n3;
        src = (row0 + k) * ps2 + l * ps3 + cell_off + piece * 16u;
      }
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_col + g * (16u * col_tile_bytes<T>()) + r0 * 256u));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_byte*)(size_t)dst, 16, src, 0, 0, 0);
    }
  };

  // ---- local sort, pass 1: the (dim 2, dim 3) class pair of each of my points -> histogram.
  // (Loads first, then the first phase's fill is issued, then they are used: the fill overlaps this.)
  unsigned short cls23[kColPerThread];
  {
    T x2[kColPerThread], x3[kColPerThread];
#pragma unroll
    for (int m = 0; m < kColPerThread; ++m) {
      const unsigned q = (unsigned)m * THREADS + tid;
      const RV r = q < count ? recs[q] : recs[0];
      x2[m] = r[2];
      x3[m] = r[3];
    }
    {
      unsigned row0, nrows;
      phase_rows(0, &row0, &nrows);
      fill(row0, nrows);
    }
#pragma unroll
    for (int m = 0; m < kColPerThread; ++m) {
      const unsigned q = (unsigned)m * THREADS + tid;
      const unsigned h2 = col_class_hint<T>(x2[m], a.start[2], a.rstep[2], a.n[2]);
      const unsigned h3 = col_class_hint<T>(x3[m], a.start[3], a.rstep[3], a.n[3]);
      const unsigned c = h2 * (unsigned)a.q3 + (h3 >> a.sh3);  // < (n2 - 1) q3 <= 1024
      cls23[m] = (unsigned short)c;
      if (q < count) atomicAdd(&s_hist[c], 1u);
    }
  }
  __syncthreads();
  if (a.stamps && tid == 0) t_stamp[2] = wall_clock64();
  // exclusive scan of the 1024 counters: CPT consecutive counters per thread, wave scan, wave totals
  {
    constexpr int CPT = (1024 + THREADS - 1) / THREADS;
    unsigned mine[CPT];
    unsigned sum = 0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) { mine[c] = tid * CPT + c < 1024u ? s_hist[tid * CPT + c] : 0u; sum += mine[c]; }
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned up = (unsigned)__shfl_up((int)incl, off);
      if (wl >= (unsigned)off) incl += up;
    }
    if (wl == 63u) s_wave[wave] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (unsigned w = 0; w < wave; ++w) run += s_wave[w];
#pragma unroll
    for (int c = 0; c < CPT; ++c) { if (tid * CPT + c < 1024u) s_hist[tid * CPT + c] = run; run += mine[c]; }
  }
  __syncthreads();
  // pass 2: slots (each counter ends up at the END of its key's stretch = the start of the next key's)
#pragma unroll
  for (int m = 0; m < kColPerThread; ++m) {
    const unsigned q = (unsigned)m * THREADS + tid;
    if (q < count) perm[atomicAdd(&s_hist[cls23[m]], 1u)] = (unsigned short)q;
  }
  __syncthreads();
  if (a.stamps && tid == 0) t_stamp[3] = wall_clock64();

  // ---- the phases: K-range r of the column in LDS, the stretch of the local order that needs it
  for (int r = 0; r < a.nphase; ++r) {
    const unsigned klo = (unsigned)(r * a.cpp) * (unsigned)a.q3;
    int chi = (r + 1) * a.cpp;
    chi = chi < ncls2 ? chi : ncls2;
    const unsigned khi = (unsigned)chi * (unsigned)a.q3;
    const unsigned ps = klo ? s_hist[klo - 1] : 0u, pe = s_hist[khi - 1];  // workgroup-uniform
    unsigned row0, nrows;
    phase_rows(r, &row0, &nrows);
    if (r > 0) {
      if (pe == ps) continue;
      __syncthreads();  // every wave has finished with the previous sub-column
      if (tid == 0) s_row = 0;
      fill(row0, nrows);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my share of the sub-column has landed
    __syncthreads();
    if (a.stamps && tid == 0 && r == 0) t_stamp[4] = wall_clock64();
    if (pe == ps) continue;
    const int row_top = (int)nrows - 4;  // largest footprint row inside the sub-column

    // Rows of 64 slots are handed to the waves as they free up (a shared counter: rows differ in
    // cost — node forms, out-of-cell points — and the waves of a SIMD share its issue slots, so a
    // fixed deal left waves idle at the end of every phase); a wave that draws a row beyond the
    // stretch is done with the phase.
    auto draw_row = [&]() -> unsigned {
      unsigned v = 0;
      if (wl == 0) v = atomicAdd(&s_row, 1u);
      return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
    };
    unsigned jw = ps + draw_row() * 64u;
    if (jw >= pe) continue;
    // dead lanes (the stretch's tail) redo its last point: they keep their wave uniform
    unsigned q = perm[jw + wl < pe ? jw + wl : pe - 1];
    RV rec = recs[q];
    for (;;) {
      const bool live = jw + wl < pe;
      const unsigned orig = index[q];  // used at the very end: its latency hides behind the planes
      const RV rcur = rec;
      const unsigned jn = ps + draw_row() * 64u;
      if (jn < pe) {  // wave-uniform: next row's record on its way while this row's planes are evaluated
        q = perm[jn + wl < pe ? jn + wl : pe - 1];
        rec = recs[q];
      }
      CubicDimRegular<T> dim[4];
      int loc[4];
      bool ok = true;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const T x = rcur[d];
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
      const int f0 = col_wave_form<T>(dim[0]), f1 = col_wave_form<T>(dim[1]);
      const int f2 = col_wave_form<T>(dim[2]), f3 = col_wave_form<T>(dim[3]);
      // footprint rows relative to the sub-column; a point whose exact rows are not all in it is
      // evaluated from the table below (its LDS reads stay inside the sub-column, result dropped)
      const int rel2 = loc[2] - (int)row0;
      const bool in_rows = rel2 >= 0 && rel2 <= row_top;
      const unsigned t0 = (unsigned)(in_rows ? rel2 : 0) * n3 + (unsigned)loc[3];
      T res;
      if ((f0 | f1 | f2 | f3) == kFormNone) res = col_reduce<T, FMA, kFormNone, PIPE, true>(lds_col, t0, n3, dim, f1, f2, f3);
      else
      switch (f0) {  // wave-uniform
        case kFormNone: res = col_reduce<T, FMA, kFormNone, PIPE>(lds_col, t0, n3, dim, f1, f2, f3); break;
        case kFormLow: res = col_reduce<T, FMA, kFormLow, PIPE>(lds_col, t0, n3, dim, f1, f2, f3); break;
        case kFormHigh: res = col_reduce<T, FMA, kFormHigh, PIPE>(lds_col, t0, n3, dim, f1, f2, f3); break;
        default: res = col_reduce<T, FMA, kFormMixed, PIPE>(lds_col, t0, n3, dim, f1, f2, f3); break;
      }
      // not my cell / not my rows (the sorts' estimates and the exact cell disagree on a boundary): from the table
      if (live && (loc[0] != ci || loc[1] != cj || !in_rows)) {
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
        if (!ok) atomicMin(a.first_bad, (unsigned long long)(a.index_base + orig));
        stream_store(a.out + orig, res);
      }
      if (jn >= pe) break;
      jw = jn;
    }
  }
  if (a.stamps) {
    // [0..4] thread 0's stamps, [5] the LAST wave's end, [6] ids, [7] count | thread 0's own duration (ticks, 16 bits) | bin
    unsigned long long* w = a.stamps + (size_t)blockIdx.x * 8u;
    const unsigned long long t_end = wall_clock64();
    if (wl == 0) atomicMax(&w[5], t_end);
    if (tid == 0) {
      for (int k = 0; k < 5; ++k) w[k] = t_stamp[k];
      unsigned hw = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned xcc = 0;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      w[6] = ((unsigned long long)xcc << 32) | hw;
      w[7] = ((unsigned long long)count << 32) | ((unsigned long long)((t_end - t_stamp[0]) & 0xFFFFu) << 16) | (unsigned)(bin & 0xFFFF);
    }
  }
}

}  // namespace interpn
