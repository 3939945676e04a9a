// N-D multicubic (N = 2..4; regular and rectilinear; f64 and f32) on a tiled copy of the grid with
// a 16-lane cooperative gather.
//
// A cubic point reads a 4^N footprint.  In C order that is 4^(N-1) rows of 4 elements, each on its
// own 128-B line (76 lines/point in 4-D), and the kernel is bound by the L2 -> L1 line rate
// (DESIGN.md section 4).  Here the handle keeps a copy of the grid in which the FIRST TWO dimensions
// (i, j) — the ones the reference reduces first — are cut into 4 x 4 tiles (128 B in f64), stepped
// 4, 2 or 1 (overlapping tiles = duplicated rows/columns) so that a 4 x 4 footprint spans 3.06,
// 2.25 ... 1 tiles; the remaining dimensions index whole tiled planes.  Two gathers:
//   * fully overlapped tiles (steps 1,1; a footprint = exactly one tile): the 64 tiles a wave needs
//     from a plane go from the table straight into LDS by LDS-DMA, in an image laid out so that
//     every lane then reads its own tile back without bank conflicts; the planes of a point are
//     software-pipelined (gather_plane / reduce_planes_dma below).  This is the form the large-grid
//     and the binned (sorted-points) evaluations run: cfg4 1.04 ms per 1e7 points on sorted points.
//   * other steps: the 16 lanes of a group fetch the 16 elements of ONE point's (i, j) footprint
//     per load instruction (elements on one line become one L2 request), 16 instructions cover the
//     group's 16 points, the 16 x 16 element matrix is transposed through LDS.
// Either way every lane reduces its own footprint: dim 0, then dim 1, then the plane dimensions in
// order — the reference's tree (src/multicubic/regular.rs:368-421) — so results are bit-identical
// to the C-order kernels.  Table reads are raw buffer loads (32-bit byte offset per lane, plane
// offset in the scalar operand): no vector address arithmetic, range-checked.
#pragma once
#include "rect_args.h"


namespace interpn {

template <typename T, int N>
struct CubicBrickArgs {
  const T* bricks;
  unsigned table_bytes;  // < 4 GiB
  const T* obs[N];
  T* out;
  unsigned long long* first_bad;
  size_t npts;
  T start[N];
  T step[N];
  int n[N];
  AxisArgs<T, N> ax;
  const CubicCellRecord<T>* crec[N];  // rectilinear, LDS-DMA form: the axes' per-cell records (nullptr: none; cubic_rect_locate_rec)
  unsigned plane_stride[N];  // d >= 2: table elements per unit index of dim d
  unsigned nbj;
  int linearize;
  // Binned evaluation (k_bin_points.hip): `obs` then holds the points in table order, point k
  // being point scatter[k] of the caller's slice; its result goes to out[scatter[k]] and a failing
  // coordinate reports index_base + scatter[k].  nullptr = points are evaluated in place.
  const unsigned* scatter;
  size_t index_base;
  const unsigned* gate;  // gated launch (GridDesc::launch_gate): null, or a word that must be non-zero for this launch to do anything
  // Binned evaluation, dealing the sorted order out to the XCDs: workgroups with equal
  // blockIdx % 8 (one XCD under the observed round-robin placement; speed only) walk one
  // contiguous eighth of the points, `eighth` points long (a multiple of 256; 0 = off), so that
  // an XCD's L2 only ever holds its own part of the table.
  size_t eighth;
};

constexpr int kCubRow = 18;  // elements per LDS row (16 used; 18 keeps 16-B alignment and spreads banks)

// Which instantiations gather by LDS-DMA (gather_plane_dma), and the LDS bytes a workgroup's
// gathers need (rectilinear axes are staged behind them).
template <typename T, int SI, int SJ> constexpr bool cubic_dma() { return SI == 1 && SJ == 1; }
// A tile is 16 elements = sizeof(T) 16-byte pieces; a wave's image holds its 64 tiles.
template <typename T> constexpr unsigned cubic_dma_image() { return 64u * (unsigned)sizeof(T) * 16u; }
template <typename T, int SI, int SJ> constexpr size_t cubic_lds_region() {
  if (cubic_dma<T, SI, SJ>()) return (size_t)(kBlock / 64) * cubic_dma_image<T>();  // one tile image per wave (8 KiB f64, 4 KiB f32)
  return (size_t)kBlock * kCubRow * (sizeof(T) > 4 ? sizeof(T) : 4);
}
// rotation of a point's pieces inside its slots (conflict-free ds_read_b128, see gather comment)
template <typename T> __device__ __forceinline__ unsigned cubic_dma_rot(unsigned p) {
  return sizeof(T) == 8 ? ((p >> 1) & 7u) : ((p >> 2) & 3u);
}

// Table reads go through a raw buffer descriptor: the per-lane part of the address is a 32-bit
// BYTE offset computed once per point (toff), the plane part (delta) is wave-uniform and rides in
// the instruction's scalar offset, so a gather costs no address arithmetic on the vector unit (the
// flat form spent one 64-bit add per load: 256 of ~2500 VALU instructions of a 4-D point).
// Bounds: on gfx9 raw buffers the descriptor's range check (num_records = table bytes) covers the
// per-lane offset (voffset + instruction offset) ONLY; the wave-uniform plane offset passed as
// soffset is added after the check.  An out-of-range `voff` therefore reads 0, but safety of
// `voff + soff` rests on the offsets being right by construction (both are products of clamped
// cell indices and the tile geometry the table was built with; dead lanes carry the offsets of a
// valid point).  Builds with -DINTERPN_HIP_DEBUG_BOUNDS fold soff into the checked voff, so that a
// layout bug reads zeros (and fails the parity tests) instead of a neighbouring allocation.
// Tables are kept below 4 GiB (abi_layout.hip::maybe_build_cubic_tiles).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t table_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <typename T>
__device__ __forceinline__ T table_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
#ifdef INTERPN_HIP_DEBUG_BOUNDS
  voff += soff;
  soff = 0;
#endif
  if constexpr (sizeof(T) == 8) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 raw = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    T v;
    __builtin_memcpy(&v, &raw, 8);
    return v;
  } else {
    const unsigned raw = __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0);
    T v;
    __builtin_memcpy(&v, &raw, 4);
    return v;
  }
}

template <int S>
__device__ __forceinline__ void tile_coord(int i0, int e, int* b, int* o) {
  if (S == 1) { *b = i0; *o = e; return; }
  int bb = S == 4 ? (i0 >> 2) : (i0 >> 1);
  int oo = (i0 - bb * S) + e;
  if (oo >= 4) { bb += 1; oo -= S; }
  *b = bb;
  *o = oo;
}

template <typename T, bool RECT> struct CubicDimSel;
template <typename T> struct CubicDimSel<T, false> { typedef CubicDimRegular<T> type; };
template <typename T> struct CubicDimSel<T, true> { typedef CubicDimRect<T> type; };

template <bool RECT, bool FMA, typename T>
__device__ __forceinline__ T cubic_node_sel(T v0, T v1, T v2, T v3, const typename CubicDimSel<T, RECT>::type& d) {
  if constexpr (RECT) return cubic_rect_node<FMA, T>(v0, v1, v2, v3, d);
  else return cubic_regular_node<FMA, T>(v0, v1, v2, v3, d);
}
// FAST (rectilinear grids): the node's divisions without divide sequences while `ok` holds (cubic_rect_node_fast)
template <bool RECT, bool FMA, bool FAST, typename T>
__device__ __forceinline__ T cubic_node_sel(T v0, T v1, T v2, T v3, const typename CubicDimSel<T, RECT>::type& d, bool& ok) {
  if constexpr (RECT && FAST) return cubic_rect_node_fast<FMA, T>(v0, v1, v2, v3, d, ok);
  else return cubic_node_sel<RECT, FMA, T>(v0, v1, v2, v3, d);
}

// Gather one (i, j) footprint plane at table offset `delta` for all lanes, reduce dims 0 and 1.
template <typename T, bool RECT, bool FMA, bool FAST = false>
__device__ __forceinline__ T gather_plane(__amdgpu_buffer_rsrc_t bricks, const unsigned* toff, unsigned delta, T __attribute__((may_alias))* lds_data,
                                          unsigned group, unsigned me, const typename CubicDimSel<T, RECT>::type* dim,
                                          unsigned interior, bool* okp = nullptr) {
  T val[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) val[r] = table_load<T>(bricks, toff[r], delta);  // byte offsets
#pragma unroll
  for (int r = 0; r < 16; ++r) lds_data[(group * 16 + r) * kCubRow + me] = val[r];
  wave_sync();
  T v[16];
  const T __attribute__((may_alias))* row = lds_data + (group * 16 + me) * kCubRow;
#pragma unroll
  for (int e = 0; e < 16; ++e) v[e] = row[e];
  wave_sync();
  // element e = ei*4 + ej; reduce dim 0 (i) for every j, then dim 1 (j)
  T w[4];
  if constexpr (!RECT) {
    // wave-uniform: every lane interior along the dimension -> the select-free node (same bits)
    if (interior & 1u) {
      cubic_tile_dim0_interior<FMA, T>(v, dim[0].tt, w);  // f32: two nodes per packed instruction
    } else {
#pragma unroll
      for (int ej = 0; ej < 4; ++ej) w[ej] = cubic_regular_node<FMA, T>(v[ej], v[4 + ej], v[8 + ej], v[12 + ej], dim[0]);
    }
    if (interior & 2u) return cubic_regular_node_interior<FMA, T>(w[0], w[1], w[2], w[3], dim[1].tt);
    return cubic_regular_node<FMA, T>(w[0], w[1], w[2], w[3], dim[1]);
  } else {
    bool ok = FAST ? *okp : true;
#pragma unroll
    for (int ej = 0; ej < 4; ++ej) w[ej] = cubic_node_sel<RECT, FMA, FAST, T>(v[ej], v[4 + ej], v[8 + ej], v[12 + ej], dim[0], ok);
    const T r = cubic_node_sel<RECT, FMA, FAST, T>(w[0], w[1], w[2], w[3], dim[1], ok);
    if constexpr (FAST) *okp = ok;
    return r;
  }
}

// Fully overlapped tiles (steps 1,1: a footprint is exactly ONE tile of 16 elements = PP = sizeof(T)
// pieces of 16 bytes): the plane's 64 tiles of a wave go from the table straight into LDS by
// LDS-DMA (`buffer_load_dwordx4 ... lds`), PP 1-KiB instructions instead of sixteen element gathers
// per lane plus sixteen ds_write: no VGPR staging, no store traffic through the vector unit.  A
// DMA instruction writes lane L's 16 bytes at base + 16 L, so in instruction q lane L fetches the
// piece that belongs at slot 64 q + L of the wave's image, which is laid out point-major (point p
// = owner lane p, PP slots each) with a point's pieces rotated: piece c sits in slot
// (c + rot(p)) & (PP - 1), rot(p) = p >> 1 (f64) or p >> 2 (f32).  The rotation makes the readers
// conflict-free: the 16 lanes a ds_read_b128 services together then hit 16 different 16-byte
// bank groups.  `dma_off[q]` = byte offset of that piece in the table without the plane part
// (computed once per point from the owners' tile offsets with ds_bpermute); the plane part
// `delta` is the instruction's scalar offset; `lds_wave` = LDS byte address of the wave's image
// (wave-uniform, in a scalar register).
template <typename T>
__device__ __forceinline__ void dma_issue_plane(__amdgpu_buffer_rsrc_t bricks, const unsigned* dma_off, unsigned delta, unsigned lds_wave) {
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
#pragma unroll
  for (int q = 0; q < (int)sizeof(T); ++q)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(bricks, (lds_byte*)(size_t)(lds_wave + (unsigned)q * 1024u), 16, dma_off[q], delta, 0, 0);
}

// Wait for the plane in flight, take my tile out of the image (v[e], e = ei * 4 + ej).
template <typename T>
__device__ __forceinline__ void dma_take_tile(unsigned lds_wave, unsigned wl, T (&v)[16]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wave_sync();
  constexpr unsigned PP = (unsigned)sizeof(T);  // pieces per tile
  constexpr int EP = 16 / (int)sizeof(T);      // elements per piece
  typedef T TP __attribute__((ext_vector_type(EP), may_alias));
  typedef __attribute__((address_space(3))) const TP lds_TP;
  const unsigned rot = cubic_dma_rot<T>(wl);
#pragma unroll
  for (int c = 0; c < (int)PP; ++c) {
    const unsigned slot = wl * PP + (((unsigned)c + rot) & (PP - 1u));
    const TP w = *(lds_TP*)(size_t)(lds_wave + slot * 16u);
#pragma unroll
    for (int k = 0; k < EP; ++k) v[EP * c + k] = w[k];
  }
  // every lane must HAVE its tile (reads returned, not merely issued) before the next plane's
  // DMA — which the caller issues next, ahead of this plane's arithmetic — overwrites the image
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wave_sync();
}

// dims 0 and 1 of one tile
template <typename T, bool RECT, bool FMA, bool FAST = false>
__device__ __forceinline__ T reduce_tile(const T (&v)[16], const typename CubicDimSel<T, RECT>::type* dim, unsigned interior, bool& ok) {
  T w4[4];
  if constexpr (!RECT) {
    if (interior & 1u) {
      cubic_tile_dim0_interior<FMA, T>(v, dim[0].tt, w4);  // f32: two nodes per packed instruction
    } else {
#pragma unroll
      for (int ej = 0; ej < 4; ++ej) w4[ej] = cubic_regular_node<FMA, T>(v[ej], v[4 + ej], v[8 + ej], v[12 + ej], dim[0]);
    }
    if (interior & 2u) return cubic_regular_node_interior<FMA, T>(w4[0], w4[1], w4[2], w4[3], dim[1].tt);
    return cubic_regular_node<FMA, T>(w4[0], w4[1], w4[2], w4[3], dim[1]);
  } else {
#pragma unroll
    for (int ej = 0; ej < 4; ++ej) w4[ej] = cubic_node_sel<RECT, FMA, FAST, T>(v[ej], v[4 + ej], v[8 + ej], v[12 + ej], dim[0], ok);
    return cubic_node_sel<RECT, FMA, FAST, T>(w4[0], w4[1], w4[2], w4[3], dim[1], ok);
  }
}

template <typename T, bool RECT, bool FMA>
__device__ __forceinline__ T reduce_tile(const T (&v)[16], const typename CubicDimSel<T, RECT>::type* dim, unsigned interior) {
  bool unused = true;
  return reduce_tile<T, RECT, FMA, false>(v, dim, interior, unused);
}

// All 4^(N-2) planes of a point, software-pipelined: the DMA of plane k+1 is issued as soon as
// every lane has taken its tile of plane k out of the image, i.e. BEFORE plane k's nodes are
// evaluated, so the table latency of the next plane hides behind the arithmetic of this one
// with one tile image per wave and no extra registers (the tile is in registers anyway).
// Plane order and reduction tree are the reference's (dim 2 inside dim 3;
// src/multicubic/regular.rs:368-421).
// FAST: see cubic_node_sel; `*ok` (in: the dimensions' `fast`; out: every division of the point was the short form's to take).
template <typename T, int N, bool RECT, bool FMA, bool FAST = false>
__device__ __forceinline__ T reduce_planes_dma(__amdgpu_buffer_rsrc_t bricks, const unsigned* dma_off, const unsigned* plane_stride,
                                               unsigned lds_wave, unsigned wl,
                                               const typename CubicDimSel<T, RECT>::type* dim, unsigned interior, bool* okp = nullptr) {
  bool ok = FAST ? *okp : true;
  static_assert(N >= 2 && N <= 4, "tiled multicubic: N = 2..4");
  constexpr int NP = N == 2 ? 1 : (N == 3 ? 4 : 16);
  auto delta_of = [&](int k) -> unsigned {  // byte offset of plane k: dim 2 index = k & 3, dim 3 index = k >> 2
    unsigned d = 0;
    if constexpr (N >= 3) d += (unsigned)(k & 3) * plane_stride[2];
    if constexpr (N >= 4) d += (unsigned)(k >> 2) * plane_stride[3];
    return d * (unsigned)sizeof(T);
  };
  dma_issue_plane<T>(bricks, dma_off, delta_of(0), lds_wave);
  T s2[4], s3[4];
  T res = (T)0;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    T v[16];
    dma_take_tile<T>(lds_wave, wl, v);
    if (k + 1 < NP) dma_issue_plane<T>(bricks, dma_off, delta_of(k + 1), lds_wave);
    const T r01 = reduce_tile<T, RECT, FMA, FAST>(v, dim, interior, ok);
    if constexpr (N == 2) {
      res = r01;
    } else {
      s2[k & 3] = r01;
      if ((k & 3) == 3) {
        const T r2 = cubic_node_sel<RECT, FMA, FAST, T>(s2[0], s2[1], s2[2], s2[3], dim[2], ok);
        if constexpr (N == 3) {
          res = r2;
        } else {
          s3[k >> 2] = r2;
          if (k == NP - 1) res = cubic_node_sel<RECT, FMA, FAST, T>(s3[0], s3[1], s3[2], s3[3], dim[3], ok);
        }
      }
    }
  }
  if constexpr (FAST) *okp = ok;
  return res;
}

// Reduce plane dimensions D..2 (D = N-1 outermost): 4 sub-results along dim D, then its node.
// FAST: see cubic_node_sel (`*okp` as in reduce_planes_dma).
template <typename T, int D, bool RECT, bool FMA, bool DMA, bool FAST = false>
struct PlaneReduce {
  __device__ __forceinline__ static T run(__amdgpu_buffer_rsrc_t bricks, const unsigned* toff, unsigned delta,
                                          const unsigned* plane_stride, T __attribute__((may_alias))* lds_data, unsigned group, unsigned me,
                                          const typename CubicDimSel<T, RECT>::type* dim, unsigned interior, bool* okp = nullptr) {
    T s[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
      s[o] = PlaneReduce<T, D - 1, RECT, FMA, DMA, FAST>::run(bricks, toff, delta + (unsigned)o * plane_stride[D] * (unsigned)sizeof(T), plane_stride,
                                                              lds_data, group, me, dim, interior, okp);
    bool ok = FAST ? *okp : true;
    const T r = cubic_node_sel<RECT, FMA, FAST, T>(s[0], s[1], s[2], s[3], dim[D], ok);
    if constexpr (FAST) *okp = ok;
    return r;
  }
};
template <typename T, bool RECT, bool FMA, bool DMA, bool FAST>
struct PlaneReduce<T, 1, RECT, FMA, DMA, FAST> {
  __device__ __forceinline__ static T run(__amdgpu_buffer_rsrc_t bricks, const unsigned* toff, unsigned delta,
                                          const unsigned*, T __attribute__((may_alias))* lds_data, unsigned group, unsigned me,
                                          const typename CubicDimSel<T, RECT>::type* dim, unsigned interior, bool* okp = nullptr) {
    static_assert(!DMA, "the LDS-DMA form runs reduce_planes_dma");
    return gather_plane<T, RECT, FMA, FAST>(bricks, toff, delta, lds_data, group, me, dim, interior, okp);
  }
};

template <typename T, int N, bool RECT, bool FMA, int SI, int SJ>
__global__ void __launch_bounds__(kBlock) k_cubic_brick(const CubicBrickArgs<T, N> a) {
  typedef typename CubicDimSel<T, RECT>::type DimT;
  if (a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // (launch-uniform)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  // One region, used first for the offset transpose (u32) and then for the data transposes (T).
  typedef T __attribute__((may_alias)) lds_T;
  lds_T* lds_data = reinterpret_cast<lds_T*>(smem_raw);
  lds_u32* lds_off = reinterpret_cast<lds_u32*>(smem_raw);
  constexpr bool DMA = cubic_dma<T, SI, SJ>();
  constexpr size_t kRegion = cubic_lds_region<T, SI, SJ>();
  unsigned char* lds_axes = smem_raw + kRegion;
  if (RECT && a.ax.use_lds) stage_axes<T, N>(a.ax, lds_axes);
  const unsigned char* axis_base = (RECT && a.ax.use_lds) ? lds_axes : a.ax.image;
  const unsigned lane = threadIdx.x;
  const unsigned me = lane & 15;
  const unsigned group = lane >> 4;
  // The offset matrix of a group lives inside the SAME bytes as its data matrix (both regions are
  // private to the group's wave): index it with the data matrix' group stride.
  const unsigned goff = group * (unsigned)(16 * kCubRow * sizeof(T) / 4);
  const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(a.bricks, a.table_bytes);
  // LDS byte address of this wave's tile image (LDS-DMA gather), in a scalar register
  const unsigned lds_wave = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw + (lane >> 6) * cubic_dma_image<T>()));
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  const size_t per_xcd = (size_t)(gridDim.x >> 3) * kBlock;  // points one XCD's workgroups cover per iteration
  const size_t niter = a.eighth ? (a.eighth + per_xcd - 1) / per_xcd : (a.npts + nthreads - 1) / nthreads;
  for (size_t it = 0; it < niter; ++it) {
    size_t i0 = it * nthreads + (size_t)blockIdx.x * kBlock + lane;
    bool live = i0 < a.npts;
    if (a.eighth) {
      const size_t within = it * per_xcd + (size_t)(blockIdx.x >> 3) * kBlock + lane;
      i0 = (size_t)(blockIdx.x & 7u) * a.eighth + within;
      live = within < a.eighth && i0 < a.npts;
    }
    const size_t dst = (a.scatter && live) ? (size_t)a.scatter[i0] : i0;  // where the result goes
    DimT dim[N];
    int loc[N];
    bool ok = true;
    [[maybe_unused]] T xs[N];
#pragma unroll
    for (int d = 0; d < N; ++d) {
      if constexpr (RECT) {
        const T x = live ? stream_load(a.obs[d] + i0) : (T)0;
        const Axis<T> ax = make_axis<T, N>(a.ax, axis_base, d);
        xs[d] = x;
        if (a.crec[0]) loc[d] = cubic_rect_locate_rec<T>(ax, a.crec[d], x, a.linearize, dim[d]);  // (wave-uniform)
        else loc[d] = cubic_rect_locate<T, true>(ax, x, a.linearize, /*fma_linear=*/false, dim[d]);  // multicubic/rectilinear.rs:366-408
      } else {
        const T x = live ? stream_load(a.obs[d] + i0) : a.start[d];
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
    }
    if (!RECT && !ok && live) atomicMin(a.first_bad, (unsigned long long)(a.index_base + dst));
    // Offsets of my point's 16 footprint elements (plane base included) -> LDS, transposed.
    unsigned pbase = 0;  // element offsets here, bytes in LDS
#pragma unroll
    for (int d = 2; d < N; ++d) pbase += (unsigned)loc[d] * a.plane_stride[d];
    unsigned toff[16];
    if constexpr (DMA) {
      // my point's tile (steps 1,1: tile index = cell) as a byte offset; instruction q of a plane's
      // DMA has me fetch piece c of point p (gather_plane_dma)
      constexpr unsigned PP = (unsigned)sizeof(T);  // 16-byte pieces per tile; 64 / PP points per DMA instruction
      const unsigned wl = lane & 63u;
      const unsigned tb = (pbase + (unsigned)(loc[0] * (int)a.nbj + loc[1]) * 16u) * (unsigned)sizeof(T);
#pragma unroll
      for (int q = 0; q < (int)PP; ++q) {
        const unsigned p = ((unsigned)q * 64u + wl) / PP;
        const unsigned c = ((wl & (PP - 1u)) - cubic_dma_rot<T>(p)) & (PP - 1u);
        toff[q] = (unsigned)__shfl((int)tb, (int)p) + c * 16u;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        int bi, oi, bj, oj;
        tile_coord<SI>(loc[0], e >> 2, &bi, &oi);
        tile_coord<SJ>(loc[1], e & 3, &bj, &oj);
        lds_off[goff + e * kCubRow + me] = (pbase + ((unsigned)(bi * (int)a.nbj + bj) * 16u) + (unsigned)(oi * 4 + oj)) * (unsigned)sizeof(T);
      }
      wave_sync();
#pragma unroll
      for (int r = 0; r < 16; ++r) toff[r] = lds_off[goff + me * kCubRow + r];
      wave_sync();
    }
    // bit d set: every lane of this wave is interior along dim d (d = 0, 1; regular grids)
    unsigned interior = 0;
    if constexpr (!RECT) {
#pragma unroll
      for (int d = 0; d < 2; ++d)
        if (__builtin_amdgcn_ballot_w64(dim[d].sat != kSatNone || dim[d].linear != 0) == 0) interior |= 1u << d;
    }
    T res;
    if constexpr (RECT) {
      // the nodes' spacing-ratio divisions first without divide sequences; a wave with a lane whose operands that form
      // does not take (a ratio or a numerator outside its exponent window, a -0) evaluates once more as the reference writes it
      bool fast = true;
#pragma unroll
      for (int d = 0; d < N; ++d) fast = fast && dim[d].fast;
      if constexpr (DMA) res = reduce_planes_dma<T, N, RECT, FMA, true>(rsrc, toff, a.plane_stride, lds_wave, lane & 63u, dim, interior, &fast);
      else res = PlaneReduce<T, N - 1, RECT, FMA, false, true>::run(rsrc, toff, 0u, a.plane_stride, lds_data, group, me, dim, interior, &fast);
      if (__any(!fast)) {
        if (a.crec[0]) {  // (the records' t and coefficients were the short forms': this wave's again, by division)
#pragma unroll
          for (int d = 0; d < N; ++d)
            (void)cubic_rect_locate<T>(make_axis<T, N>(a.ax, axis_base, d), xs[d], a.linearize, /*fma_linear=*/false, dim[d]);
        }
        if constexpr (DMA) res = reduce_planes_dma<T, N, RECT, FMA>(rsrc, toff, a.plane_stride, lds_wave, lane & 63u, dim, interior);
        else res = PlaneReduce<T, N - 1, RECT, FMA, false>::run(rsrc, toff, 0u, a.plane_stride, lds_data, group, me, dim, interior);
      }
    } else if constexpr (DMA)
      res = reduce_planes_dma<T, N, RECT, FMA>(rsrc, toff, a.plane_stride, lds_wave, lane & 63u, dim, interior);
    else
      res = PlaneReduce<T, N - 1, RECT, FMA, false>::run(rsrc, toff, 0u, a.plane_stride, lds_data, group, me, dim, interior);
    if (live) stream_store(a.out + dst, res);
  }
}

// Tiled table builder.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_cubic_tiles(const T* __restrict__ vals, T* __restrict__ tiles, size_t nplanes,
                                                              int n0, int n1, int si, int sj, unsigned nbi, unsigned nbj) {
  const size_t per_plane = (size_t)nbi * nbj * 16;
  const size_t total = nplanes * per_plane;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const size_t plane = e / per_plane;
    size_t b = e - plane * per_plane;
    const unsigned within = (unsigned)(b & 15);
    b >>= 4;
    const unsigned bj = (unsigned)(b % nbj);
    const unsigned bi = (unsigned)(b / nbj);
    const int i = (int)bi * si + (int)(within >> 2);
    const int j = (int)bj * sj + (int)(within & 3);
    T v = (T)0;
    // vals is C-ordered (i, j, plane...) with the plane dims fastest: index = (i*n1 + j)*nplanes + plane
    if (i < n0 && j < n1) v = vals[((size_t)i * n1 + j) * nplanes + plane];
    tiles[e] = v;
  }
}


}  // namespace interpn
