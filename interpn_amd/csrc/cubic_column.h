// 4-D multicubic on SORTED points with (part of) the table column of a cell resident in LDS.
//
// After the counting sort of k_bin_points.hip with one bin per (i, j) cell of dims 0 and 1, all
// points of a bin read the SAME 4 x 4 (i, j) footprint of every (k, l) plane: n2 x n3 tiles of the
// fully overlapped tile table (cubic_brick.h), 128 B each in f64 — 128 KiB for cfg4's 32 x 32
// planes.  A part of a bin (<= 32 points per thread of a wave group) is sorted locally by the
// class pair of dims 2, 3 and evaluated out of LDS: a point still reads its 16 tiles = 2 KiB, but
// from the CU's own LDS (256 B/clk) instead of 16 L2 lines.
//
// Round 4 shape (profiles/r04_column_stamps.txt has the in-kernel time stamps behind it):
//
//  * PERSISTENT workgroups, one per CU, of GROUPS wave groups (two groups of six waves for cfg4).
//    Each group draws parts from a global counter and works through them on its own — its own LDS
//    sub-column, local order, histogram, and a group barrier made of an LDS counter (no s_barrier
//    after the prologue) — so the set-up of one group's part (part lookup, record loads, local
//    sort, LDS-DMA fill: memory latency, no arithmetic; 13 % of a round-3 workgroup's life) runs
//    under the plane arithmetic of the other group, and there is no dispatch gap between parts.
//    (Two 384-thread WORKGROUPS per CU would be the obvious form; the dispatcher does not
//    co-schedule them at three waves per SIMD — measured: 92 % of the time one workgroup per CU —
//    hence the groups inside one workgroup.)
//  * the column is resident only a K-RANGE AT A TIME.  The local order is by the class of dim 2
//    first, so the points whose dim-2 class lies in [r * cpp, (r + 1) * cpp) are one contiguous
//    stretch of it ("phase" r) and need only the tile rows k = loc2 .. loc2 + 3 of those classes:
//    cpp + 3 rows of n3 tiles.  Columns larger than the LDS (48^4: 288 KiB) are more phases.
//  * rows of 64 slots are handed to a group's waves on demand (LDS counter), a wave whose row is
//    beyond the stretch is done, and the record of the next row is requested before this row's
//    planes are evaluated.
//  * LDS layout: tile T of the sub-column (T = row * n3 + l) at T * PITCH, PITCH = tile + 16 bytes
//    (144 / 80): consecutive tiles fall on different 16-byte bank groups (9 and 5 are odd), lanes
//    that read the same tile broadcast, and a tile's address is ONE multiply-add per point plus
//    wave-uniform and immediate offsets (the 16-tile interleave of round 3 spent 28 VALU
//    instructions per plane row on addresses).  Rows are filled by LDS-DMA with lane -> (tile l,
//    piece) fixed per instruction slot and the row in the scalar offset: no per-lane division
//    (round 3's fill divided by n3 per lane and instruction: 13 % of the kernel's VALU work).
//
//  * (second session of round 4) COEFFICIENT COLUMNS: all points of a part share the arm of their
//    64 dim-0 nodes, whose inputs are table values only — the group rewrites the tile lines of the
//    resident sub-column as the spline's coefficients once, and a point's dim-0 node is Horner's
//    three steps (section "Coefficient columns" below; 0.79 -> 0.54 ms for cfg4's kernel,
//    rectilinear 1.74 -> 0.81).  The tile is taken half at a time in f64; the next row's record is
//    consumed before this row's results are stored (vmcnt counts in order); the local order lives in
//    the tiles' padding, so that cfg4's padded column (144 KiB) is resident whole.  The kernel is then
//    at the fabric's request rate — one 128-byte read per point for its gathered record, one partial
//    write for its result — and its plane loop is 7 % of it (profiles/REJECTED.md, ablation row).
//
// Arithmetic, plane order and reduction tree are those of cubic_brick.h / the reference
// (src/multicubic/regular.rs:325-623): bit-identical results.  A point whose exact cell is not
// the part's, or whose exact dim-2 rows are not in the phase's sub-column (the sorts estimate
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
  const unsigned* part_prefix;  // parts in front of bin b; [nbins] = total
  unsigned* work;               // next part to hand out (zeroed by the sort's scan kernel)
  const unsigned* bin_flags;    // linearised extrapolation: one bit per sorted bin, set by the sort if any of its points lies outside the grid along dim 0 (a hint); else null
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
  unsigned sub_bytes;        // LDS bytes of a group's sub-column; what of its local order (16-bit) does not fit the tiles' padding sits behind them
  unsigned perm_pad;         // entries of the local order kept in the padding of the sub-column's tiles (0: bare tiles)
  unsigned group_bytes;      // dynamic LDS bytes per group (sub-column + local order, 16-byte multiple)
  unsigned long long* stamps;  // measurement aid (option debug_stamps): 8 words per part, or null
  // Rectilinear grids (RECT kernels): the handle's axis image (coordinates + bucket tables of all
  // four axes), staged once per workgroup behind the groups' regions at byte `axes_lds_off` of the
  // dynamic LDS.  Classes are then exact (partition_point), not estimates.
  AxisArgs<T, 4> ax;
  unsigned axes_lds_off;
  // dim 0 by coefficients (see "Coefficient columns" below); 0: every node from the table values
  int coef;
  // LDS bytes from tile to tile: the tile + 16 (see the head of this file), or the bare tile where
  // only that lets the whole column fit (one phase per part instead of two: k_cubic_column.hip)
  unsigned pitch;
  // 1: the sort left every point's local sort key in the upper eight bits of its index word
  // (k_bin_points.hip::column_key; regular grids, slices of at most 2^24 points): the local sort's
  // first pass reads those 4 bytes per point instead of the 32-byte records
  int index_keys;
#ifdef INTERPN_COLUMN_CREC_VARIANT
  // Regression build only (tools/libinterpn_colvariant.so, profiles/NOTES.md section H): the argument and the
  // never-taken branch that made round 5's compiler mis-structure the node's three class arms (see cubic_rect_node).
  const CubicCellRecord<T>* crec[4];
#endif
};

constexpr int kColPerThread = 32;  // points of a part per thread of its group at most (the local sort's key registers)
constexpr unsigned kColMaxPart = 12288;  // points of a part at most (16-bit local order; = kColumnMaxPart of interpn_host.h)
// points per thread of a group of GT threads: 16 for 768 threads, 32 for 384 and fewer
constexpr int col_per_thread(int gt) { return (int)((kColMaxPart + gt - 1) / gt) < kColPerThread ? (((int)((kColMaxPart + gt - 1) / gt) + 1) / 2 * 2) : kColPerThread; }
constexpr int kColKeys = 1024;     // keys of the local sort


// LDS pitch of a tile: its 16 elements + 16 bytes (see the head of this file)
template <typename T> constexpr unsigned col_pitch() { return 16u * (unsigned)sizeof(T) + 16u; }
template <typename T> constexpr unsigned col_pitch_units() { return (unsigned)sizeof(T) + 1u; }  // in 16-byte units: 9 / 5
// LDS bytes of `nrows` rows of n3 tiles, whole KiB
template <typename T> __host__ __device__ inline size_t col_lds_bytes(unsigned nrows, unsigned n3) {
  return ((size_t)nrows * n3 * col_pitch<T>() + 1023u) / 1024u * 1024u;
}

template <typename T> __device__ __forceinline__ T dev_fabs(T a);
template <> __device__ __forceinline__ double dev_fabs<double>(double a) { return __builtin_fabs(a); }
template <> __device__ __forceinline__ float dev_fabs<float>(float a) { return __builtin_fabsf(a); }
template <typename T> __device__ __forceinline__ T dev_fmax(T a, T b);
template <> __device__ __forceinline__ double dev_fmax<double>(double a, double b) { return __builtin_fmax(a, b); }
template <> __device__ __forceinline__ float dev_fmax<float>(float a, float b) { return __builtin_fmaxf(a, b); }
template <typename T> __device__ __forceinline__ T dev_fmin(T a, T b);
template <> __device__ __forceinline__ double dev_fmin<double>(double a, double b) { return __builtin_fmin(a, b); }
template <> __device__ __forceinline__ float dev_fmin<float>(float a, float b) { return __builtin_fminf(a, b); }

// my tile (16 elements, e = ei * 4 + ej) at LDS byte address `addr`
template <typename T>
__device__ __forceinline__ void col_take_tile(unsigned addr, T (&v)[16]) {
  constexpr int PP = (int)sizeof(T);       // 16-byte pieces per tile
  constexpr int EP = 16 / (int)sizeof(T);  // elements per piece
  typedef T TP __attribute__((ext_vector_type(EP), may_alias));
  typedef __attribute__((address_space(3))) const TP lds_TP;
#pragma unroll
  for (int c = 0; c < PP; ++c) {
    const TP w = *(lds_TP*)(size_t)(addr + (unsigned)c * 16u);
#pragma unroll
    for (int k = 0; k < EP; ++k) v[EP * c + k] = w[k];
  }
}

// Per-dimension state of a point on a regular grid, as the column kernel keeps it: `tt` as in
// CubicDimRegular (t, -t or t - 1), the class as three predicates (wave masks, not integers).
template <typename T>
struct ColDim {
  T tt;
  bool low, high, lin;  // saturated low / high (inside or outside); outside with linearised extrapolation
};

template <typename T>
__device__ __forceinline__ CubicDimRegular<T> col_full_dim(const ColDim<T>& d) {
  CubicDimRegular<T> r;
  r.tt = d.tt;
  r.sat = d.low ? kSatLow : (d.high ? kSatHigh : kSatNone);
  r.linear = d.lin ? 1 : 0;
  return r;
}

// One point from the table in global memory (points that are not in this part's cell or rows):
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
// Within a part all points share the class of dims 0 and 1, and the local sort puts points of one
// (dim 2, dim 3) class pair next to each other, so almost every wave is uniform along every
// dimension: all lanes interior (the reference's Saturation::None arm), all saturated low, or all
// saturated high, none of them extrapolating linearly.  Each of these is the reference's arm for
// that case without the selects of the general form (same operations, same bits,
// multicubic/regular.rs:495-623); anything else takes the general form.
enum : int { kFormNone = 0, kFormLow = 1, kFormHigh = 2, kFormMixed = 3 };

template <bool FMA, int FORM, typename T>
__device__ __forceinline__ T col_node(T v0, T v1, T v2, T v3, const ColDim<T>& d) {
  const T two = (T)2;
  if constexpr (FORM == kFormNone) {
    return cubic_regular_node_interior<FMA, T>(v0, v1, v2, v3, d.tt);
  } else if constexpr (FORM == kFormLow) {   // InsideLow / OutsideLow without linearisation: regular.rs:507-527
    const T dy = v0 - v1;
    const T k0 = -((v2 - v0) / two);
    const T k1 = mul_add<FMA>(two, dy, -k0);  // regular.rs:525-528
    return hermite<FMA>(d.tt, v1, dy, k0, k1);
  } else if constexpr (FORM == kFormHigh) {  // InsideHigh / OutsideHigh without linearisation: regular.rs:563-582
    const T dy = v3 - v2;
    const T k0 = (v3 - v1) / two;
    const T k1 = mul_add<FMA>(two, dy, -k0);
    return hermite<FMA>(d.tt, v2, dy, k0, k1);
  } else {
    return cubic_regular_node<FMA, T>(v0, v1, v2, v3, col_full_dim<T>(d));
  }
}

template <bool FMA, typename T>
__device__ __forceinline__ T col_node_rt(int form, T v0, T v1, T v2, T v3, const ColDim<T>& d) {
  switch (form) {  // wave-uniform
    case kFormNone: return col_node<FMA, kFormNone, T>(v0, v1, v2, v3, d);
    case kFormLow: return col_node<FMA, kFormLow, T>(v0, v1, v2, v3, d);
    case kFormHigh: return col_node<FMA, kFormHigh, T>(v0, v1, v2, v3, d);
    default: return col_node<FMA, kFormMixed, T>(v0, v1, v2, v3, d);
  }
}

// the wave's form along one dimension (every lane must call this)
template <typename T>
__device__ __forceinline__ int col_wave_form(const ColDim<T>& d) {
  if (__builtin_amdgcn_ballot_w64(d.lin) != 0) return kFormMixed;
  if (__builtin_amdgcn_ballot_w64(d.low || d.high) == 0) return kFormNone;
  if (__builtin_amdgcn_ballot_w64(!d.low) == 0) return kFormLow;
  if (__builtin_amdgcn_ballot_w64(!d.high) == 0) return kFormHigh;
  return kFormMixed;
}

// ---- Coefficient columns -----------------------------------------------------------------------
// All points of a part share the class of dim 0, i.e. the saturation arm of the 64 dim-0 nodes of
// their footprint, and those nodes' inputs are table values only.  What the reference computes
// for such a node before it touches t — y0, dy, k0, k1 and the spline's c1, c2, c3
// (multicubic/regular.rs:495-582, mod.rs:72-81; rectilinear.rs:437-533 with mod.rs:103-117) — is
// therefore the same for every point of the part: once the sub-column has landed, the group
// rewrites each tile line (v[ei][ej], ei = 0..3) in place as (y0, c1, c2, c3)[ej], by the
// reference's own operations in its order, and a point's dim-0 node is Horner's three steps
// (mod.rs:83-91) on them: 3 instead of 14 instructions (rectilinear: 3 instead of ~40 with two
// IEEE divisions) for 64 of a point's 85 nodes.  Same operations on the same operands: the same
// bits.  A point whose own arm is not the part's (a point the sort's estimate put in the wrong
// bin; a point outside the grid when extrapolation is linearised) is evaluated from the table in
// global memory like every other point that does not belong to its part.
template <bool FMA, typename T>
__device__ __forceinline__ HermiteCoef<T> col_node_coef(int form, T v0, T v1, T v2, T v3) {
  const T two = (T)2;
  if (form == kFormNone) {         // regular.rs:495-505
    const T dy = v2 - v1;
    const T k0 = (v2 - v0) / two;
    const T k1 = (v3 - v1) / two;
    return hermite_coef<T>(v1, dy, k0, k1);
  } else if (form == kFormLow) {   // regular.rs:507-527
    const T dy = v0 - v1;
    const T k0 = -((v2 - v0) / two);
    const T k1 = mul_add<FMA>(two, dy, -k0);
    return hermite_coef<T>(v1, dy, k0, k1);
  } else {                         // regular.rs:563-582
    const T dy = v3 - v2;
    const T k0 = (v3 - v1) / two;
    const T k1 = mul_add<FMA>(two, dy, -k0);
    return hermite_coef<T>(v2, dy, k0, k1);
  }
}

// All 16 planes of a point out of the LDS sub-column, dim 0 in form F0: dim 2 index = k2, dim 3
// index = k3 (the reference's order, multicubic/regular.rs:368-421).  `a0` = LDS address of the
// point's tile (k2, k3) = (0, 0); `rowpitch` = bytes between tile rows (wave-uniform).  The tile of
// the next plane is requested before this plane's nodes are evaluated (32 more VGPRs).
// NONE01: dims 0 and 1 of the wave are interior (all but the boundary bins: 80 of the 85 nodes of
// a point belong to them) — their nodes are compiled without form tests; the five nodes of dims 2, 3
// always go by the wave's form (f2, f3): after the local sort a wave of 64 points spans about six
// (dim 2, dim 3) class pairs, and a third of all waves touch a boundary class of dim 3.
template <typename T, bool FMA, int F0, bool NONE01 = false>
__device__ __forceinline__ T col_reduce(unsigned a0, unsigned rowpitch, unsigned PITCH, const ColDim<T>* dim, int f1, int f2, int f3) {
  T s3[4];
  T cur[16];
  unsigned ak = a0;  // tile (0, k3)
  col_take_tile<T>(ak, cur);
#pragma unroll 1
  for (int k3 = 0; k3 < 4; ++k3) {
    T s2[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
      T nxt[16];
      // plane after (k2, k3): (k2 + 1, k3), or (0, k3 + 1); behind the last plane this reads one
      // tile too many ((0, 4): the next tile of the point's first row or the first of the next row,
      // still inside the sub-column)
      const unsigned an = k2 < 3 ? ak + (unsigned)(k2 + 1) * rowpitch : ak + PITCH;
      col_take_tile<T>(an, nxt);
      T w[4];
      if constexpr (F0 == kFormNone) {
        cubic_tile_dim0_interior<FMA, T>(cur, dim[0].tt, w);  // f32: two nodes per packed instruction (interpn_device.h)
      } else {
#pragma unroll
        for (int ej = 0; ej < 4; ++ej) w[ej] = col_node<FMA, F0, T>(cur[ej], cur[4 + ej], cur[8 + ej], cur[12 + ej], dim[0]);
      }
      if constexpr (NONE01) s2[k2] = col_node<FMA, kFormNone, T>(w[0], w[1], w[2], w[3], dim[1]);
      else s2[k2] = col_node_rt<FMA, T>(f1, w[0], w[1], w[2], w[3], dim[1]);
#pragma unroll
      for (int e = 0; e < 16; ++e) cur[e] = nxt[e];
    }
    ak += PITCH;
    s3[k3] = col_node_rt<FMA, T>(f2, s2[0], s2[1], s2[2], s2[3], dim[2]);
  }
  return col_node_rt<FMA, T>(f3, s3[0], s3[1], s3[2], s3[3], dim[3]);
}

// ---- all 16 planes of a point out of a COEFFICIENT sub-column -----------------------------------
// Half a tile at a time in f64: the 16-byte pieces 2k + h (k = y0, c1, c2, c3) hold the
// coefficients of the lines ej = 2h, 2h + 1, so a half is four loads = 16 registers and feeds two
// of the tile's four dim-0 nodes; the next half (or the next plane's first) is requested before
// this half's Horner steps.  32 registers of tile data in flight instead of the 64 of col_reduce
// (146 registers, nothing spilled; a 1024-thread shape at 128 registers was measured and is not
// built: profiles/REJECTED.md).  An f32 tile is four pieces (piece k = coefficient k of all four
// lines) and is taken whole.
template <typename T> struct ColHalf {
  static constexpr int EP = 16 / (int)sizeof(T);   // elements per piece: 2 / 4
  static constexpr int H = (int)sizeof(T) == 8 ? 2 : 1;  // halves per tile
  typedef T TP __attribute__((ext_vector_type(EP), may_alias));
  typedef __attribute__((address_space(3))) const TP lds_TP;
  TP c[4];
  __device__ __forceinline__ void take(unsigned tile_addr, int h) {
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = *(lds_TP*)(size_t)(tile_addr + (unsigned)(k * H + h) * 16u);
  }
  // the dim-0 nodes of my lines: w[h * EP + j]
  template <bool FMA>
  __device__ __forceinline__ void horner(T t, int h, T (&w)[4]) const {
    if constexpr (sizeof(T) == 4) {  // two nodes per packed instruction, lane-wise the scalar operations
      const float_pair tt = {t, t};
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float_pair y0 = {c[0][2 * p], c[0][2 * p + 1]}, c1 = {c[1][2 * p], c[1][2 * p + 1]};
        const float_pair c2 = {c[2][2 * p], c[2][2 * p + 1]}, c3 = {c[3][2 * p], c[3][2 * p + 1]};
        float_pair r;
        if constexpr (FMA) {
          r = __builtin_elementwise_fma(__builtin_elementwise_fma(__builtin_elementwise_fma(c3, tt, c2), tt, c1), tt, y0);
        } else {
          const float_pair i0 = tt * c3;
          const float_pair i1 = c2 + i0;
          const float_pair i2 = tt * i1;
          const float_pair i3 = c1 + i2;
          const float_pair i4 = tt * i3;
          r = y0 + i4;
        }
        w[2 * p] = r.x;
        w[2 * p + 1] = r.y;
      }
    } else {
#pragma unroll
      for (int j = 0; j < EP; ++j) w[h * EP + j] = hermite_eval<FMA, T>(t, c[0][j], c[1][j], c[2][j], c[3][j]);
    }
  }
};

// regular grids; NONE1: dim 1 of the wave is interior (its node without form tests); f1..f3: the wave's forms
template <typename T, bool FMA, bool NONE1>
__device__ __forceinline__ T col_reduce_coef(unsigned a0, unsigned rowpitch, unsigned PITCH, const ColDim<T>* dim, int f1, int f2, int f3) {
  constexpr int H = ColHalf<T>::H;
  T s3[4];
  unsigned ak = a0;  // tile (0, k3)
  ColHalf<T> cur;
  cur.take(ak, 0);
#pragma unroll 1
  for (int k3 = 0; k3 < 4; ++k3) {
    T s2[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
      const unsigned at = ak + (unsigned)k2 * rowpitch;
      T w[4];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        // behind the last plane this reads half a tile too many ((0, 4): the next tile of the
        // point's first row or the first of the next row, still inside the sub-column)
        ColHalf<T> nxt;
        if (h + 1 < H) nxt.take(at, h + 1);
        else nxt.take(k2 < 3 ? at + rowpitch : ak + PITCH, 0);
        cur.template horner<FMA>(dim[0].tt, h, w);
        cur = nxt;
      }
      if constexpr (NONE1) s2[k2] = col_node<FMA, kFormNone, T>(w[0], w[1], w[2], w[3], dim[1]);
      else s2[k2] = col_node_rt<FMA, T>(f1, w[0], w[1], w[2], w[3], dim[1]);
    }
    ak += PITCH;
    s3[k3] = col_node_rt<FMA, T>(f2, s2[0], s2[1], s2[2], s2[3], dim[2]);
  }
  return col_node_rt<FMA, T>(f3, s3[0], s3[1], s3[2], s3[3], dim[3]);
}

// rectilinear grids (the reference's node for dims 1..3: multicubic/rectilinear.rs:413-545)
template <typename T, bool FMA>
__device__ __forceinline__ T col_reduce_coef_rect(unsigned a0, unsigned rowpitch, unsigned PITCH, const CubicDimRect<T>* dim) {
  constexpr int H = ColHalf<T>::H;
  T s3[4];
  unsigned ak = a0;
  ColHalf<T> cur;
  cur.take(ak, 0);
#pragma unroll 1
  for (int k3 = 0; k3 < 4; ++k3) {
    T s2[4];
#pragma unroll 1
    for (int k2 = 0; k2 < 4; ++k2) {
      const unsigned at = ak + (unsigned)k2 * rowpitch;
      T w[4];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        ColHalf<T> nxt;
        if (h + 1 < H) nxt.take(at, h + 1);
        else nxt.take(k2 < 3 ? at + rowpitch : ak + PITCH, 0);
        cur.template horner<FMA>(dim[0].t, h, w);
        cur = nxt;
      }
      s2[k2] = cubic_rect_node<FMA, T>(w[0], w[1], w[2], w[3], dim[1]);
    }
    ak += PITCH;
    s3[k3] = cubic_rect_node<FMA, T>(s2[0], s2[1], s2[2], s2[3], dim[2]);
  }
  return cubic_rect_node<FMA, T>(s3[0], s3[1], s3[2], s3[3], dim[3]);
}

// The waves that are not interior along every dimension (boundary bins, boundary classes of dims
// 2, 3, extrapolating points): out of line, so that their node forms' registers do not weigh on the
// allocation of the common path (measured: the inlined switch made the compiler spill 24 registers
// around every form).
template <typename T, bool FMA>
__device__ __noinline__ T col_reduce_general(unsigned a0, unsigned rowpitch, unsigned pitch, T tt0, T tt1, T tt2, T tt3, unsigned cls, int forms) {
  ColDim<T> dim[4];
  const T tt[4] = {tt0, tt1, tt2, tt3};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    dim[d].tt = tt[d];
    dim[d].low = (cls >> (3 * d)) & 1u;
    dim[d].high = (cls >> (3 * d + 1)) & 1u;
    dim[d].lin = (cls >> (3 * d + 2)) & 1u;
  }
  const int f0 = forms & 3, f1 = (forms >> 2) & 3, f2 = (forms >> 4) & 3, f3 = (forms >> 6) & 3;
  if (forms & 0x100) return col_reduce_coef<T, FMA, false>(a0, rowpitch, pitch, dim, f1, f2, f3);  // coefficient column (group-uniform)
  switch (f0) {  // wave-uniform
    case kFormNone: return col_reduce<T, FMA, kFormNone>(a0, rowpitch, pitch, dim, f1, f2, f3);
    case kFormLow: return col_reduce<T, FMA, kFormLow>(a0, rowpitch, pitch, dim, f1, f2, f3);
    case kFormHigh: return col_reduce<T, FMA, kFormHigh>(a0, rowpitch, pitch, dim, f1, f2, f3);
    default: return col_reduce<T, FMA, kFormMixed>(a0, rowpitch, pitch, dim, f1, f2, f3);
  }
}

// class estimate of the local sort (a hint only): 0 = floc <= 0, c = floc, n - 2 = floc >= n - 2
template <typename T>
__device__ __forceinline__ unsigned col_class_hint(T x, T start, T rstep, int n) {
  const T u = (x - start) * rstep;
  return u >= (T)1 ? (u < (T)(n - 2) ? (unsigned)(int)u : (unsigned)(n - 2)) : 0u;
}

// ---- rectilinear grids --------------------------------------------------------------------------
// Class of a coordinate along a rectilinear axis, from the reference's cell search
// (multicubic/rectilinear.rs:377: iloc = partition_point(g < x) - 2): 0 = saturated low (iloc <= -1,
// inside or outside), c = iloc + 1 for interior cells, n - 2 = saturated high (iloc >= n - 3) — the
// same numbering as on regular grids, footprint cell = clamp(class - 1, 0, n - 4); exact, since it
// is the search the kernel itself uses.
template <typename T>
__device__ __forceinline__ unsigned col_rect_class(const Axis<T>& ax, T x) {
  int iloc = axis_partition_point<T>(ax, x) - 2;
  iloc = iloc < -1 ? -1 : (iloc > ax.n - 3 ? ax.n - 3 : iloc);
  return (unsigned)(iloc + 1);
}

// All 16 planes of a point on a rectilinear grid out of the LDS sub-column: the reference's tree
// (multicubic/rectilinear.rs:290-356) with its node (rectilinear.rs:413-545; two IEEE divisions by
// the spacing ratios per node, which is what this kernel spends its time on: no tile prefetch).
template <typename T, bool FMA>
__device__ __forceinline__ T col_reduce_rect(unsigned a0, unsigned rowpitch, unsigned PITCH, const CubicDimRect<T>* dim) {
  T s3[4];
  unsigned ak = a0;
#pragma unroll 1
  for (int k3 = 0; k3 < 4; ++k3) {
    T s2[4];
#pragma unroll 1
    for (int k2 = 0; k2 < 4; ++k2) {
      T v[16];
      col_take_tile<T>(ak + (unsigned)k2 * rowpitch, v);
      s2[k2] = reduce_tile<T, true, FMA>(v, dim, 0u);
    }
    ak += PITCH;
    s3[k3] = cubic_rect_node<FMA, T>(s2[0], s2[1], s2[2], s2[3], dim[2]);
  }
  return cubic_rect_node<FMA, T>(s3[0], s3[1], s3[2], s3[3], dim[3]);
}

// One rectilinear point from the table in global memory (a point that is not in this part's cell or rows).
template <typename T, bool FMA>
__device__ __noinline__ T col_slow_point_rect(__amdgpu_buffer_rsrc_t rsrc, unsigned tile_off_bytes, unsigned ps2_bytes, unsigned ps3_bytes,
                                              const CubicDimRect<T>* dim) {
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
      s2[k2] = reduce_tile<T, true, FMA>(v, dim, 0u);
    }
    s3[k3] = cubic_rect_node<FMA, T>(s2[0], s2[1], s2[2], s2[3], dim[2]);
  }
  return cubic_rect_node<FMA, T>(s3[0], s3[1], s3[2], s3[3], dim[3]);
}

// Barrier of one wave group (GW waves) of a persistent workgroup: an LDS counter that only ever
// grows; `epoch` (wave-uniform register) is the count that completes the next barrier.  LDS
// instructions of a wave execute in order and the LDS serialises the waves' accesses, so what a
// wave wrote to LDS before its add is visible to every wave that sees the counter complete; the
// LDS-DMA of a fill is waited for (vmcnt) by the issuing wave before it arrives here.  GW == 0: the
// group is the whole workgroup (s_barrier).
// `diag` (diagnosis builds, INTERPN_COLUMN_DIAG): set to non-zero when a wave reaches a barrier with
// lanes switched off — an s_barrier there would be executed by waves whose EXEC is empty as well.
template <int GW>
__device__ __forceinline__ void col_group_barrier(unsigned* ctr, unsigned& epoch, unsigned wl, unsigned* diag = nullptr) {
  if (diag && __builtin_amdgcn_read_exec() != ~0ull) *diag |= 1u;
  if constexpr (GW == 0) {
    __syncthreads();
  } else {
    epoch += (unsigned)GW;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (wl == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - epoch) < 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// Registers: 12 waves per CU = three per SIMD (168 VGPRs).  GROUPS wave groups of THREADS / GROUPS
// threads each; GROUPS == 1: the whole workgroup is one group and synchronises with s_barrier.
// STAMPS: the measurement build (option debug_stamps) — time stamps cost registers the product kernel needs.
// RECT: rectilinear grid (exact classes from the axis search, the reference's rectilinear node).
template <typename T, bool RECT, bool FMA, int THREADS, int GROUPS, bool STAMPS = false>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_cubic_column(const CubicColumnArgs<T> a) {
  static_assert(THREADS % (64 * GROUPS) == 0, "whole waves per group");
  constexpr int GT = THREADS / GROUPS;   // threads of a group
  constexpr int GW = GT / 64;            // its waves
  constexpr int PT = col_per_thread(GT); // points of a part per thread at most
#ifdef INTERPN_COLUMN_SBARRIER  // measurement / diagnosis builds only (tools/): the one-group form on s_barrier
  constexpr int BAR = GROUPS == 1 ? 0 : GW;
#else
  constexpr int BAR = GW;  // (s_barrier for GROUPS == 1 — BAR = 0 — hangs on the GPU at 32^4 inside this persistent loop: not used)
#endif
  constexpr unsigned PP = (unsigned)sizeof(T);  // 16-byte pieces of a tile
  const unsigned PITCH = a.pitch;               // bytes from tile to tile (group-uniform)
  const unsigned PU = PITCH >> 4;               // ... in 16-byte units
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_col[];
  __shared__ unsigned s_hist_all[GROUPS][kColKeys];  // local sort: points per key, then (after pass 2) the END of every key's stretch
  __shared__ unsigned s_wave_all[GROUPS][GW];
  __shared__ unsigned s_ctl_all[GROUPS][8];           // 0 barrier counter | 1 part | 2 bin | 3 begin | 4 end | 5 next row
  const unsigned tid = threadIdx.x;
  const unsigned grp = GROUPS == 1 ? 0u : (unsigned)__builtin_amdgcn_readfirstlane((int)(tid / (unsigned)GT));
  unsigned* const s_hist = s_hist_all[grp];
  unsigned* const s_wave = s_wave_all[grp];
  unsigned* const s_ctl = s_ctl_all[grp];
  if (tid - grp * (unsigned)GT < 8) s_ctl[tid - grp * (unsigned)GT] = 0;
  const unsigned char* const axis_base = smem_col + a.axes_lds_off;  // RECT: the axes in LDS
  if constexpr (RECT) {
    const unsigned words = a.ax.image_bytes >> 2;
    const unsigned* src = reinterpret_cast<const unsigned*>(a.ax.image);
    unsigned* dst = reinterpret_cast<unsigned*>(smem_col + a.axes_lds_off);
    for (unsigned k = tid; k < words; k += THREADS) dst[k] = src[k];
  }
  __syncthreads();  // the only s_barrier of the workgroup
  unsigned epoch = 0;
  const unsigned total_parts = (unsigned)__builtin_amdgcn_readfirstlane((int)a.part_prefix[a.nbins]);
  const unsigned n3 = (unsigned)a.n[3];
  const int ncls2 = a.n[2] - 1;
  const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(a.tiles, a.table_bytes);
  const unsigned ps2 = a.plane_stride[2] * (unsigned)sizeof(T), ps3 = a.plane_stride[3] * (unsigned)sizeof(T);
  const unsigned lds_col = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_col + grp * a.group_bytes;
  // Local order (slot -> point of the part, 16 bits each): its first a.perm_pad entries sit in the
  // 16 bytes of padding behind every tile of the sub-column (eight entries per tile: the fills and
  // the coefficient pass write tile pieces only), the rest behind the sub-column.  That is what
  // lets cfg4's whole padded column (144 KiB) share the LDS with the order of a 12 288-point part.
  unsigned char* const grp_base = smem_col + (size_t)grp * a.group_bytes;
  const unsigned perm_pad = a.perm_pad;
  auto perm_at = [&](unsigned i) -> unsigned short* {
    const unsigned in_pad = (i >> 3) * PITCH + 16u * (unsigned)sizeof(T) + (i & 7u) * 2u;
    const unsigned in_tail = a.sub_bytes + (i - perm_pad) * 2u;
    return reinterpret_cast<unsigned short*>(grp_base + (i < perm_pad ? in_pad : in_tail));
  };
  const unsigned rowpitch = n3 * PITCH;  // LDS bytes of a tile row
  typedef T RV __attribute__((ext_vector_type(4)));

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
  // the stretch of the local order that belongs to phase r (valid once pass 2 and a barrier are behind us)
  auto phase_stretch = [&](int r, unsigned* ps, unsigned* pe) {
    const unsigned klo = (unsigned)(r * a.cpp) * (unsigned)a.q3;
    int chi = (r + 1) * a.cpp;
    chi = chi < ncls2 ? chi : ncls2;
    const unsigned khi = (unsigned)chi * (unsigned)a.q3;
    *ps = klo ? (unsigned)__builtin_amdgcn_readfirstlane((int)s_hist[klo - 1]) : 0u;
    *pe = (unsigned)__builtin_amdgcn_readfirstlane((int)s_hist[khi - 1]);
  };

  const unsigned gtid_k = tid - grp * (unsigned)GT;
  const unsigned index_mask = a.index_keys ? 0x00FFFFFFu : 0xFFFFFFFFu;
#ifdef INTERPN_COLUMN_DIAG
  // Diagnosis build (tools/column_barrier_diag.py): does the source meet s_barrier's contract?  Per part,
  // every wave reports how many group barriers it went through (they must agree) and whether it ever
  // reached one with lanes switched off; the words go where the STAMPS build keeps its time stamps.
  unsigned diag_bits = 0, diag_epoch0 = 0;
  unsigned diag_prev = ~0u;
#define COL_DIAG_ARG (&diag_bits)
#else
#define COL_DIAG_ARG nullptr
#endif
  for (;;) {
    // Per-part copies of the thread ids that the optimiser cannot see through: everything derived
    // from them (16 record addresses, fill offsets, ...) is then recomputed per part instead of
    // being hoisted out of the persistent loop and kept in (spilled) registers for the whole kernel.
    unsigned gtid = gtid_k;
    asm volatile("" : "+v"(gtid));
    const unsigned gwave = (unsigned)__builtin_amdgcn_readfirstlane((int)(gtid >> 6)), wl = gtid & 63u;
#ifdef INTERPN_COLUMN_DIAG
    if (a.stamps && diag_prev != ~0u && wl == 0) {  // the part just finished: [1] most, [2] fewest barriers of a wave, [3] lanes-off flag
      unsigned long long* ws = a.stamps + (size_t)diag_prev * 8u;
      const unsigned long long nbar = (epoch - diag_epoch0) / (unsigned)GW;
      atomicMax(&ws[1], nbar);
      atomicMin(&ws[2], nbar);
      if (diag_bits) atomicOr(&ws[3], 1ull);
      atomicAdd(&ws[0], 1ull);  // waves that reported
    }
    diag_bits = 0;
#endif
    // ---- next part for this group
    if (gtid == 0) s_ctl[1] = atomicAdd(a.work, 1u);
    col_group_barrier<BAR>(&s_ctl[0], epoch, wl, COL_DIAG_ARG);
    const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane((int)s_ctl[1]);  // group-uniform values go to scalar registers
#ifdef INTERPN_COLUMN_DIAG
    diag_prev = w < total_parts ? w : ~0u;
    diag_epoch0 = epoch;  // barriers of this part: from here to the next part's first one (inclusive)
#endif
    if (w >= total_parts) break;
    unsigned long long t_stamp[5] = {0, 0, 0, 0, 0};
    if (STAMPS && gtid == 0) t_stamp[0] = wall_clock64();
    // Which (bin, part) is it?  part_prefix is non-decreasing: the bin b with part_prefix[b] <= w <
    // part_prefix[b + 1] — every thread tests bins.
    for (int b = (int)gtid; b < a.nbins; b += GT) {
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
        s_ctl[3] = lo_p < b1 ? lo_p : b1;
        s_ctl[4] = hi_p;
        s_ctl[2] = (unsigned)b;
      }
    }
    for (unsigned c = gtid; c < (unsigned)kColKeys; c += GT) s_hist[c] = 0;
    if (gtid == 0) s_ctl[5] = 0;
    col_group_barrier<BAR>(&s_ctl[0], epoch, wl, COL_DIAG_ARG);
    const int bin = __builtin_amdgcn_readfirstlane((int)s_ctl[2]);
    const unsigned begin = (unsigned)__builtin_amdgcn_readfirstlane((int)s_ctl[3]), end = (unsigned)__builtin_amdgcn_readfirstlane((int)s_ctl[4]);
    if (begin >= end) continue;          // group-uniform
    const unsigned count = end - begin;  // <= PT * GT (the scan cut the bin accordingly)
    if (STAMPS && gtid == 0) t_stamp[1] = wall_clock64();
    const unsigned key = (unsigned)(((unsigned long long)(unsigned)bin * a.inv_mult) % (unsigned)a.nbins);
    const int c0 = (int)(key / (unsigned)a.nb1), c1 = (int)(key % (unsigned)a.nb1);  // nominal classes of dims 0, 1
    const int ci = c0 - 1 < 0 ? 0 : (c0 - 1 > a.n[0] - 4 ? a.n[0] - 4 : c0 - 1);      // their footprint cell
    const int cj = c1 - 1 < 0 ? 0 : (c1 - 1 > a.n[1] - 4 ? a.n[1] - 4 : c1 - 1);
    const unsigned cell_off = (unsigned)(ci * (int)a.nbj + cj) * 16u * (unsigned)sizeof(T);  // my cell's tile inside a plane, bytes
    const RV* __restrict__ recs = reinterpret_cast<const RV*>(a.records) + begin;
    const unsigned* __restrict__ index = a.index + begin;
    // dim 0 by coefficients ("Coefficient columns" above)?  xf = the arm every point of the part
    // shares along dim 0, or -1: the tiles keep the table values.  The saturated classes hold
    // points inside and outside the grid alike; with linearised extrapolation the outside ones
    // have no spline, so those parts keep the values.
    // ... unless the sort saw none of the bin's points outside (a.bin_flags: the usual case, points
    // inside the grid): then the saturated arm's spline serves all of them; a point that extrapolates
    // after all (the flag is a hint) goes to the table in global memory like any point off its part's arm.
    int xf = -1;
    if (a.coef) {
      if (c0 > 0 && c0 < a.n[0] - 2) xf = kFormNone;
      else if (!a.linearize) xf = c0 == 0 ? kFormLow : kFormHigh;
      else if (a.bin_flags && ((a.bin_flags[(unsigned)bin >> 5] >> ((unsigned)bin & 31u)) & 1u) == 0) xf = c0 == 0 ? kFormLow : kFormHigh;
    }
    // tile lines (v[ei][ej], ei = 0..3) of `nrows` rows -> (y0, c1, c2, c3)[ej], in place
    auto to_coef = [&](unsigned nrows) {
      typedef __attribute__((address_space(3))) T lds_T;
      CubicDimRect<T> d0;
      if constexpr (RECT) {
        d0.sat = xf;  // kForm* == kSat*
        d0.linear = 0;
        d0.fma_linear = 0;
        cubic_rect_dim_setup<T>(make_axis<T, 4>(a.ax, axis_base, 0).g, ci, (T)0, d0);  // d0.t is not used
      }
      const unsigned lines = nrows * n3 * 4u;
      for (unsigned k = gtid; k < lines; k += GT) {
        const unsigned ad = lds_col + (k >> 2) * PITCH + (k & 3u) * (unsigned)sizeof(T);
        lds_T* const pv = (lds_T*)(size_t)ad;
        const T v0 = pv[0], v1 = pv[4], v2 = pv[8], v3 = pv[12];
        HermiteCoef<T> c;
        if constexpr (RECT) c = cubic_rect_node_coef<FMA, T>(v0, v1, v2, v3, d0);
        else c = col_node_coef<FMA, T>(xf, v0, v1, v2, v3);
        pv[0] = c.y0;
        pv[4] = c.c1;
        pv[8] = c.c2;
        pv[12] = c.c3;
      }
    };

    // Sub-column fill, row by row: instruction slot j of a row moves the row's 16-byte units
    // 64 j .. 64 j + 63 (1 KiB); unit u = tile l = u / PU, piece u % PU (the last unit of a tile
    // is its padding: not written).  Lane -> (l, piece) does not depend on the row; the row goes
    // into the instruction's scalar offset.  Rows are dealt to the group's waves.
    auto fill = [&](unsigned row0, unsigned nrows) {
      typedef __attribute__((address_space(3))) unsigned char lds_byte;
      const unsigned units = n3 * PU;
      const unsigned slots = (units + 63u) / 64u;
      for (unsigned j = 0; j < slots; ++j) {
        const unsigned u = j * 64u + wl;
        const unsigned l = PU == PP ? u / PP : u / (PP + 1u), piece = u - l * PU;  // PU is PP or PP + 1
        const bool valid = l < n3 && piece < PP;
        const unsigned voff = l * ps3 + cell_off + piece * 16u;
        for (unsigned k = gwave; k < nrows; k += GW) {
          const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds_col + k * rowpitch + j * 1024u));
          if (valid) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_byte*)(size_t)dst, 16, voff, (row0 + k) * ps2, 0, 0);
        }
      }
    };

    // ---- local sort, pass 1: the (dim 2, dim 3) class pair of each of my points -> histogram,
    // in two batches of 16 loads per thread; the first phase's fill is issued behind the first.
    unsigned cls23[PT / 2];  // two 16-bit keys per register
    constexpr int NB = PT > 16 ? 2 : 1;  // batches of at most 16 loads per thread (all of a batch in flight together)
    constexpr int HB = PT / NB;
#pragma unroll
    for (int h = 0; h < NB; ++h) {
      if (a.index_keys) {  // group-uniform: the keys come with the index words
        unsigned ck[HB];
#pragma unroll
        for (int m = 0; m < HB; ++m) {
          const unsigned q = (unsigned)(h * HB + m) * (unsigned)GT + gtid;
          ck[m] = q < count ? index[q] >> 24 : 0u;
        }
#pragma unroll
        for (int m = 0; m < HB; ++m) {
          const unsigned q = (unsigned)(h * HB + m) * (unsigned)GT + gtid;
          const int mm = h * HB + m;
          if (mm & 1) cls23[mm / 2] |= ck[m] << 16;
          else cls23[mm / 2] = ck[m];
          if (q < count) atomicAdd(&s_hist[ck[m]], 1u);
        }
      } else if (h == 1 && (unsigned)HB * (unsigned)GT >= count) {  // group-uniform: nothing in the second batch
#pragma unroll
        for (int m = 0; m < HB / 2; ++m) cls23[HB / 2 + m] = 0;
      } else {
        T x2[HB], x3[HB];
#pragma unroll
        for (int m = 0; m < HB; ++m) {
          const unsigned q = (unsigned)(h * HB + m) * (unsigned)GT + gtid;
          const RV r = q < count ? recs[q] : recs[0];
          x2[m] = r[2];
          x3[m] = r[3];
        }
#pragma unroll
        for (int m = 0; m < HB; ++m) {
          const unsigned q = (unsigned)(h * HB + m) * (unsigned)GT + gtid;
          unsigned h2, h3;
          if constexpr (RECT) {
            h2 = col_rect_class<T>(make_axis<T, 4>(a.ax, axis_base, 2), x2[m]);
            h3 = col_rect_class<T>(make_axis<T, 4>(a.ax, axis_base, 3), x3[m]);
          } else {
            h2 = col_class_hint<T>(x2[m], a.start[2], a.rstep[2], a.n[2]);
            h3 = col_class_hint<T>(x3[m], a.start[3], a.rstep[3], a.n[3]);
          }
          const unsigned c = h2 * (unsigned)a.q3 + (h3 >> a.sh3);  // < (n2 - 1) q3 <= 1024
          const int mm = h * HB + m;
          if (mm & 1) cls23[mm / 2] |= c << 16;
          else cls23[mm / 2] = c;
          if (q < count) atomicAdd(&s_hist[c], 1u);
        }
      }
      if (h == NB - 1) {  // the first sub-column travels while the scan runs (behind the record loads: vmcnt is in order)
        unsigned row0, nrows;
        phase_rows(0, &row0, &nrows);
        fill(row0, nrows);
      }
    }
    col_group_barrier<BAR>(&s_ctl[0], epoch, wl, COL_DIAG_ARG);
    if (STAMPS && gtid == 0) t_stamp[2] = wall_clock64();
    // exclusive scan of the counters: CPT consecutive counters per thread, wave scan, wave totals
    {
      constexpr int CPT = (kColKeys + GT - 1) / GT;
      unsigned mine[CPT];
      unsigned sum = 0;
#pragma unroll
      for (int c = 0; c < CPT; ++c) { mine[c] = gtid * CPT + c < (unsigned)kColKeys ? s_hist[gtid * CPT + c] : 0u; sum += mine[c]; }
      unsigned incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, off);
        if (wl >= (unsigned)off) incl += up;
      }
      if (wl == 63u) s_wave[gwave] = incl;
      col_group_barrier<BAR>(&s_ctl[0], epoch, wl, COL_DIAG_ARG);
      unsigned run = incl - sum;
      for (unsigned ww = 0; ww < gwave; ++ww) run += s_wave[ww];
#pragma unroll
      for (int c = 0; c < CPT; ++c) { if (gtid * CPT + c < (unsigned)kColKeys) s_hist[gtid * CPT + c] = run; run += mine[c]; }
    }
    col_group_barrier<BAR>(&s_ctl[0], epoch, wl, COL_DIAG_ARG);
    // pass 2: slots (each counter ends up at the END of its key's stretch = the start of the next key's)
#pragma unroll
    for (int m = 0; m < PT; ++m) {
      const unsigned q = (unsigned)m * (unsigned)GT + gtid;
      const unsigned c = (m & 1) ? (cls23[m / 2] >> 16) : (cls23[m / 2] & 0xFFFFu);
      if (q < count) *perm_at(atomicAdd(&s_hist[c], 1u)) = (unsigned short)q;
    }
    if (STAMPS && gtid == 0) t_stamp[3] = wall_clock64();

    // ---- the phases: K-range r of the column in LDS, the stretch of the local order that needs it
    for (int r = 0; r < a.nphase; ++r) {
      unsigned row0, nrows, ps = 0, pe = 0;
      phase_rows(r, &row0, &nrows);
      if (r > 0) {
        phase_stretch(r, &ps, &pe);  // final: phase 0's barrier lies behind pass 2
        if (pe == ps) continue;      // group-uniform: nobody needs these rows
        col_group_barrier<BAR>(&s_ctl[0], epoch, wl, COL_DIAG_ARG);  // every wave has finished with the previous sub-column
        if (gtid == 0) s_ctl[5] = 0;
        fill(row0, nrows);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my share of the sub-column has landed
      col_group_barrier<BAR>(&s_ctl[0], epoch, wl, COL_DIAG_ARG);
      if (r == 0) {
        if (STAMPS && gtid == 0) t_stamp[4] = wall_clock64();
        phase_stretch(0, &ps, &pe);
        if (pe == ps) continue;
      }
      if (xf >= 0) {  // group-uniform
        to_coef(nrows);
        col_group_barrier<BAR>(&s_ctl[0], epoch, wl, COL_DIAG_ARG);
      }
      const int row_top = (int)nrows - 4;  // largest footprint row inside the sub-column

      // Rows of 64 slots are handed to the waves as they free up (a shared counter: rows differ in
      // cost — node forms, out-of-cell points — and the waves of a SIMD share its issue slots); a
      // wave that draws a row beyond the stretch is done with the phase.
      auto draw_row = [&]() -> unsigned {
        unsigned v = 0;
        if (wl == 0) v = atomicAdd(&s_ctl[5], 1u);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
      };
      unsigned jw = ps + draw_row() * 64u;
      if (jw >= pe) continue;
      // dead lanes (the stretch's tail) redo its last point: they keep their wave uniform
      unsigned q = *perm_at(jw + wl < pe ? jw + wl : pe - 1);
      RV rec = recs[q];
      T rcur[4] = {rec[0], rec[1], rec[2], rec[3]};
      for (;;) {
        const unsigned orig = index[q] & index_mask;  // used at the very end: its latency hides behind the planes
        const unsigned jn = ps + draw_row() * 64u;
        if (jn < pe) {  // wave-uniform: next row's record on its way while this row's planes are evaluated
          q = *perm_at(jn + wl < pe ? jn + wl : pe - 1);
          rec = recs[q];
        }
        T res;
        bool ok = true;
        if constexpr (RECT) {
          CubicDimRect<T> dim[4];
          int loc[4];
#pragma unroll
          for (int d = 0; d < 4; ++d)  // multicubic/rectilinear.rs:366-408 (never fails: NaN takes cell 0 and propagates)
#ifdef INTERPN_COLUMN_CREC_VARIANT
            if (a.crec[0]) loc[d] = cubic_rect_locate_rec<T>(make_axis<T, 4>(a.ax, axis_base, d), a.crec[d], rcur[d], a.linearize, dim[d]);
            else
#endif
            loc[d] = cubic_rect_locate<T>(make_axis<T, 4>(a.ax, axis_base, d), rcur[d], a.linearize, /*fma_linear=*/false, dim[d]);
          const int rel2 = loc[2] - (int)row0;
          const bool in_rows = rel2 >= 0 && rel2 <= row_top;
          const unsigned a0 = lds_col + ((unsigned)(in_rows ? rel2 : 0) * n3 + (unsigned)loc[3]) * PITCH;
          if (xf >= 0) res = col_reduce_coef_rect<T, FMA>(a0, rowpitch, PITCH, dim);
          else res = col_reduce_rect<T, FMA>(a0, rowpitch, PITCH, dim);
          // only deliberately mis-binned points (bin_scramble): classes are exact here; in a coefficient
          // column also the points whose arm along dim 0 is not the part's
          const bool arm0 = xf < 0 || (dim[0].sat == xf && !dim[0].linear);
          if (loc[0] != ci || loc[1] != cj || !in_rows || !arm0) {
            const unsigned toff = ((unsigned)loc[2] * a.plane_stride[2] + (unsigned)loc[3] * a.plane_stride[3] +
                                   (unsigned)(loc[0] * (int)a.nbj + loc[1]) * 16u) * (unsigned)sizeof(T);
            CubicDimRect<T> dcopy[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) dcopy[d] = dim[d];
            res = col_slow_point_rect<T, FMA>(rsrc, toff, ps2, ps3, dcopy);
          }
        } else {
          ColDim<T> dim[4];
          int loc[4];
          unsigned cls = 0;    // per lane: bit 3d low, 3d + 1 high, 3d + 2 linearised (the general forms' input)
          unsigned forms = 0;  // per wave: the form of dim d in bits 2d, 2d + 1
  #pragma unroll
          for (int d = 0; d < 4; ++d) {
            const T x = rcur[d];
            const T floc = dev_floor<T>((x - a.start[d]) / a.step[d]);        // multicubic/regular.rs:435-438
            // num-traits <isize as NumCast>::from: Some iff -2^63 <= floc < 2^63, and `floc - 1` must not
            // overflow isize (floc != -2^63): together |floc| < 2^63 (NaN fails)
            ok &= dev_fabs<T>(floc) < (T)9223372036854775808.0;
            // (the conversions are redone per row: hoisted out of the persistent loop they sit in
            // registers the plane loop needs, and spill)
            int nd = a.n[d];
            asm volatile("" : "+s"(nd));
            const T nn2 = (T)(nd - 2);
            // regular.rs:440-442: iloc = floc - 1 clamped to [0, n - 4], in the float domain (exact
            // integers below 2^31; +-inf clamp; NaN -> 0: that point has failed anyway)
            T c = floc - (T)1;
            c = dev_fmax<T>(c, (T)0);
            c = dev_fmin<T>(c, (T)(nd - 4));
            const int l = (int)c;
            // regular.rs:445-466 on floc = iloc + 1
            const bool low = floc <= (T)0, high = floc >= nn2;
            const bool lin = (floc < (T)0 || floc > nn2) && a.linearize != 0;
            const T index_one_loc = mul_add<false>(a.step[d], c + (T)1, a.start[d]);  // regular.rs:356-360, never fused; (T)(l + 1) == c + 1
            const T t = (x - index_one_loc) / a.step[d];
            dim[d].low = low;
            dim[d].high = high;
            dim[d].lin = lin;
            dim[d].tt = low ? -t : (high ? t - (T)1 : t);
            loc[d] = l;
            const int f = col_wave_form<T>(dim[d]);  // wave-uniform
            forms |= (unsigned)f << (2 * d);
            if (f != kFormNone) cls |= ((low ? 1u : 0u) | (high ? 2u : 0u) | (lin ? 4u : 0u)) << (3 * d);
          }
          // footprint rows relative to the sub-column; a point whose exact rows are not all in it is
          // evaluated from the table below (its LDS reads stay inside the sub-column, result dropped)
          const int rel2 = loc[2] - (int)row0;
          const bool in_rows = rel2 >= 0 && rel2 <= row_top;
          const unsigned a0 = lds_col + ((unsigned)(in_rows ? rel2 : 0) * n3 + (unsigned)loc[3]) * PITCH;
          bool arm0 = true;
          if (xf >= 0) {  // coefficient column (group-uniform): dim 0 is Horner's steps whatever the wave's form there
            if ((forms & 0xCu) == 0) {  // dim 1 interior: the common path
              res = col_reduce_coef<T, FMA, true>(a0, rowpitch, PITCH, dim, 0, (int)((forms >> 4) & 3u), (int)((forms >> 6) & 3u));
            } else {
              res = col_reduce_general<T, FMA>(a0, rowpitch, PITCH, dim[0].tt, dim[1].tt, dim[2].tt, dim[3].tt, cls, (int)(forms | 0x100u));
            }
            // my own arm along dim 0 must be the part's
            arm0 = !dim[0].lin && (xf == kFormNone ? !(dim[0].low || dim[0].high) : (xf == kFormLow ? dim[0].low : dim[0].high));
          } else {  // the tiles hold the table values (saturated class of dim 0 under linearised extrapolation; column_coef = 0)
            res = col_reduce_general<T, FMA>(a0, rowpitch, PITCH, dim[0].tt, dim[1].tt, dim[2].tt, dim[3].tt, cls, (int)forms);
          }
          // not my cell / not my rows (the sorts' estimates and the exact cell disagree on a boundary) / not the
          // part's arm along dim 0: from the table
          if (loc[0] != ci || loc[1] != cj || !in_rows || !arm0) {
            const unsigned toff = ((unsigned)loc[2] * a.plane_stride[2] + (unsigned)loc[3] * a.plane_stride[3] +
                                   (unsigned)(loc[0] * (int)a.nbj + loc[1]) * 16u) * (unsigned)sizeof(T);
            CubicDimRegular<T> dcopy[4];
  #pragma unroll
            for (int d = 0; d < 4; ++d) {  // from the packed classes (all zero where the wave's form is None: sat None, not linearised)
              dcopy[d].tt = dim[d].tt;
              dcopy[d].sat = ((cls >> (3 * d)) & 1u) ? kSatLow : (((cls >> (3 * d + 1)) & 1u) ? kSatHigh : kSatNone);
              dcopy[d].linear = (int)((cls >> (3 * d + 2)) & 1u);
            }
            res = col_slow_point<T, FMA>(rsrc, toff, ps2, ps3, dcopy);
          }
        }
        if (!ok) atomicMin(a.first_bad, (unsigned long long)(a.index_base + orig));
        // The next row's record is taken BEFORE this row's results are stored: vmcnt counts in
        // order, so a wait for the record behind the store would also wait for the 64 scattered
        // stores to be acknowledged (measured: the waves spent two thirds of a row's time there).
        // Every lane stores — the dead lanes of the stretch's tail hold copies of its last point
        // and write the same value to the same address — so that the store is straight-line code
        // and the compiler can count what is outstanding.
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          rcur[d] = rec[d];
          asm volatile("" : "+v"(rcur[d]) : : "memory");
        }
        stream_store(a.out + orig, res);
        if (jn >= pe) break;
        jw = jn;
      }
    }
#ifndef INTERPN_COLUMN_DIAG
    if (STAMPS && a.stamps) {
      // [0..4] thread 0's stamps (part drawn | part known | histogram | local order | first sub-column),
      // [5] the group's LAST wave's end, [6] ids, [7] count | group | bin
      unsigned long long* ws = a.stamps + (size_t)w * 8u;
      const unsigned long long t_end = wall_clock64();
      if (wl == 0) atomicMax(&ws[5], t_end);
      if (gtid == 0) {
        for (int k = 0; k < 5; ++k) ws[k] = t_stamp[k];
        unsigned hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ws[6] = ((unsigned long long)xcc << 32) | hw;
        ws[7] = ((unsigned long long)count << 32) | ((unsigned long long)grp << 16) | (unsigned)(bin & 0xFFFF);
      }
    }
#endif
  }
#undef COL_DIAG_ARG
}

}  // namespace interpn
