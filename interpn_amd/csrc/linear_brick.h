// N-D multilinear (N >= 3; regular and rectilinear; f64 and f32) on a bricked copy of the grid
// with a quad-cooperative gather.
//
// Why: with the grid in C order a point's 2^N corners lie on 2^(N-1) different 128-B lines, and on
// MI355X the per-XCD L2 -> L1 line rate (16 lines/clk/XCD, ~2.7e11 lines/s chip-wide), not HBM,
// bounds the kernel (profiles/r01_tune_*).  Here the handle keeps a second
// copy of the grid in which the LAST THREE dimensions (i, j, k) are cut into 2 x 2 x KW bricks of
// one 128-B line each (KW = 4 for f64, 8 for f32), stepped KW-1 along k (a k-pair never leaves a
// brick row) and 1 or 2 along i and j (overlapping bricks = duplicated planes / rows); leading
// dimensions, if any, index whole brick tables.  A 3-D cell then spans 1 (steps 1,1), 1.5 (1,2)
// or 2.25 (2,2) lines instead of 4.25.  Because the TCP only merges lanes of the same instruction,
// the four lanes of a quad fetch the four (i,j) k-pairs of ONE point per load instruction, so
// pieces on the same line become a single L2 request; the pieces are transposed back through LDS
// and every lane finishes its own point with the reference's arithmetic and operation order
// (leading dimensions are reduced first, exactly as dims 0..N-4 are in the reference's tree,
// src/multilinear/regular.rs:347-403), so results are bit-identical to the C-order kernels.
#pragma once

#include "lane_axes.h"
#include "rect_args.h"

namespace interpn {

// CELL == 0: 2(i) x 2(j) x KW(k) bricks over the LAST THREE dimensions, KW = 32 bytes of elements
//            (4 f64 / 8 f32), stepped (SI, SJ, KW-1); one brick = one (i,j) group = 128 B.
// CELL == 1: 2(h) x 2(i) x 2(j) x KW(k) bricks over the LAST FOUR dimensions, KW = 16 bytes of
//            elements (2 f64 / 4 f32), stepped (1, 1, 1, KW-1): a whole 4-D cell (16 f64) is ONE
//            128-B line, where the 3-D bricks need two (one per h plane).  16x the grid in f64
//            (10.7x in f32).  The brick holds its two h planes as two (i,j) groups of 64 B, so the
//            h-pair is reached with an in-brick offset of IJ elements instead of a table stride.
// CELL == 2 (f32 only, round 3): 2(i) x 4(j) x 4(k) bricks over the last three dimensions, stepped
//            (1, 3, 3): every cell lies inside ONE brick = one line, like the fully overlapped
//            2 x 2 x 8 bricks, but the table is 3.56x the grid instead of 4.57x — at 64^3 3.4 MiB
//            instead of 4.85 MiB, i.e. inside the 4 MiB L2 of an XCD.  Only f32 packs 32 values in
//            a line; the f64 analogue of this shape IS the 2 x 2 x 4 brick.
template <typename T, int CELL = 0> struct BrickGeom {
  static constexpr int KW = CELL == 2 ? 4 : (CELL ? 16 : 32) / (int)sizeof(T);  // elements per brick row
  static constexpr int SK = KW - 1;                             // brick step along k
  static constexpr int IJ = 4 * KW;                             // elements of one (i,j) group: the four pieces of a gather
  static constexpr int ELEMS = CELL == 2 ? 32 : (CELL ? 2 : 1) * IJ;  // elements per brick (128 B)
};

template <typename T, int N>
struct BrickArgs {
  const T* bricks;
  const T* obs[N];
  T* out;
  unsigned long long* first_bad;
  size_t npts;
  T start[N];
  T step[N];
  int n[N];
  AxisArgs<T, N> ax;
  unsigned lead_stride[N > 3 ? N - 3 : 1];  // elements of the brick table per unit of a leading index
  unsigned nbj, nbk;
  unsigned iters;  // kBlock-wide iterations per workgroup
  // Gated launch (abi_sweep.hip: a large 3-D batch whose path — this kernel or the sweep kernel — a sampling kernel
  // in front decides on the device): null, or a word that must be non-zero for this launch to do anything.
  const unsigned* gate;
};

template <typename T, int SI, int SJ, int CELL = 0>
__device__ __forceinline__ unsigned brick_piece(unsigned nbj, unsigned nbk, int i, int j, unsigned kpart, int di, int dj) {
  constexpr int KW = BrickGeom<T, CELL>::KW;
  if constexpr (CELL == 2) {  // rows (oi, oj) of 4 elements, oi = 0..1, oj = 0..3; j stepped 3
    const int bj3 = j / 3;
    const int oj3 = j - 3 * bj3 + dj;
    return ((unsigned)(i * (int)nbj + bj3) * nbk) * 32u + (unsigned)((di * 4 + oj3) * 4) + kpart;
  }
  int bi, oi, bj, oj;
  if (SI == 1) { bi = i; oi = di; }
  else { bi = i >> 1; oi = (i & 1) + di; if (oi == 2) { bi += 1; oi = 0; } }
  if (SJ == 1) { bj = j; oj = dj; }
  else { bj = j >> 1; oj = (j & 1) + dj; if (oj == 2) { bj += 1; oj = 0; } }
  return ((unsigned)(bi * (int)nbj + bj) * nbk) * (unsigned)BrickGeom<T, CELL>::ELEMS + (unsigned)((oi * 2 + oj) * KW) + kpart;
}

// x / D for a small constant D and x < 2^20, without the quarter-rate integer multiplies a division by a constant compiles
// to: (x + 0.5) / D lies at least 0.5 / D from every integer, the f32 evaluation of x * (1 / D) + 0.5 / D errs by less than
// 2^20 / D * 2^-23 < 0.05 (x and the fma exact up to one rounding, the two constants 2^-24 relative), so truncation gives
// floor(x / D).  Three full-rate instructions.
template <int D>
__device__ __forceinline__ unsigned div_small(unsigned x) {
  static_assert(D >= 2 && D <= 7, "0.5 / D must stay clear of the error bound");
  return (unsigned)__builtin_fmaf((float)x, 1.0f / (float)D, 0.5f / (float)D);
}

// Byte offset of piece (0, 0) of cell (i, j, k) in the ONE-LINE-PER-CELL layouts (steps 1,1: 2 x 2 x KW bricks, or the
// f32 2 x 4 x 4 bricks) — brick_piece<T, 1, 1, CELL>(nbj, nbk, i, j, bk * ELEMS + (k - bk * SK), 0, 0) * sizeof(T) with
// 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24: full rate, where v_mul_lo_u32, v_mul_hi_u32 and v_mad_u64_u32 run at a
// quarter of it: four of them per row were 16 of a sweep row's ~115 issue slots).  The host takes the sweep kernels only
// where i * nbj + j, nbk < 2^24 and every axis is shorter than 2^20 (k_linear_sweep.hip::sweep_applies).
// a * b + c on 24-bit operands, b wave-uniform (the compiler, left to itself, turns mul24 + add back into v_mad_u64_u32)
__device__ __forceinline__ unsigned mad24_vsv(unsigned a, unsigned b_uniform, unsigned c) {
  unsigned r;
  asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
  return r;
}
// c - a * D for a small constant D (v_mad_i32_i24 with an inline constant)
template <int D>
__device__ __forceinline__ unsigned msub24_const(unsigned a, unsigned c) {
  static_assert(D >= 1 && D <= 16, "inline constant");
  unsigned r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(-D), "v"(c));
  return r;
}

template <typename T, int CELL>
__device__ __forceinline__ unsigned brick_line_bytes24(unsigned nbj, unsigned nbk, int i, int j, int k) {
  typedef BrickGeom<T, CELL> G;
  const unsigned bk = div_small<G::SK>((unsigned)k);
  const unsigned kin = msub24_const<G::SK>(bk, (unsigned)k);  // k - bk * SK
  if constexpr (CELL == 2) {  // rows (oi, oj) of 4 elements; j stepped 3
    const unsigned bj3 = div_small<3>((unsigned)j);
    const unsigned oj3 = msub24_const<3>(bj3, (unsigned)j);
    const unsigned line = mad24_vsv(mad24_vsv((unsigned)i, nbj, bj3), nbk, bk);
    return (line * 32u + oj3 * 4u + kin) * (unsigned)sizeof(T);
  } else {
    const unsigned line = mad24_vsv(mad24_vsv((unsigned)i, nbj, (unsigned)j), nbk, bk);
    return (line * (unsigned)G::ELEMS + kin) * (unsigned)sizeof(T);
  }
}

#ifndef INTERPN_PIECE_ROW
#define INTERPN_PIECE_ROW 5
#endif
constexpr int kPieceRow = INTERPN_PIECE_ROW;  // piece slots per point row in LDS (4 used + 1 pad against bank conflicts)

template <typename T>
struct Cell {
  T v[2][2][2];  // [di][dj][dk]
};

// One cooperative gather: lane q of a quad loads piece q of the quad's points r = 0..3 (offsets in
// `toff`, plus the wave-uniform `delta` of the current leading-dimension combination), then the
// 4x4 piece matrix is transposed through LDS so that every lane owns its point's four pieces.
template <typename T>
__device__ __forceinline__ Cell<T> gather_cell(const T* __restrict__ bricks, const uint4& toff, unsigned delta,
                                               typename LeafVec<T, 2>::type* lds_piece, unsigned quad, unsigned q) {
  typedef typename LeafVec<T, 2>::type P;
  P pc[4];
// (non-temporal gathers were measured: 64^3 2.4 ms, 128^3 2.9 ms -- both much worse -- so the
  // table is read with the default cache policy)
  pc[0] = *reinterpret_cast<const P*>(bricks + toff.x + delta);
  pc[1] = *reinterpret_cast<const P*>(bricks + toff.y + delta);
  pc[2] = *reinterpret_cast<const P*>(bricks + toff.z + delta);
  pc[3] = *reinterpret_cast<const P*>(bricks + toff.w + delta);
#pragma unroll
  for (int r = 0; r < 4; ++r) lds_piece[(quad * 4 + r) * kPieceRow + q] = pc[r];
  wave_sync();
  Cell<T> c;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const P w = lds_piece[(quad * 4 + q) * kPieceRow + p];
    c.v[p >> 1][p & 1][0] = w.x;
    c.v[p >> 1][p & 1][1] = w.y;
  }
  wave_sync();
  return c;
}

// Two cooperative gathers with all eight loads issued before the first LDS exchange (the two
// halves of a 4-D cell share one line when CELL == 1; otherwise twice the lines are in flight).
template <typename T>
__device__ __forceinline__ void gather_cell_pair(const T* __restrict__ bricks, const uint4& toff, unsigned d0, unsigned d1,
                                                 typename LeafVec<T, 2>::type* lds_piece, unsigned quad, unsigned q,
                                                 Cell<T>& c0, Cell<T>& c1) {
  typedef typename LeafVec<T, 2>::type P;
  P pa[4], pb[4];
  pa[0] = *reinterpret_cast<const P*>(bricks + toff.x + d0);
  pa[1] = *reinterpret_cast<const P*>(bricks + toff.y + d0);
  pa[2] = *reinterpret_cast<const P*>(bricks + toff.z + d0);
  pa[3] = *reinterpret_cast<const P*>(bricks + toff.w + d0);
  pb[0] = *reinterpret_cast<const P*>(bricks + toff.x + d1);
  pb[1] = *reinterpret_cast<const P*>(bricks + toff.y + d1);
  pb[2] = *reinterpret_cast<const P*>(bricks + toff.z + d1);
  pb[3] = *reinterpret_cast<const P*>(bricks + toff.w + d1);
#pragma unroll
  for (int r = 0; r < 4; ++r) lds_piece[(quad * 4 + r) * kPieceRow + q] = pa[r];
  wave_sync();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const P w = lds_piece[(quad * 4 + q) * kPieceRow + p];
    c0.v[p >> 1][p & 1][0] = w.x;
    c0.v[p >> 1][p & 1][1] = w.y;
  }
  wave_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) lds_piece[(quad * 4 + r) * kPieceRow + q] = pb[r];
  wave_sync();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const P w = lds_piece[(quad * 4 + q) * kPieceRow + p];
    c1.v[p >> 1][p & 1][0] = w.x;
    c1.v[p >> 1][p & 1][1] = w.y;
  }
  wave_sync();
}

// Reduce leading dimensions 0..D-1 (dim 0 innermost), element-wise on the 8 trailing corners.
// `ls[d]` is the table offset between the two footprint values of leading dimension d.
template <typename T, int D, bool FMA>
struct LeadReduce {
  __device__ __forceinline__ static Cell<T> run(const T* __restrict__ bricks, const uint4& toff, unsigned delta,
                                                const unsigned* ls, const T* t,
                                                typename LeafVec<T, 2>::type* lds_piece, unsigned quad, unsigned q) {
    Cell<T> a, b;
    if constexpr (D == 1) {
      gather_cell_pair<T>(bricks, toff, delta, delta + ls[0], lds_piece, quad, q, a, b);
    } else {
      a = LeadReduce<T, D - 1, FMA>::run(bricks, toff, delta, ls, t, lds_piece, quad, q);
      b = LeadReduce<T, D - 1, FMA>::run(bricks, toff, delta + ls[D - 1], ls, t, lds_piece, quad, q);
    }
    Cell<T> r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const T y0 = a.v[e >> 2][(e >> 1) & 1][e & 1];
      const T dy = b.v[e >> 2][(e >> 1) & 1][e & 1] - y0;
      r.v[e >> 2][(e >> 1) & 1][e & 1] = mul_add<FMA>(t[D - 1], dy, y0);  // regular.rs:378-385
    }
    return r;
  }
};
template <typename T, bool FMA>
struct LeadReduce<T, 0, FMA> {
  __device__ __forceinline__ static Cell<T> run(const T* __restrict__ bricks, const uint4& toff, unsigned delta,
                                                const unsigned*, const T*, typename LeafVec<T, 2>::type* lds_piece,
                                                unsigned quad, unsigned q) {
    return gather_cell<T>(bricks, toff, delta, lds_piece, quad, q);
  }
};

// PPL = points per lane.  With PPL = 2 a lane owns two consecutive points, so coordinates and
// results move as 2*sizeof(T)-byte vectors (16 B in f64): the streams then cost the L2 fewer
// channel-cycles per line (measured -4 % at 64^3, -7 % at 32^3; tools/tune_layout ... w).  Needs all
// obs/out pointers aligned to PPL*sizeof(T); the launcher falls back to a smaller PPL otherwise.
// f32 takes PPL = 4 for the same 16-byte accesses (round 3, 3-D).
// AXR != 0 (rectilinear, every axis <= 64 coordinates): the axes live in registers, one
// coordinate per lane, and are searched with cross-lane reads instead of LDS gathers.
//   AXR == 1: the reference's binary-search probe sequence, six lockstep steps (any axis);
//   AXR == 2: a 255-bucket lane table (sorted finite axes) brackets the answer to the bucket's
//             few coordinates: 1 + 2*scan probes instead of 14 (the LDS pipe is what the
//             rectilinear kernel waits for, profiles/r01_sq_counters_regular_vs_rectilinear.txt).
//
// ABL is a measurement aid and is 0 in every kernel libinterpn_hip.so contains; the other values
// are instantiated only by tools/ablate_linear3d.hip (bench.py's `roofline.ablation`):
//   ABL == 1  stream-only: coordinates are read and the result is written as usual, but the cell
//             values are synthesised from the coordinates (no table access, no LDS exchange);
//   ABL == 2  gather-only: the table gathers, the LDS exchange and the arithmetic run as usual on
//             coordinates synthesised from the point index (uniform over the grid), nothing is
//             read from obs and nothing is stored.
//   ABL == 3  the unmodified data movement with the six IEEE divisions per point replaced by
//             multiplications with a reciprocal (results differ in the last bits: timing only).
//   ABL == 4  coordinates streamed by LDS-DMA (`global_load_lds_dwordx4`, non-temporal) into a
//             per-wave LDS slab and read back with ds_read_b128, instead of VGPR-destination loads
//             (PPL = 2, regular grids, full rows only — the harness launches whole rows).  Correct
//             results; kept out of the library because it measured no gain.
// Their outputs are meaningless by construction.

template <typename T, int N, bool RECT, bool FMA, int SI, int SJ, int PPL, int AXR = 0, int ABL = 0, int CELL = 0>
__global__ void __launch_bounds__(kBlock) k_linear_brick(const BrickArgs<T, N> a) {
  static_assert(CELL == 0 || (CELL == 1 && N >= 4 && SI == 1 && SJ == 1) || (CELL == 2 && sizeof(T) == 4 && SI == 1 && SJ == 1),
                "4-D cell bricks: N >= 4, fully overlapped; 2 x 4 x 4 bricks: f32");
  typedef typename LeafVec<T, 2>::type P;
  typedef BrickGeom<T, CELL> Geom;
  constexpr int L = N - 3;
  constexpr int SK = Geom::SK;
  if (a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;  // (launch-uniform)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  P* lds_piece = reinterpret_cast<P*>(smem_raw);                                               // [quad][r][kPieceRow]
  lds_u32* lds_off = reinterpret_cast<lds_u32*>(smem_raw + kBlock * kPieceRow * sizeof(P));  // [quad][piece][r]
  unsigned char* lds_axes = smem_raw + kBlock * kPieceRow * sizeof(P) + kBlock * 16;
  LaneAxes<T, N> la;
  if constexpr (RECT && AXR != 0) {
    la = load_lane_axes<T, N, AXR>(a.ax);
  } else if (RECT && a.ax.use_lds) {
    stage_axes<T, N>(a.ax, lds_axes);
  }
  const unsigned char* axis_base = (RECT && AXR == 0 && a.ax.use_lds) ? lds_axes : a.ax.image;
  const unsigned lane = threadIdx.x;
  const unsigned q = lane & 3;
  const unsigned quad = lane >> 2;
  typedef T T2 __attribute__((ext_vector_type(2)));
  // Block b owns the contiguous lane slots [b, b+1) * iters * kBlock (PPL points per slot); the
  // grid covers the batch once, so the hardware dispatcher balances the load across XCDs.
  const size_t nslots = (a.npts + PPL - 1) / PPL;
  const size_t first = (size_t)blockIdx.x * a.iters * kBlock;
  for (unsigned it = 0; it < a.iters; ++it) {
    // Every lane runs every iteration (dead lanes still fetch pieces for their quad).
    const size_t s0 = first + (size_t)it * kBlock + lane;
    if (s0 - lane >= nslots) break;  // block-uniform
    const size_t i0 = s0 * PPL;
    T xin[PPL][N];
    bool live[PPL];
#pragma unroll
    for (int h = 0; h < PPL; ++h) live[h] = i0 + h < a.npts;
    if constexpr (ABL == 2) {
#pragma unroll
      for (int h = 0; h < PPL; ++h)
#pragma unroll
        for (int d = 0; d < N; ++d) xin[h][d] = ablate_coord<T>(i0 + h, d, a.start[d], a.step[d], a.n[d]);
    } else if constexpr (ABL == 4 && PPL == 2 && !RECT) {
      // one 1-KiB LDS-DMA per dimension and wave: lane l's 16 bytes land at slab + 16*l
      unsigned char* slab = lds_axes + (threadIdx.x >> 6) * (N * 1024u);
#pragma unroll
      for (int d = 0; d < N; ++d)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.obs[d] + i0),
                                         (__attribute__((address_space(3))) void*)(slab + d * 1024u), 16, 0, 2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < N; ++d) {
        const T2 v = *reinterpret_cast<const T2*>(slab + d * 1024u + (threadIdx.x & 63u) * 16u);
        xin[0][d] = v.x;
        xin[PPL - 1][d] = v.y;
      }
      wave_sync();
    } else if (PPL >= 2) {
      typedef T TV __attribute__((ext_vector_type(PPL)));
#pragma unroll
      for (int d = 0; d < N; ++d) {
        TV v;
#pragma unroll
        for (int h = 0; h < PPL; ++h) v[h] = RECT ? (T)0 : a.start[d];
        if (live[PPL - 1]) {
          v = stream_load(reinterpret_cast<const TV*>(a.obs[d] + i0));
        } else {  // the batch's ragged tail: element by element
#pragma unroll
          for (int h = 0; h < PPL; ++h)
            if (live[h]) v[h] = stream_load(a.obs[d] + i0 + h);
        }
#pragma unroll
        for (int h = 0; h < PPL; ++h) xin[h][d] = v[h];
      }
    } else {
#pragma unroll
      for (int d = 0; d < N; ++d) xin[0][d] = live[0] ? stream_load(a.obs[d] + i0) : (RECT ? (T)0 : a.start[d]);
    }
    // AXR: the PPL x N axis searches advance in lockstep across lanes (lane_axes.h).
    int cell_r[PPL][N];
    T x0_r[PPL][N], x1_r[PPL][N];
    if constexpr (RECT && AXR != 0) lane_axes_locate<T, N, PPL, AXR>(a.ax, la, xin, cell_r, x0_r, x1_r);
    T resv[PPL];
#pragma unroll
    for (int h = 0; h < PPL; ++h) {
      T t[N];
      int loc[N];
      bool ok = true;
#pragma unroll
      for (int d = 0; d < N; ++d) {
        const T x = xin[h][d];
        if (RECT) {
          T x0, x1;
          int l;
          if constexpr (AXR != 0) {
            l = cell_r[h][d];
            x0 = x0_r[h][d];
            x1 = x1_r[h][d];
          } else {
            const Axis<T> ax = make_axis<T, N>(a.ax, axis_base, d);
            l = axis_cell<T>(ax, x, &x0, &x1);  // multilinear/rectilinear.rs:353-370, :310-311
          }
          const T step = x1 - x0;
          t[d] = (x - x0) / step;                       // rectilinear.rs:310-313
          loc[d] = l;
        } else {
          T floc;
          if constexpr (ABL == 3) {  // timing probe: what would the six IEEE divides per point cost?
            floc = dev_floor<T>((x - a.start[d]) * ((T)1 / a.step[d]));
          } else {
            ok &= regular_floc<T>(x, a.start[d], a.step[d], &floc);        // multilinear/regular.rs:415-418
          }
          const int l = clamp_loc<T>(floc, a.n[d] - 2);                     // regular.rs:420-422
          const T izl = mul_add<FMA>(a.step[d], (T)l, a.start[d]);          // regular.rs:334-337
          if constexpr (ABL == 3) t[d] = (x - izl) * ((T)1 / a.step[d]);
          else t[d] = (x - izl) / a.step[d];                                // regular.rs:339
          loc[d] = l;
        }
      }
      if (!RECT && !ok && live[h]) atomicMin(a.first_bad, (unsigned long long)(i0 + h));
      // Offsets of my point's four pieces (lower corner of the leading dims included) -> LDS,
      // transposed: lane q reads piece q of points 0..3.
      unsigned lead = 0;
#pragma unroll
      for (int d = 0; d < L; ++d) lead += (unsigned)loc[d] * a.lead_stride[d];
      const unsigned bk = (unsigned)loc[N - 1] / (unsigned)SK;
      const unsigned kpart = bk * (unsigned)Geom::ELEMS + ((unsigned)loc[N - 1] - bk * (unsigned)SK);
      if constexpr (ABL != 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
          lds_off[(quad * 4 + p) * 4 + q] = lead + brick_piece<T, SI, SJ, CELL>(a.nbj, a.nbk, loc[N - 3], loc[N - 2], kpart, p >> 1, p & 1);
        wave_sync();
      }
      Cell<T> c;
      if constexpr (ABL == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) c.v[e >> 2][(e >> 1) & 1][e & 1] = t[e % N] + (T)(kpart + lead + (unsigned)e);
      } else {
        const uint4 toff = *reinterpret_cast<const uint4*>(&lds_off[(quad * 4 + q) * 4]);
        // offsets between the two footprint values of each leading dimension: whole brick tables,
        // except the h dimension of 4-D cell bricks, whose pair sits inside the brick
        unsigned ls[L > 0 ? L : 1];
#pragma unroll
        for (int d = 0; d < L; ++d) ls[d] = (CELL == 1 && d == L - 1) ? (unsigned)Geom::IJ : a.lead_stride[d];
        c = LeadReduce<T, L, FMA>::run(a.bricks, toff, 0u, ls, t, lds_piece, quad, q);
      }
      // Trailing three dims, reference order (multilinear/regular.rs:347-403): i first, k last.
      T r[2];
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        const T c0 = mul_add<FMA>(t[N - 3], c.v[1][0][dk] - c.v[0][0][dk], c.v[0][0][dk]);
        const T c1 = mul_add<FMA>(t[N - 3], c.v[1][1][dk] - c.v[0][1][dk], c.v[0][1][dk]);
        r[dk] = mul_add<FMA>(t[N - 2], c1 - c0, c0);
      }
      resv[h] = mul_add<FMA>(t[N - 1], r[1] - r[0], r[0]);
    }
    if constexpr (ABL == 2) {
      // keep the results alive without a store the memory system would see
      if (resv[0] == (T)1234567.25 && resv[PPL - 1] == (T)-7654321.5) a.out[i0] = resv[0];
    } else if (PPL >= 2) {
      typedef T TV __attribute__((ext_vector_type(PPL)));
      if (live[PPL - 1]) {
        TV v;
#pragma unroll
        for (int h = 0; h < PPL; ++h) v[h] = resv[h];
        stream_store(reinterpret_cast<TV*>(a.out + i0), v);
      } else {
#pragma unroll
        for (int h = 0; h < PPL; ++h)
          if (live[h]) stream_store(a.out + i0 + h, resv[h]);
      }
    } else if (live[0]) {
      stream_store(a.out + i0, resv[0]);
    }
  }
}

// Brick table builder: one thread per brick element.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_bricks(const T* __restrict__ vals, T* __restrict__ bricks, size_t nlead,
                                                         int n0, int n1, int n2, int si, int sj, unsigned nbi, unsigned nbj,
                                                         unsigned nbk) {
  constexpr int KW = BrickGeom<T>::KW;
  constexpr int EL = BrickGeom<T>::ELEMS;
  const size_t per_lead = (size_t)nbi * nbj * nbk * EL;
  const size_t total = nlead * per_lead;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const size_t lead = e / per_lead;
    size_t b = e - lead * per_lead;
    const unsigned within = (unsigned)(b % EL);
    b /= EL;
    const unsigned bk = (unsigned)(b % nbk); b /= nbk;
    const unsigned bj = (unsigned)(b % nbj); b /= nbj;
    const unsigned bi = (unsigned)b;
    const int i = (int)bi * si + (int)(within / (2 * KW));
    const int j = (int)bj * sj + (int)((within / KW) & 1);
    const int k = (int)bk * (KW - 1) + (int)(within % KW);
    T v = (T)0;
    if (i < n0 && j < n1 && k < n2) v = vals[((lead * n0 + i) * n1 + j) * n2 + k];
    bricks[e] = v;
  }
}

// f32 2 x 4 x 4 bricks (CELL == 2): table[lead][bi][bj][bk][oi 0..1][oj 0..3][ok 0..3], steps (1, 3, 3).
template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_j4_bricks(const T* __restrict__ vals, T* __restrict__ bricks, size_t nlead,
                                                            int n0, int n1, int n2, unsigned nbi, unsigned nbj, unsigned nbk) {
  const size_t per_lead = (size_t)nbi * nbj * nbk * 32;
  const size_t total = nlead * per_lead;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const size_t lead = e / per_lead;
    size_t b = e - lead * per_lead;
    const unsigned within = (unsigned)(b & 31);
    b >>= 5;
    const unsigned bk = (unsigned)(b % nbk); b /= nbk;
    const unsigned bj = (unsigned)(b % nbj); b /= nbj;
    const unsigned bi = (unsigned)b;
    const int i = (int)bi + (int)(within >> 4);
    const int j = (int)bj * 3 + (int)((within >> 2) & 3);
    const int k = (int)bk * 3 + (int)(within & 3);
    T v = (T)0;
    if (i < n0 && j < n1 && k < n2) v = vals[((lead * n0 + i) * n1 + j) * n2 + k];
    bricks[e] = v;
  }
}

// 4-D cell bricks: table[lead][bh][bi][bj][bk][dh][di][dj][KW], all steps 1 except KW-1 along k.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_cell_bricks(const T* __restrict__ vals, T* __restrict__ bricks, size_t nlead,
                                                              int nh, int n0, int n1, int n2, unsigned nbh, unsigned nbi,
                                                              unsigned nbj, unsigned nbk) {
  typedef BrickGeom<T, 1> G;
  const size_t per_lead = (size_t)nbh * nbi * nbj * nbk * G::ELEMS;
  const size_t total = nlead * per_lead;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const size_t lead = e / per_lead;
    size_t b = e - lead * per_lead;
    const unsigned within = (unsigned)(b % G::ELEMS);
    b /= G::ELEMS;
    const unsigned bk = (unsigned)(b % nbk); b /= nbk;
    const unsigned bj = (unsigned)(b % nbj); b /= nbj;
    const unsigned bi = (unsigned)(b % nbi); b /= nbi;
    const unsigned bh = (unsigned)b;
    const int h = (int)bh + (int)(within / G::IJ);
    const unsigned wij = within % G::IJ;
    const int i = (int)bi + (int)(wij / (2 * G::KW));
    const int j = (int)bj + (int)((wij / G::KW) & 1);
    const int k = (int)bk * G::SK + (int)(wij % G::KW);
    T v = (T)0;
    if (h < nh && i < n0 && j < n1 && k < n2) v = vals[(((lead * nh + h) * n0 + i) * n1 + j) * n2 + k];
    bricks[e] = v;
  }
}

}  // namespace interpn
