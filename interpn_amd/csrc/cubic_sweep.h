// 3-D (and, with N = 2, 2-D) multicubic on the fully overlapped tile table with the points of every wave ordered on chip by
// their cell index along the table's slowest dimension, and all waves of the chip walking that index in
// step with a clock: the sweep evaluation of linear_sweep.h (read its head first: the window argument,
// the per-wave sort, the clock, the rounds dealt on demand are the same) with cubic_brick.h's rows.
//
// Why: a 3-D cubic point reads four tile lines (one per plane of its footprint along dim 2); on a 64^3
// f64 grid the one-tile-per-footprint table is 30 MiB and unordered points miss the L2 on every one of
// them: 5.1 GB over the fabric per 1e7 points, 0.60-0.65 ms.  The table is stored
// plane by plane (dim 2 slowest: cubic_brick.h), so points ordered by their dim-2 cell inside the window
// the chip holds make an XCD fetch every plane once per window and read it from its L2 four times per
// point.  The column evaluation of sorted points (cubic3_column.h) needs a global sort that costs more
// than it saves at this size; here the order is made on chip, in the pass that evaluates.
//
// The rows are cubic_brick.h's, instruction for instruction: per-dimension cell, saturation class and t
// (multicubic/regular.rs:435-466, rectilinear.rs:366-408), the plane's 64 tiles by LDS-DMA into the
// wave's image, the reference's reduction tree (regular.rs:368-421).  A point's result depends on its
// coordinates only: the same bits whatever the order, the period or the deal.
#pragma once

#include "cubic_brick.h"
#include "sweep_rounds.h"

namespace interpn {

// N = 3: the points are ordered by their dim-2 cell (the tile table is stored plane by plane); N = 2: one plane, its tiles stored
// row of tiles by row of tiles: ordered by their dim-0 cell.
template <typename T, int N = 3>
struct CubicSweepArgs {
  SweepRounds<T, N> r;     // sweep_rounds.h: the streams, the sort key, the rounds
  CubicBrickArgs<T, N> c;  // bricks (steps 1,1), table_bytes, first_bad, start, step, n, ax, plane_stride, nbj, linearize (obs / out / npts: in `r`)
  T rstep[N];              // regular grids: RN(1 / step[d]), a division in T on the host (interpn_device.h: floor_quotient_fast / divide_fast)
  unsigned fastdiv;        // != 0: every step lies where those forms are the reference's values (StepCellRange<T>)
};

// LDS per wave: the scaffold's row buffer is also the tile image of cubic_brick.h's LDS-DMA gather (64 tiles: 8 KiB f64, 4 KiB f32).
template <typename T, int K, int KL, int N = 3>
using CubicSweepLds = SweepRoundsLds<T, N, K, KL, 64u * (unsigned)sizeof(T) * 16u>;

template <typename T, bool RECT, bool FMA, int K, int KL, int THREADS, int N = 3>
__global__ void __launch_bounds__(THREADS) k_cubic_sweep(const CubicSweepArgs<T, N> s) {
  static_assert(N == 2 || N == 3, "tiled multicubic sweep: N = 2, 3");
  typedef typename CubicDimSel<T, RECT>::type DimT;
  typedef CubicSweepLds<T, K, KL, N> L;
  const CubicBrickArgs<T, N>& a = s.c;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wave = threadIdx.x >> 6;
  unsigned char* const lds_axes = smem_raw + (THREADS / 64) * L::kWave + L::kWorkgroup;
  if constexpr (RECT) {  // (the scaffold's barrier follows)
    if (a.ax.use_lds) {
      const unsigned words = a.ax.image_bytes >> 2;
      const unsigned* src = reinterpret_cast<const unsigned*>(a.ax.image);
      unsigned* dst = reinterpret_cast<unsigned*>(lds_axes);
      for (unsigned k = threadIdx.x; k < words; k += THREADS) dst[k] = src[k];
    }
  }
  const unsigned char* axis_base = (RECT && a.ax.use_lds) ? lds_axes : a.ax.image;
  const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(a.bricks, a.table_bytes);
  // LDS byte address of this wave's tile image (= its row buffer), in a scalar register
  const unsigned lds_wave = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw + wave * L::kWave));
  sweep_rounds<T, N, K, KL, THREADS, (N == 3 ? 2 : 0), L>(s.r, smem_raw, [&](auto, const T (&xr)[N], size_t gi) -> T {
    DimT dim[N];
    int loc[N];
    bool ok = true;
    if constexpr (RECT) {
#pragma unroll
      for (int d = 0; d < N; ++d) {
        const Axis<T> ax = make_axis<T, N>(a.ax, axis_base, d);
        if (a.crec[0]) loc[d] = cubic_rect_locate_rec<T>(ax, a.crec[d], xr[d], a.linearize, dim[d]);  // (wave-uniform)
        else loc[d] = cubic_rect_locate<T, true>(ax, xr[d], a.linearize, /*fma_linear=*/false, dim[d]);  // multicubic/rectilinear.rs:366-408
      }
    } else {
      // cell, saturation class and t of the three dimensions: first without the six divide sequences
      // (interpn_device.h::floor_quotient_fast / divide_fast: the same bits where their conditions hold), then, if a
      // lane of the wave is on a grid plane, not finite or far out, once more as the reference writes them
      auto locate_all = [&](auto fast_c) -> bool {
        constexpr bool FAST = decltype(fast_c)::value;
        bool exact = true;
        ok = true;
#pragma unroll
        for (int d = 0; d < N; ++d) {
          const T xx = xr[d];
          T floc;
          if constexpr (FAST) {
            exact = floor_quotient_fast(xx - a.start[d], s.rstep[d], &floc) && exact;  // (|floc| < 2^31: representable)
          } else {
            ok &= regular_floc<T>(xx, a.start[d], a.step[d], &floc);   // multicubic/regular.rs:435-438
            ok &= floc != (T)-9223372036854775808.0;                   // `- 1` would overflow isize
          }
          const T nn = (T)a.n[d];
          const int l = clamp_loc<T>(floc - (T)1, a.n[d] - 4);       // regular.rs:440-442
          int sat;
          bool outside;
          if (floc < (T)0) { sat = kSatLow; outside = true; }        // regular.rs:445-466 on floc = iloc + 1
          else if (floc == (T)0) { sat = kSatLow; outside = false; }
          else if (floc > nn - (T)2) { sat = kSatHigh; outside = true; }
          else if (floc == nn - (T)2) { sat = kSatHigh; outside = false; }
          else { sat = kSatNone; outside = false; }
          const T index_one_loc = mul_add<false>(a.step[d], (T)(l + 1), a.start[d]);  // regular.rs:356-360, never fused
          T t;
          if constexpr (FAST) exact = divide_fast(xx - index_one_loc, a.step[d], s.rstep[d], &t) && exact;
          else t = (xx - index_one_loc) / a.step[d];
          dim[d].sat = sat;
          dim[d].linear = (outside && a.linearize) ? 1 : 0;
          dim[d].tt = sat == kSatLow ? -t : (sat == kSatHigh ? t - (T)1 : t);
          loc[d] = l;
        }
        return exact;
      };
      bool exact = false;
      if (s.fastdiv) exact = locate_all(std::true_type{});
      if (__any(!exact)) (void)locate_all(std::false_type{});
      if (!ok && gi < s.r.npts) atomicMin(a.first_bad, (unsigned long long)gi);
    }
    // my point's tile (steps 1,1: tile index = cell) as a byte offset; instruction q of a plane's DMA has me
    // fetch piece c of point p (cubic_brick.h::dma_issue_plane)
    constexpr unsigned PP = (unsigned)sizeof(T);
    unsigned tb = (unsigned)(loc[0] * (int)a.nbj + loc[1]) * 16u;
    if constexpr (N == 3) tb += (unsigned)loc[2] * a.plane_stride[2];
    tb *= (unsigned)sizeof(T);
    unsigned toff[PP];
#pragma unroll
    for (int q = 0; q < (int)PP; ++q) {
      const unsigned p = ((unsigned)q * 64u + lane) / PP;
      const unsigned c = ((lane & (PP - 1u)) - cubic_dma_rot<T>(p)) & (PP - 1u);
      toff[q] = (unsigned)__shfl((int)tb, (int)p) + c * 16u;
    }
    unsigned interior = 0;
    if constexpr (!RECT) {
#pragma unroll
      for (int d = 0; d < 2; ++d)
        if (__builtin_amdgcn_ballot_w64(dim[d].sat != kSatNone || dim[d].linear != 0) == 0) interior |= 1u << d;
    }
    if constexpr (RECT) {  // (cubic_brick.h: the nodes' divisions in their short form first)
      bool fast = true;
#pragma unroll
      for (int d = 0; d < N; ++d) fast = fast && dim[d].fast;
      T res = reduce_planes_dma<T, N, RECT, FMA, true>(rsrc, toff, a.plane_stride, lds_wave, lane, dim, interior, &fast);
      if (__any(!fast)) {
        if (a.crec[0]) {
#pragma unroll
          for (int d = 0; d < N; ++d)
            (void)cubic_rect_locate<T>(make_axis<T, N>(a.ax, axis_base, d), xr[d], a.linearize, /*fma_linear=*/false, dim[d]);
        }
        res = reduce_planes_dma<T, N, RECT, FMA>(rsrc, toff, a.plane_stride, lds_wave, lane, dim, interior);
      }
      return res;
    } else {
      return reduce_planes_dma<T, N, RECT, FMA>(rsrc, toff, a.plane_stride, lds_wave, lane, dim, interior);
    }
  });
}

}  // namespace interpn
