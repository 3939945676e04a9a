// 3-D multicubic on SORTED points with the table column of a cell resident in LDS (round 5): the
// N = 3 counterpart of cubic_column.h.
//
// After the counting sort of k_bin_points.hip with one bin per pair of saturation classes of dims 0
// and 1, all points of a bin read the SAME 4 x 4 (i, j) footprint of every k plane: n2 tiles of the
// fully overlapped tile table (cubic_brick.h), 128 B each in f64 — 8 KiB for a 64^3 grid, against
// 128 KiB for cfg4's 32 x 32 planes.  Evaluated in place a point reads its four tiles as four lines
// of a table 16x the grid (64^3: 33.5 MiB, every line an L2 miss: 0.60 ms per 1e7 points at the
// fabric's line rate); here a workgroup loads the bin's column once (64 lines for ~2500 points) and
// every point takes its four tiles from LDS.
//
// Shape: 256-thread workgroups, several per CU, persistent (parts drawn from the counter the sort's
// scan leaves at zero).  Per part: look the part up, load the column (16-byte pieces, coalesced per
// tile), then one point per lane and step — record, per-dimension set-up, four tiles, the
// reference's 21 nodes — in the order the sort left the points in (no local sort: the column is a
// single row of tiles, a tile's LDS pitch of 144 / 80 bytes spreads the lanes' reads over the bank
// groups).  The workgroup synchronises with s_barrier (three per part, every thread of the
// workgroup at every one of them; the loop's exit is workgroup-uniform).
//
// Arithmetic, plane order and reduction tree are those of cubic_brick.h / the reference
// (src/multicubic/regular.rs:325-623, rectilinear.rs:265-545): bit-identical results.  A point whose
// exact footprint cell is not the part's (the sort estimates classes on regular grids; any point may
// be mis-binned on purpose: option bin_scramble) is evaluated from the table in global memory by
// the same tree.
#pragma once
#include "cubic_column.h"

namespace interpn {

template <typename T>
struct Cubic3ColumnArgs {
  const T* tiles;            // fully overlapped tile table [plane k][bi][bj][16]
  unsigned table_bytes;
  const T* records;          // the slice's points in bin order, FOUR elements each (x0, x1, x2, x2: the 4-D sort's record form)
  const unsigned* index;     // sorted position -> index within the slice
  T* out;
  unsigned long long* first_bad;
  size_t index_base;
  const unsigned* bin_end;      // end of bin b in sorted order
  const unsigned* part_prefix;  // parts in front of bin b; [nbins] = total
  unsigned* work;               // next part to hand out (zeroed by the sort's scan kernel)
  int nbins;
  int nb1;                   // classes along dim 1 (n1 - 1)
  unsigned inv_mult;         // sorted bin b holds class pair (b * inv_mult) % nbins
  T start[3];
  T step[3];
  int n[3];
  int linearize;
  unsigned plane_stride;     // table elements per unit index of dim 2
  unsigned nbj;
  AxisArgs<T, 3> ax;         // RECT: the handle's axis image, staged behind the column when it fits (ax.use_lds)
  unsigned axes_lds_off;
};

// One point from the table in global memory: the reference's tree on the point's own footprint.
template <typename T, bool RECT, bool FMA>
__device__ __noinline__ T col3_slow_point(__amdgpu_buffer_rsrc_t rsrc, unsigned tile_off_bytes, unsigned ps_bytes,
                                          const typename CubicDimSel<T, RECT>::type* dim) {
  T s[4];
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    T v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = table_load<T>(rsrc, tile_off_bytes + (unsigned)r * ps_bytes + (unsigned)e * (unsigned)sizeof(T), 0u);
    s[r] = reduce_tile<T, RECT, FMA>(v, dim, 0u);
  }
  return cubic_node_sel<RECT, FMA, T>(s[0], s[1], s[2], s[3], dim[2]);
}

template <typename T, bool RECT, bool FMA>
__global__ void __launch_bounds__(kBlock) k_cubic3_column(const Cubic3ColumnArgs<T> a) {
  typedef typename CubicDimSel<T, RECT>::type DimT;
  constexpr unsigned PITCH = col_pitch<T>();               // tile + 16 bytes
  constexpr unsigned PP = (unsigned)sizeof(T);             // 16-byte pieces of a tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c3[];
  __shared__ unsigned s_ctl[8];  // 0 part | 1 bin | 2 begin | 3 end
  const unsigned tid = threadIdx.x;
  const unsigned char* axis_base = a.ax.image;
  if constexpr (RECT) {
    if (a.ax.use_lds) {
      const unsigned words = a.ax.image_bytes >> 2;
      const unsigned* src = reinterpret_cast<const unsigned*>(a.ax.image);
      unsigned* dst = reinterpret_cast<unsigned*>(smem_c3 + a.axes_lds_off);
      for (unsigned k = tid; k < words; k += kBlock) dst[k] = src[k];
      axis_base = smem_c3 + a.axes_lds_off;
    }
  }
  const unsigned total_parts = a.part_prefix[a.nbins];
  const __amdgpu_buffer_rsrc_t rsrc = table_rsrc(a.tiles, a.table_bytes);
  const unsigned ps_bytes = a.plane_stride * (unsigned)sizeof(T);
  const unsigned lds_col = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_c3;
  typedef T RV __attribute__((ext_vector_type(4)));
  typedef unsigned U4 __attribute__((ext_vector_type(4)));
  for (;;) {
    __syncthreads();  // every thread is done with the previous part's column and control words
    if (tid == 0) s_ctl[0] = atomicAdd(a.work, 1u);
    __syncthreads();
    const unsigned w = s_ctl[0];
    if (w >= total_parts) break;  // workgroup-uniform
    // Which (bin, part)?  part_prefix is non-decreasing: the bin b with part_prefix[b] <= w < part_prefix[b + 1].
    for (int b = (int)tid; b < a.nbins; b += kBlock) {
      const unsigned p0 = a.part_prefix[b], p1 = a.part_prefix[b + 1];
      if (p0 <= w && w < p1) {
        const unsigned b0 = b ? a.bin_end[b - 1] : 0u;
        const unsigned b1 = a.bin_end[b];
        const unsigned cnt = b1 - b0, nparts = p1 - p0, j = w - p0;
        const unsigned per = (cnt + nparts - 1) / nparts;
        const unsigned lo = b0 + j * per;
        unsigned hi = lo + per;
        if (hi > b1) hi = b1;
        s_ctl[1] = (unsigned)b;
        s_ctl[2] = lo < b1 ? lo : b1;
        s_ctl[3] = hi;
      }
    }
    __syncthreads();
    const unsigned bin = s_ctl[1], begin = s_ctl[2], end = s_ctl[3];
    if (begin >= end) continue;  // workgroup-uniform
    const unsigned key = (unsigned)(((unsigned long long)bin * a.inv_mult) % (unsigned)a.nbins);
    const int c0 = (int)(key / (unsigned)a.nb1), c1 = (int)(key % (unsigned)a.nb1);  // nominal classes of dims 0, 1
    const int ci = c0 - 1 < 0 ? 0 : (c0 - 1 > a.n[0] - 4 ? a.n[0] - 4 : c0 - 1);      // their footprint cell
    const int cj = c1 - 1 < 0 ? 0 : (c1 - 1 > a.n[1] - 4 ? a.n[1] - 4 : c1 - 1);
    const unsigned cell_off = (unsigned)(ci * (int)a.nbj + cj) * 16u * (unsigned)sizeof(T);  // the cell's tile inside a plane, bytes
    // the column: tile k = the cell's tile of plane k, 16-byte pieces, PP consecutive lanes per tile
    {
      const unsigned units = (unsigned)a.n[2] * PP;
      const unsigned char* tb = reinterpret_cast<const unsigned char*>(a.tiles) + cell_off;
      for (unsigned u = tid; u < units; u += kBlock) {
        const unsigned k = u / PP, piece = u % PP;
        const U4 v = *reinterpret_cast<const U4*>(tb + (size_t)k * ps_bytes + piece * 16u);
        *reinterpret_cast<U4*>(smem_c3 + k * PITCH + piece * 16u) = v;
      }
    }
    __syncthreads();
    const RV* __restrict__ recs = reinterpret_cast<const RV*>(a.records);
    for (unsigned p = begin + tid; p < end; p += kBlock) {
      const RV rec = recs[p];
      const unsigned orig = a.index[p];
      const T x[3] = {rec[0], rec[1], rec[2]};
      DimT dim[3];
      int loc[3];
      bool ok = true;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if constexpr (RECT) {
          const Axis<T> ax = make_axis<T, 3>(a.ax, axis_base, d);
          loc[d] = cubic_rect_locate<T>(ax, x[d], a.linearize, /*fma_linear=*/false, dim[d]);  // multicubic/rectilinear.rs:366-408
        } else {
          T floc;
          ok &= regular_floc<T>(x[d], a.start[d], a.step[d], &floc);   // multicubic/regular.rs:435-438
          ok &= floc != (T)-9223372036854775808.0;                     // `- 1` would overflow isize
          const T nn = (T)a.n[d];
          const int l = clamp_loc<T>(floc - (T)1, a.n[d] - 4);         // regular.rs:440-442
          int sat;
          bool outside;
          if (floc < (T)0) { sat = kSatLow; outside = true; }          // regular.rs:445-466 on floc = iloc + 1
          else if (floc == (T)0) { sat = kSatLow; outside = false; }
          else if (floc > nn - (T)2) { sat = kSatHigh; outside = true; }
          else if (floc == nn - (T)2) { sat = kSatHigh; outside = false; }
          else { sat = kSatNone; outside = false; }
          const T index_one_loc = mul_add<false>(a.step[d], (T)(l + 1), a.start[d]);  // regular.rs:356-360, never fused
          const T t = (x[d] - index_one_loc) / a.step[d];
          dim[d].sat = sat;
          dim[d].linear = (outside && a.linearize) ? 1 : 0;
          dim[d].tt = sat == kSatLow ? -t : (sat == kSatHigh ? t - (T)1 : t);
          loc[d] = l;
        }
      }
      if (!RECT && !ok) atomicMin(a.first_bad, (unsigned long long)(a.index_base + orig));
      T res;
      if (loc[0] == ci && loc[1] == cj) {
        T s[4];
        const unsigned a0 = lds_col + (unsigned)loc[2] * PITCH;
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
          T v[16];
          col_take_tile<T>(a0 + (unsigned)r * PITCH, v);
          s[r] = reduce_tile<T, RECT, FMA>(v, dim, 0u);
        }
        res = cubic_node_sel<RECT, FMA, T>(s[0], s[1], s[2], s[3], dim[2]);  // regular.rs:415-421 / rectilinear.rs:346-355
      } else {  // not this part's cell: the point's own footprint from the table in global memory
        const unsigned off = ((unsigned)loc[2] * a.plane_stride + (unsigned)(loc[0] * (int)a.nbj + loc[1]) * 16u) * (unsigned)sizeof(T);
        res = col3_slow_point<T, RECT, FMA>(rsrc, off, ps_bytes, dim);
      }
      stream_store(a.out + orig, res);
    }
  }
}

}  // namespace interpn
