// Rectilinear axes held in registers, one coordinate per lane, searched with cross-lane reads
// (ds_bpermute through the LDS crossbar: no LDS memory, hence none of the bank conflicts random
// 8-byte gathers on a tiny axis image suffer).  Usable when every axis has at most 64 coordinates.
// Shared by the multilinear brick kernels (N = 3..6), the 2-D brick kernel and the nearest kernel.
//
//   MODE 1  the reference's probe sequence (core::slice::partition_point's size-halving search,
//           src/multilinear/rectilinear.rs:363), six lockstep steps — valid for ANY axis the
//           reference accepts (its `new` only checks g[1] > g[0]);
//   MODE 2  a 255-bucket lane table (sorted finite axes; built at handle creation): the byte
//           count of coordinates in front of each bucket, four per lane, brackets the answer.
//           No bucket holds more than one coordinate (evenly spread axes; the host reads the
//           largest bucket population back when the table is built): one coordinate read decides
//           the cell and is itself one of the two brackets, so a search is 3 cross-lane reads
//           (table, probe, other bracket) instead of the probe sequence's 9.
//   MODE 3  the same table for clustered axes: `scan` (> 1) coordinate probes walk the bucket,
//           then the two brackets are read.
// (Two modes rather than one kernel with both paths: together they cost 110 VGPRs and two waves
// of occupancy per SIMD, cfg3 1.38 -> 1.43 ms.)
//
// Every function here must run with the WHOLE wave active (inactive lanes cannot be read): the
// callers keep dead lanes alive with a harmless coordinate and mask only the store.
#pragma once

#include "interpn_kernels.h"

namespace interpn {

template <typename T, int N>
struct LaneAxes {
  T g[N];            // g[d] = coordinate min(lane, n_d - 1) of axis d
  unsigned tab[N];   // MODE 2: this lane's word of the lane table of axis d
  unsigned scan;     // MODE 2: coordinate probes per search (wave-uniform)
};

template <typename T, int N, int MODE>
__device__ __forceinline__ LaneAxes<T, N> load_lane_axes(const AxisArgs<T, N>& ax) {
  LaneAxes<T, N> la;
  la.scan = 0;
  const int wl = (int)(threadIdx.x & 63u);
#pragma unroll
  for (int d = 0; d < N; ++d) {
    const T* g = reinterpret_cast<const T*>(ax.image + ax.g_off[d]);
    la.g[d] = g[wl < ax.n[d] ? wl : ax.n[d] - 1];
    la.tab[d] = 0;
    if constexpr (MODE >= 2) {
      const unsigned* words = reinterpret_cast<const unsigned*>(ax.image + ax.ltab_off[d]);
      la.tab[d] = words[wl];
      const unsigned pop = words[64];  // uniform
      la.scan = pop > la.scan ? pop : la.scan;
    }
  }
  la.scan = __builtin_amdgcn_readfirstlane(la.scan);
  return la;
}

// For PPL points per lane and N axes, in lockstep (every step issues PPL*N independent cross-lane
// reads): cell = clamp(partition_point(g < x) - 1, 0, n-2) and the bracketing coordinates
// x0 = g[cell], x1 = g[cell+1] (multilinear/rectilinear.rs:353-370 and :310-311).
template <typename T, int N, int PPL, int MODE>
__device__ __forceinline__ void lane_axes_locate(const AxisArgs<T, N>& ax, const LaneAxes<T, N>& la, const T (&xin)[PPL][N],
                                                 int (&cell)[PPL][N], T (&x0)[PPL][N], T (&x1)[PPL][N]) {
  if constexpr (MODE == 1) {
    int size[N];
#pragma unroll
    for (int d = 0; d < N; ++d) size[d] = ax.n[d];
#pragma unroll
    for (int h = 0; h < PPL; ++h)
#pragma unroll
      for (int d = 0; d < N; ++d) cell[h][d] = 0;
#pragma unroll
    for (int step = 0; step < 6; ++step) {  // six halving steps cover 64 coordinates
#pragma unroll
      for (int d = 0; d < N; ++d) {
        const int half = size[d] >> 1;  // 0 once size is 1: the probe then re-reads g[base] and keeps base
#pragma unroll
        for (int h = 0; h < PPL; ++h) {
          const int mid = cell[h][d] + half;
          cell[h][d] = (half > 0 && __shfl(la.g[d], mid) < xin[h][d]) ? mid : cell[h][d];
        }
        size[d] -= half;
      }
    }
#pragma unroll
    for (int h = 0; h < PPL; ++h)
#pragma unroll
      for (int d = 0; d < N; ++d) cell[h][d] += (__shfl(la.g[d], cell[h][d]) < xin[h][d]) ? 1 : 0;
  } else {
    // Coordinates in earlier buckets are < x and those in later buckets are >= x (bucket_of is
    // monotone and the table was built with it), so starting at the bucket's first coordinate
    // and stepping while g[idx] < x — at most `scan` times — lands on the count of g < x.
#pragma unroll
    for (int h = 0; h < PPL; ++h)
#pragma unroll
      for (int d = 0; d < N; ++d) {
        const int b = bucket_of<T>(xin[h][d], ax.g0[d], ax.lscale[d], kLaneBuckets);
        const unsigned w = __shfl(la.tab[d], b >> 2);
        cell[h][d] = (int)((w >> ((b & 3) * 8)) & 0xFFu);
      }
    if constexpr (MODE == 2) {
      // One coordinate probe decides the cell, and that coordinate is always one of the two
      // bracketing ones: with c = min(idx, n-1) the probed index and l the clamped cell, c is l
      // (probed = x0) or l+1 (probed = x1) in every case — idx = 0, 1..n-1, n, either outcome of
      // the compare — so only the OTHER bracket is fetched: 1 + 2 + 2 cross-lane reads per search
      // in f64 instead of 1 + 2 + 4.
      // (three lockstep levels, each issuing its PPL*N independent reads before any is consumed)
      int cidx[PPL][N];
      T gi[PPL][N];
#pragma unroll
      for (int h = 0; h < PPL; ++h)
#pragma unroll
        for (int d = 0; d < N; ++d) {
          const int n = ax.n[d];
          cidx[h][d] = cell[h][d] < n ? cell[h][d] : n - 1;
          gi[h][d] = __shfl(la.g[d], cidx[h][d]);
        }
      T other[PPL][N];
#pragma unroll
      for (int h = 0; h < PPL; ++h)
#pragma unroll
        for (int d = 0; d < N; ++d) {
          const int n = ax.n[d];
          const int idx = cell[h][d];
          int l = idx + ((idx < n && gi[h][d] < xin[h][d]) ? 1 : 0) - 1;  // partition_point - 1
          l = l > 0 ? l : 0;
          l = l < n - 2 ? l : n - 2;
          cell[h][d] = l;
          other[h][d] = __shfl(la.g[d], cidx[h][d] == l ? l + 1 : l);
        }
#pragma unroll
      for (int h = 0; h < PPL; ++h)
#pragma unroll
        for (int d = 0; d < N; ++d) {
          const bool probed_is_x0 = cidx[h][d] == cell[h][d];
          x0[h][d] = probed_is_x0 ? gi[h][d] : other[h][d];
          x1[h][d] = probed_is_x0 ? other[h][d] : gi[h][d];
        }
      return;
    }
    for (unsigned s = 0; s < la.scan; ++s) {  // uniform trip count
#pragma unroll
      for (int h = 0; h < PPL; ++h)
#pragma unroll
        for (int d = 0; d < N; ++d) {
          const int n = ax.n[d];
          const int idx = cell[h][d];
          const T gi = __shfl(la.g[d], idx < n ? idx : n - 1);
          cell[h][d] = idx + ((idx < n && gi < xin[h][d]) ? 1 : 0);
        }
    }
  }
  // ... then the cell: clamp(partition_point - 1, 0, n-2) (rectilinear.rs:365-367)
#pragma unroll
  for (int h = 0; h < PPL; ++h)
#pragma unroll
    for (int d = 0; d < N; ++d) {
      const int n = ax.n[d];
      int l = cell[h][d] - 1;
      l = l > 0 ? l : 0;
      l = l < n - 2 ? l : n - 2;
      cell[h][d] = l;
    }
#pragma unroll
  for (int h = 0; h < PPL; ++h)
#pragma unroll
    for (int d = 0; d < N; ++d) {
      x0[h][d] = __shfl(la.g[d], cell[h][d]);
      x1[h][d] = __shfl(la.g[d], cell[h][d] + 1);
    }
}

// Host side: which mode a rectilinear grid gets (0 = the axes do not fit one coordinate per lane).
// The handle's `axis_regs` option (latched from INTERPN_HIP_AXIS_REGS at creation) overrides:
// 0 | 1 | 2, where 2 (= use the lane tables) resolves to mode 2 or 3 by the largest bucket
// population and falls back to 1 when a table is missing (axis not proven sorted).
inline int lane_axes_mode(const GridDesc& g) {
  if (g.kind != kRectilinear) return 0;
  bool tables = true;
  int scan = 0;
  for (int d = 0; d < g.ndims; ++d) {
    if (g.n[d] > 64) return 0;
    tables = tables && g.axis_ltab_off[d] != 0;
    scan = g.axis_lscan[d] > scan ? g.axis_lscan[d] : scan;
  }
  int mode = tables ? (scan <= 1 ? 2 : 3) : 1;
  if (g.cfg.axis_regs == 0) mode = 0;
  else if (g.cfg.axis_regs == 1) mode = 1;
  return mode;
}

}  // namespace interpn
