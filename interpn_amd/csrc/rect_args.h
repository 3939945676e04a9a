// Shared argument packing for the rectilinear launchers.
#pragma once
#include <cstdlib>

#include "interpn_kernels.h"

namespace interpn {

// `big_lds`: the kernel has no other LDS use (C-order, 2-D brick and nearest kernels) and N <= 2,
// where the axes are long and the search is what the kernel spends its time on: the image may
// then take up to 60 KiB (measured: 2-D 1000^2 rectilinear 2.22 -> 1.66 ms, 1-D 3000 1.22 ->
// 0.64 ms per 1e8 points against the search through L1/L2).
// `records`: the kernel searches with axis_cell (multilinear, nearest) and may be given the
// per-bucket records instead of coordinates + tables.
template <typename T, int N>
inline size_t fill_axis_args(const GridDesc& g, AxisArgs<T, N>& ax, bool big_lds = false, bool records = false) {
  ax.image = static_cast<const unsigned char*>(g.axis_image);
  ax.image_bytes = g.axis_image_bytes;
  ax.use_rec = 0;
  for (int d = 0; d < N; ++d) ax.rec_off[d] = 0;
  for (int d = 0; d < N; ++d) {
    ax.g_off[d] = g.axis_g_off[d];
    ax.tab_off[d] = g.axis_tab_off[d];
    ax.n[d] = g.n[d];
    ax.M[d] = g.axis_buckets[d];
    ax.g0[d] = (T)g.axis_g0[d];
    ax.scale[d] = (T)g.axis_scale[d];
    ax.ltab_off[d] = g.axis_ltab_off[d];
    ax.lscale[d] = (T)g.axis_lscale[d];
  }
  const Thresholds th = thresholds(g.cfg);
  size_t cap = (big_lds && N <= 2) ? th.axis_lds_wide : th.axis_lds;
  if (g.cfg.axis_lds_kb >= 0 && g.cfg.axis_lds_kb <= 60) cap = (size_t)g.cfg.axis_lds_kb * 1024;  // tuning knob
  bool lanes = g.cfg.axis_regs != 0;  // the lane-resident search (lane_axes.h) takes axes of <= 64 coordinates
  for (int d = 0; d < N; ++d) lanes = lanes && g.n[d] <= 64;
  if (records && !lanes && g.cfg.axis_records != 0 && g.axis_rec_bytes && g.axis_rec_bytes <= cap) {
    ax.use_rec = g.axis_rec_compact ? 2 : 1;
    ax.image += g.axis_rec_base;
    ax.image_bytes = g.axis_rec_bytes;
    for (int d = 0; d < N; ++d) {
      ax.rec_off[d] = g.axis_rec_off[d] - g.axis_rec_base;
      if (g.axis_rec_compact) ax.g_off[d] = g.axis_recg_off[d] - g.axis_rec_base;  // the region's own copy of the coordinates
    }
    ax.use_lds = 1;
    return g.axis_rec_bytes;
  }
  ax.use_lds = g.axis_image_bytes <= cap;
  return ax.use_lds ? g.axis_image_bytes : 0;
}

template <typename T, int N>
inline size_t fill_rect_args(const GridDesc& g, const T* const* obs, T* out, size_t npts, RectArgs<T, N>& a) {  // C-order kernels
  a.vals = static_cast<const T*>(g.vals);
  a.out = out;
  a.npts = npts;
  a.linearize = g.linearize;
  unsigned acc = 1;
  for (int d = N - 1; d >= 0; --d) {
    a.stride[d] = acc;
    acc *= (unsigned)g.n[d];
  }
  for (int d = 0; d < N; ++d) a.obs[d] = obs[d];
  return fill_axis_args<T, N>(g, a.ax, /*big_lds=*/true, /*records=*/true);
}

}  // namespace interpn
