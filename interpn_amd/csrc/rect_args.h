// Shared argument packing for the rectilinear launchers.
#pragma once
#include "interpn_kernels.h"

namespace interpn {

template <typename T, int N>
inline size_t fill_axis_args(const GridDesc& g, AxisArgs<T, N>& ax) {
  ax.image = static_cast<const unsigned char*>(g.axis_image);
  ax.image_bytes = g.axis_image_bytes;
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
  ax.use_lds = g.axis_image_bytes <= kMaxGridLdsBytes;
  return ax.use_lds ? g.axis_image_bytes : 0;
}

template <typename T, int N>
inline size_t fill_rect_args(const GridDesc& g, const T* const* obs, T* out, size_t npts, RectArgs<T, N>& a) {
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
  return fill_axis_args<T, N>(g, a.ax);
}

}  // namespace interpn
