// Shared argument packing for the rectilinear launchers.
#pragma once
#include "interpn_kernels.h"

namespace interpn {

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
  unsigned off = 0;
  for (int d = 0; d < N; ++d) {
    a.obs[d] = obs[d];
    a.grid[d] = static_cast<const T*>(g.grid[d]);
    a.n[d] = g.n[d];
    a.lds_off[d] = off;
    off += (unsigned)g.n[d];
  }
  const size_t lds_bytes = (size_t)off * sizeof(T);
  a.use_lds = lds_bytes <= kMaxGridLdsBytes;
  return a.use_lds ? lds_bytes : 0;
}

}  // namespace interpn
