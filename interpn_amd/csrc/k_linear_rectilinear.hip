// multilinear::rectilinear launchers (reference: src/multilinear/rectilinear.rs:49-83).
#include "rect_args.h"

#ifndef INTERPN_U_LINEAR
#define INTERPN_U_LINEAR 2
#endif

namespace interpn {

template <typename T, int N, bool FMA>
static hipError_t launch_n(const GridDesc& g, const T* const* obs, T* out, size_t npts, hipStream_t stream) {
  RectArgs<T, N> a;
  const size_t lds = fill_rect_args<T, N>(g, obs, out, npts, a);
  constexpr int U = INTERPN_U_LINEAR;
  const unsigned blocks = grid_blocks(npts, U, g.cfg);
  g.tag.set("k_linear_rectilinear", {N, FMA, U}, 0b010u);
  hipLaunchKernelGGL((k_linear_rectilinear<T, N, FMA, U>), dim3(blocks), dim3(kBlock), lds, stream, a);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_linear_rectilinear(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                                     unsigned long long*, hipStream_t stream) {
#define CASE(N)                                                             \
  case N:                                                                   \
    return g.fma ? launch_n<T, N, true>(g, obs, out, npts, stream)          \
                 : launch_n<T, N, false>(g, obs, out, npts, stream);
  switch (g.ndims) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6)
    default: return hipErrorInvalidValue;
  }
#undef CASE
}

template hipError_t launch_linear_rectilinear<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t);
template hipError_t launch_linear_rectilinear<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t);

}  // namespace interpn
