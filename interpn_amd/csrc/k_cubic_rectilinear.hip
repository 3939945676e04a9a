// multicubic::rectilinear launchers (reference: src/multicubic/rectilinear.rs:54-104).
#include "rect_args.h"

namespace interpn {

template <typename T, int N, bool FMA>
static hipError_t launch_n(const GridDesc& g, const T* const* obs, T* out, size_t npts, hipStream_t stream) {
  RectArgs<T, N> a;
  const size_t lds = fill_rect_args<T, N>(g, obs, out, npts, a);
  const unsigned blocks = grid_blocks(npts, 1, g.cfg);
  g.tag.set("k_cubic_rectilinear", {N, FMA}, 0b10u);
  hipLaunchKernelGGL((k_cubic_rectilinear<T, N, FMA>), dim3(blocks), dim3(kBlock), lds, stream, a);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_cubic_rectilinear(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                                    unsigned long long*, hipStream_t stream) {
#define CASE(N)                                                             \
  case N:                                                                   \
    return g.fma ? launch_n<T, N, true>(g, obs, out, npts, stream)          \
                 : launch_n<T, N, false>(g, obs, out, npts, stream);
  switch (g.ndims) {
    CASE(1) CASE(2) CASE(3) CASE(4)
    default: return hipErrorInvalidValue;
  }
#undef CASE
}

template hipError_t launch_cubic_rectilinear<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t);
template hipError_t launch_cubic_rectilinear<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t);

}  // namespace interpn
