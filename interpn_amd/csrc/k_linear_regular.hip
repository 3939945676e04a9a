// multilinear::regular launchers (reference: src/multilinear/regular.rs:51-117 dispatch on ndims).
#include <cstdlib>

#include "interpn_kernels.h"

#ifndef INTERPN_U_LINEAR
#define INTERPN_U_LINEAR 2
#endif

namespace interpn {

template <typename T, int N, bool FMA>
static hipError_t launch_n(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                           unsigned long long* first_bad, hipStream_t stream) {
  RegularArgs<T, N> a;
  a.vals = static_cast<const T*>(g.vals);
  a.out = out;
  a.first_bad = first_bad;
  a.npts = npts;
  a.linearize = 0;
  unsigned acc = 1;
  for (int d = N - 1; d >= 0; --d) {
    a.obs[d] = obs[d];
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
    a.stride[d] = acc;
    acc *= (unsigned)g.n[d];
  }
  constexpr int U = INTERPN_U_LINEAR;
  const unsigned blocks = g.cfg.persistent ? grid_blocks(npts, U, g.cfg) : one_pass_blocks(npts, U);
  g.tag.set("k_linear_regular", {N, FMA, U}, 0b010u);
  hipLaunchKernelGGL((k_linear_regular<T, N, FMA, U>), dim3(blocks), dim3(kBlock), 0, stream, a);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_linear_regular(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                                 unsigned long long* first_bad, hipStream_t stream) {
#define CASE(N)                                                                         \
  case N:                                                                               \
    return g.fma ? launch_n<T, N, true>(g, obs, out, npts, first_bad, stream)           \
                 : launch_n<T, N, false>(g, obs, out, npts, first_bad, stream);
  switch (g.ndims) {
    CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6)
    default: return hipErrorInvalidValue;
  }
#undef CASE
}

template hipError_t launch_linear_regular<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t);
template hipError_t launch_linear_regular<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t);

}  // namespace interpn
