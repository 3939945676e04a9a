// Runtime-N kernels: the recursive arms of the reference's `interpn` dispatch
// (multilinear N = 7,8; multicubic N = 5..8) and grids too large for 32-bit indexing.
#include <cstdlib>

#include "interpn_kernels.h"

namespace interpn {

template <typename T, int METHOD, int KIND, bool FMA>
static hipError_t launch_mk(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                            unsigned long long* first_bad, hipStream_t stream) {
  GenericArgs<T> a;
  a.vals = static_cast<const T*>(g.vals);
  a.out = out;
  a.first_bad = first_bad;
  a.npts = npts;
  a.ndims = g.ndims;
  a.linearize = g.linearize;
  // FMA sites that differ between the reference's flattened and recursive arms:
  a.fma_index = g.ndims <= 6;   // multilinear/regular.rs:337 vs regular_recursive.rs:310-313
  a.fma_linear = g.ndims >= 5;  // multicubic/rectilinear.rs:500,539 vs rectilinear_recursive.rs:467,527
  unsigned long long acc = 1;
  for (int d = kMaxDims - 1; d >= 0; --d) {
    if (d < g.ndims) {
      a.obs[d] = obs[d];
      a.start[d] = (T)g.start[d];
      a.step[d] = (T)g.step[d];
      a.grid[d] = static_cast<const T*>(g.grid[d]);
      a.n[d] = g.n[d];
      a.stride[d] = acc;
      acc *= (unsigned long long)g.n[d];
    } else {
      a.obs[d] = nullptr;
      a.start[d] = (T)0;
      a.step[d] = (T)1;
      a.grid[d] = nullptr;
      a.n[d] = 0;
      a.stride[d] = 0;
    }
  }
  const unsigned blocks = grid_blocks(npts, 1, g.cfg);
  // Compile-time-N form for the dimension counts the reference's recursive arms serve;
  // INTERPN_HIP_GENERIC_RUNTIME=1 keeps the runtime-N form (testing).
  const char* env = getenv("INTERPN_HIP_GENERIC_RUNTIME");
  const bool runtime_n = env && env[0] == '1';
  // Row-vector form (FP trees side by side) where its FP x larger register footprint still fits
  // the 256 architectural VGPRs; beyond that the compiler spills into AGPRs, which measured
  // slower (cubic rectilinear N >= 6 in f64) and, for f64 cubic regular N = 8, gave wrong
  // results on ROCm 7.2 — those shapes keep the one-tree form.  INTERPN_HIP_GENERIC_VEC=0|1
  // overrides (testing).
  bool vec = true;
  if (METHOD == kCubic) {
    const bool f32 = sizeof(T) == 4;
    vec = KIND == kRegular ? (g.ndims <= 7 || f32) : (g.ndims == 5 || (f32 && g.ndims == 6));
  }
  if (const char* venv = getenv("INTERPN_HIP_GENERIC_VEC")) vec = venv[0] != '0';
#define GO_N(NN)                                                                                                   \
  do {                                                                                                             \
    if (vec) hipLaunchKernelGGL((k_generic_n<T, METHOD, KIND, FMA, NN, true>), dim3(blocks), dim3(kBlock), 0, stream, a);  \
    else hipLaunchKernelGGL((k_generic_n<T, METHOD, KIND, FMA, NN, false>), dim3(blocks), dim3(kBlock), 0, stream, a);     \
  } while (0)
  if (!runtime_n && g.ndims == 8) GO_N(8);
  else if (!runtime_n && g.ndims == 7) GO_N(7);
  else if (!runtime_n && METHOD == kCubic && g.ndims == 6) GO_N(6);
  else if (!runtime_n && METHOD == kCubic && g.ndims == 5) GO_N(5);
  else hipLaunchKernelGGL((k_generic<T, METHOD, KIND, FMA>), dim3(blocks), dim3(kBlock), 0, stream, a);
#undef GO_N
  return hipGetLastError();
}

template <typename T>
hipError_t launch_generic(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                          unsigned long long* first_bad, hipStream_t stream) {
#define PICK(M, K)                                                                          \
  return g.fma ? launch_mk<T, M, K, true>(g, obs, out, npts, first_bad, stream)            \
               : launch_mk<T, M, K, false>(g, obs, out, npts, first_bad, stream)
  if (g.method == kLinear) {
    if (g.kind == kRegular) PICK(kLinear, kRegular);
    PICK(kLinear, kRectilinear);
  }
  if (g.kind == kRegular) PICK(kCubic, kRegular);
  PICK(kCubic, kRectilinear);
#undef PICK
}

template hipError_t launch_generic<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t);
template hipError_t launch_generic<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t);

}  // namespace interpn
