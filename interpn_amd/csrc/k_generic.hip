// Runtime-N kernels: the recursive arms of the reference's `interpn` dispatch
// (multilinear N = 7,8; multicubic N = 5..8) and grids too large for 32-bit indexing.
#include <cstdlib>

#include "interpn_kernels.h"

namespace interpn {

// Which (type, method, kind, N) get the row-vector form of k_generic_n compiled.  Its state is
// store[N-1][FP][FP]: for f64 multicubic that is 2 x 16 x (N-1) registers, and from N = 7 (regular)
// / N = 6 (rectilinear) on hipcc (ROCm 7.2) can no longer keep it in the 256 architectural VGPRs:
// -Rpass-analysis=kernel-resource-usage shows 255 VGPRs + 2 AGPRs (regular N = 7), 256 + 48
// (regular N = 8), 256 + 34..128 AGPRs and 12..32 scratch spills (rectilinear N = 6..8), each on
// top of 50-80 SGPRs already spilled into VGPR lanes.  The 256 + 48 shape returned wrong results
// on the GPU (f64 cubic regular N = 8, round 1) while its one-tree twin (97..190 VGPRs, no AGPR)
// is correct, i.e. the failure is in the compiler's AGPR spill code for this register-exhausted
// kernel, not in the source.  Rule: a row-vector instantiation exists only where the compiler
// reports AGPRs == 0 and ScratchSize == 0 (tests/test_build_resources.py asserts exactly that on
// every build), so neither the heuristic nor the `generic_vec` option can reach a spilled kernel.
template <typename T, int METHOD, int KIND, int N>
constexpr bool generic_vec_ok() {
  if (METHOD == kLinear || sizeof(T) == 4) return true;  // <= 209 VGPRs, no AGPRs
  return KIND == kRegular ? N <= 6 : N == 5;             // f64 multicubic: 221 / 236 VGPRs, no AGPRs
}

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
  // Compile-time-N form for the dimension counts the reference's recursive arms serve; the
  // `generic_runtime` option keeps the runtime-N form (testing).
  const bool runtime_n = g.cfg.generic_runtime != 0;
  // Row-vector form (FP trees side by side over one vector load per row) where it pays:
  // everywhere for multilinear, for multicubic while FP x the tree state still fits the register
  // file (generic_vec_ok below decides what is compiled at all).  The `generic_vec` option (0|1)
  // overrides the choice among the compiled forms.
  bool vec = true;
  if (METHOD == kCubic) {
    const bool f32 = sizeof(T) == 4;
    vec = KIND == kRegular ? true : (g.ndims == 5 || (f32 && g.ndims == 6));
  }
  if (g.cfg.generic_vec >= 0) vec = g.cfg.generic_vec != 0;
#define GO_N(NN)                                                                                                   \
  do {                                                                                                             \
    if constexpr (generic_vec_ok<T, METHOD, KIND, NN>()) {                                                         \
      if (vec) {                                                                                                   \
        g.tag.set("k_generic_n", {METHOD, KIND, FMA, NN, 1}, 0b10100u);                                            \
        hipLaunchKernelGGL((k_generic_n<T, METHOD, KIND, FMA, NN, true>), dim3(blocks), dim3(kBlock), 0, stream, a); \
        break;                                                                                                     \
      }                                                                                                            \
    }                                                                                                              \
    g.tag.set("k_generic_n", {METHOD, KIND, FMA, NN, 0}, 0b10100u);                                                \
    hipLaunchKernelGGL((k_generic_n<T, METHOD, KIND, FMA, NN, false>), dim3(blocks), dim3(kBlock), 0, stream, a);  \
  } while (0)
  if (!runtime_n && g.ndims == 8) GO_N(8);
  else if (!runtime_n && g.ndims == 7) GO_N(7);
  else if (!runtime_n && METHOD == kCubic && g.ndims == 6) GO_N(6);
  else if (!runtime_n && METHOD == kCubic && g.ndims == 5) GO_N(5);
  else {
    g.tag.set("k_generic", {METHOD, KIND, FMA}, 0b100u);
    hipLaunchKernelGGL((k_generic<T, METHOD, KIND, FMA>), dim3(blocks), dim3(kBlock), 0, stream, a);
  }
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
