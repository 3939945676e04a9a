// check_bounds as a streaming any-reduction (reference: src/multilinear/regular.rs:145-182,
// rectilinear.rs:109-134): one coalesced pass over a coordinate array, HBM-bound.
#include "interpn_kernels.h"

namespace interpn {

template <typename T>
__global__ void __launch_bounds__(kBlock) k_check_bounds(const T* __restrict__ x, size_t n, T lo, T hi, T atol,
                                                         unsigned* flag) {
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
    const T v = x[i];
    bad |= ((v - lo) <= -atol) || ((v - hi) >= atol);  // regular.rs:170
  }
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

template <typename T>
hipError_t launch_check_bounds(const T* x, size_t n, T lo, T hi, T atol, unsigned* flag, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  LaunchConfig cfg;
  const unsigned blocks = grid_blocks(n, 4, cfg);
  hipLaunchKernelGGL((k_check_bounds<T>), dim3(blocks), dim3(kBlock), 0, stream, x, n, lo, hi, atol, flag);
  return hipGetLastError();
}

// tab[b] = first k with bucket_of(g[k]) >= b, b = 0..M (bucket_of is non-decreasing in k for a
// sorted axis, so a binary search finds it); tab[M] = n.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_buckets(const T* __restrict__ g, int n, int M, T g0, T scale,
                                                          unsigned* __restrict__ tab) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b > M) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (bucket_of<T>(g[mid], g0, scale, M) < b) lo = mid + 1; else hi = mid;
  }
  tab[b] = (unsigned)lo;
}

template <typename T>
hipError_t build_buckets(const T* g, int n, int M, T g0, T scale, unsigned* tab, hipStream_t stream) {
  const unsigned blocks = (unsigned)((M + 1 + kBlock - 1) / kBlock);
  hipLaunchKernelGGL((k_build_buckets<T>), dim3(blocks), dim3(kBlock), 0, stream, g, n, M, g0, scale, tab);
  return hipGetLastError();
}
// Lane table (interpn_host.h::GridDesc::axis_ltab_off): kLaneBuckets buckets over an axis of at
// most 64 coordinates.  words[b >> 2] byte (b & 3) = number of coordinates in front of bucket b
// (b = 0..255; <= 64, fits a byte); words[64] = largest number of coordinates sharing a bucket.
// `words` must be zero on entry.
template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_lane_table(const T* __restrict__ g, int n, T g0, T scale,
                                                             unsigned* __restrict__ words) {
  const int b = threadIdx.x;  // one block of 256 threads
  auto first_at_or_after = [&](int bucket) {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (bucket_of<T>(g[mid], g0, scale, kLaneBuckets) < bucket) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  const int lo = first_at_or_after(b);
  atomicOr(&words[b >> 2], (unsigned)lo << ((b & 3) * 8));
  if (b < kLaneBuckets) atomicMax(&words[64], (unsigned)(first_at_or_after(b + 1) - lo));
}

template <typename T>
hipError_t build_lane_table(const T* g, int n, T g0, T scale, unsigned* words65, hipStream_t stream) {
  hipLaunchKernelGGL((k_build_lane_table<T>), dim3(1), dim3(kBlock), 0, stream, g, n, g0, scale, words65);
  return hipGetLastError();
}
template hipError_t build_lane_table<double>(const double*, int, double, double, unsigned*, hipStream_t);
template hipError_t build_lane_table<float>(const float*, int, float, float, unsigned*, hipStream_t);

template hipError_t build_buckets<double>(const double*, int, int, double, double, unsigned*, hipStream_t);
template hipError_t build_buckets<float>(const float*, int, int, float, float, unsigned*, hipStream_t);

template hipError_t launch_check_bounds<double>(const double*, size_t, double, double, double, unsigned*, hipStream_t);
template hipError_t launch_check_bounds<float>(const float*, size_t, float, float, float, unsigned*, hipStream_t);

}  // namespace interpn
