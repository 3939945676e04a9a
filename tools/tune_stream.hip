// Stream-rate microbenchmark: how fast can one MI355X move the 3-read / 1-write f64 stream that
// bounds the interpolation kernels from below, and which access shape gets there?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/tune_stream tools/tune_stream.hip
// Run:   tools/tune_stream [points=1e8]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT> __device__ __forceinline__ d2 ld(const d2* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void st(d2* p, d2 v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// grid-stride, 16 B per lane
template <bool NTL, bool NTS, int MODE>  // MODE 0: 3R1W, 1: 3R (sink), 2: 1W, 3: 1R1W copy
__global__ void __launch_bounds__(256) k_gs(const d2* x, const d2* y, const d2* z, d2* o, size_t n2) {
  const size_t nthreads = (size_t)gridDim.x * 256;
  d2 acc = {0.0, 0.0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += nthreads) {
    if constexpr (MODE == 0) { d2 a = ld<NTL>(x + i), b = ld<NTL>(y + i), c = ld<NTL>(z + i); st<NTS>(o + i, a + b + c); }
    if constexpr (MODE == 1) { acc += ld<NTL>(x + i) + ld<NTL>(y + i) + ld<NTL>(z + i); }
    if constexpr (MODE == 2) { d2 v = {(double)i, 1.0}; st<NTS>(o + i, v); }
    if constexpr (MODE == 3) { st<NTS>(o + i, ld<NTL>(x + i)); }
  }
  if constexpr (MODE == 1) if (acc.x == 123.456) o[0] = acc;
}

// block-contiguous chunks: block b owns elements [b*chunk, (b+1)*chunk); unrolled by U wave-rows
template <bool NTL, bool NTS, int U>
__global__ void __launch_bounds__(256) k_chunk(const d2* x, const d2* y, const d2* z, d2* o, size_t n2, size_t chunk) {
  size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n2 ? lo + chunk : n2;
  for (size_t base = lo + threadIdx.x; base < hi; base += 256 * U) {
    d2 a[U], b[U], c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t i = base + 256 * u; if (i < hi) { a[u] = ld<NTL>(x + i); b[u] = ld<NTL>(y + i); c[u] = ld<NTL>(z + i); } }
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t i = base + 256 * u; if (i < hi) st<NTS>(o + i, a[u] + b[u] + c[u]); }
  }
}

static void time_it(const char* name, std::function<void()> fn, double bytes, int reps = 9) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  fn(); CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
  std::sort(ms.begin(), ms.end());
  printf("%-52s med %7.3f ms  min %7.3f ms  %6.0f GB/s\n", name, ms[ms.size() / 2], ms[0], bytes / ms[ms.size() / 2] / 1e6);
  fflush(stdout);
}

int main(int argc, char** argv) {
  size_t P = argc > 1 ? (size_t)atof(argv[1]) : 100000000;
  double *dx[3], *dout;
  for (int d = 0; d < 3; ++d) { CK(hipMalloc(&dx[d], P * 8)); CK(hipMemset(dx[d], 0, P * 8)); }
  CK(hipMalloc(&dout, P * 8));
  const d2 *x = (const d2*)dx[0], *y = (const d2*)dx[1], *z = (const d2*)dx[2];
  d2* o = (d2*)dout;
  const size_t n2 = P / 2;
  const double B4 = P * 32.0, B3 = P * 24.0, B1 = P * 8.0, B2 = P * 16.0;
#define GS(NTL, NTS, MODE, BLK, bytes, label) time_it(label, [&] { hipLaunchKernelGGL((k_gs<NTL, NTS, MODE>), dim3(BLK), dim3(256), 0, 0, x, y, z, o, n2); }, bytes)
  GS(false, false, 0, 2048, B4, "3R1W grid-stride plain            2048 blk");
  GS(true, false, 0, 2048, B4, "3R1W grid-stride nt-load          2048 blk");
  GS(false, true, 0, 2048, B4, "3R1W grid-stride nt-store         2048 blk");
  GS(true, true, 0, 2048, B4, "3R1W grid-stride nt both          2048 blk");
  GS(true, true, 0, 1024, B4, "3R1W grid-stride nt both          1024 blk");
  GS(true, true, 0, 4096, B4, "3R1W grid-stride nt both          4096 blk");
  GS(true, true, 0, 16384, B4, "3R1W grid-stride nt both         16384 blk");
  GS(false, false, 1, 2048, B3, "3R   grid-stride plain            2048 blk");
  GS(true, false, 1, 2048, B3, "3R   grid-stride nt               2048 blk");
  GS(false, false, 2, 2048, B1, "1W   grid-stride plain            2048 blk");
  GS(false, true, 2, 2048, B1, "1W   grid-stride nt               2048 blk");
  GS(false, false, 3, 2048, B2, "1R1W grid-stride plain            2048 blk");
  GS(true, true, 3, 2048, B2, "1R1W grid-stride nt               2048 blk");
#define CH(NTL, NTS, U, BLK, label) time_it(label, [&] { size_t chunk = ((n2 + BLK - 1) / BLK + 255) / 256 * 256; hipLaunchKernelGGL((k_chunk<NTL, NTS, U>), dim3(BLK), dim3(256), 0, 0, x, y, z, o, n2, chunk); }, B4)
  CH(false, false, 1, 2048, "3R1W block-chunk U1 plain         2048 blk");
  CH(true, true, 1, 2048, "3R1W block-chunk U1 nt            2048 blk");
  CH(true, true, 2, 2048, "3R1W block-chunk U2 nt            2048 blk");
  CH(true, true, 4, 2048, "3R1W block-chunk U4 nt            2048 blk");
  CH(true, true, 4, 8192, "3R1W block-chunk U4 nt            8192 blk");
  CH(true, true, 4, 65536, "3R1W block-chunk U4 nt           65536 blk");
  CH(false, false, 4, 65536, "3R1W block-chunk U4 plain        65536 blk");
  return 0;
}
