// Does the L2 fetch half lines?  Random quad-cooperative gathers from a table far larger than L2:
//   A: the 4 lanes of a quad read the first 64 B of a random 128-B line
//   B: the 4 lanes read 16 B at offsets 0/32/64/96 of a random 128-B line (both halves touched)
//   C: the 4 lanes read a random aligned 64-B unit (either half)
//   D: the 4 lanes read the whole... 2 x 64 B: lanes read 32 B each (full 128-B line)
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/tune_sector tools/tune_sector.hip ; run: tools/tune_sector [table MiB]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const double2* __restrict__ tab, size_t nlines, double* out, size_t nquads, int reps) {
  const size_t nthreads = (size_t)gridDim.x * 256;
  double acc = 0;
  for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < nquads * 4; s += nthreads) {
    const size_t quad = s >> 2; const unsigned q = s & 3;
    for (int r = 0; r < reps; ++r) {
      const unsigned long long h = mix(quad * 16 + r);
      const size_t line = h % nlines;
      size_t off;  // in 16-B units
      if (MODE == 0) off = line * 8 + q;
      else if (MODE == 1) off = line * 8 + 2 * q;
      else if (MODE == 2) off = line * 8 + ((h >> 40) & 1) * 4 + q;
      else off = line * 8 + 2 * q;
      double2 v = tab[off];
      acc += v.x + v.y;
      if (MODE == 3) { double2 w = tab[off + 1]; acc += w.x + w.y; }
    }
  }
  if (acc == 1.2345) out[0] = acc;
}

static double time_it(const char* name, std::function<void()> fn, double lines) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  fn(); CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int r = 0; r < 5; ++r) { CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
  std::sort(ms.begin(), ms.end());
  printf("%-44s med %7.3f ms  %6.1f Glines/s\n", name, ms[2], lines / ms[2] / 1e6); fflush(stdout);
  return ms[2];
}

int main(int argc, char** argv) {
  size_t mib = argc > 1 ? atol(argv[1]) : 128;
  size_t bytes = mib << 20, nlines = bytes / 128;
  double2* tab; double* out;
  CK(hipMalloc(&tab, bytes)); CK(hipMemset(tab, 0, bytes)); CK(hipMalloc(&out, 8));
  const size_t nquads = 25000000; const int reps = 4;  // 1e8 line visits
  printf("table %zu MiB\n", mib);
  time_it("A first half of a line (64 B)", [&] { hipLaunchKernelGGL(k_gather<0>, dim3(2048), dim3(256), 0, 0, tab, nlines, out, nquads, reps); }, 1e8);
  time_it("B 4 x 16 B spread over both halves", [&] { hipLaunchKernelGGL(k_gather<1>, dim3(2048), dim3(256), 0, 0, tab, nlines, out, nquads, reps); }, 1e8);
  time_it("C random half (64 B)", [&] { hipLaunchKernelGGL(k_gather<2>, dim3(2048), dim3(256), 0, 0, tab, nlines, out, nquads, reps); }, 1e8);
  time_it("D whole line (4 x 32 B)", [&] { hipLaunchKernelGGL(k_gather<3>, dim3(2048), dim3(256), 0, 0, tab, nlines, out, nquads, reps); }, 1e8);
  return 0;
}
