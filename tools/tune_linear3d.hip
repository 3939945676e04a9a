// Tuning harness (not part of the library): ablations and launch-shape sweeps of the 3-D
// multilinear-regular kernel on one MI355X.  Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950
//   -I interpn_amd/csrc tools/tune_linear3d.hip -o tools/tune_linear3d
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#include "interpn_kernels.h"

using namespace interpn;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// MODE bits: 1 = no gather (fake values), 2 = reciprocal multiply instead of divide,
//            4 = no obs load (synthesise x from index), 8 = no store
template <int U, int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_var(const RegularArgs<double, 3> a, double rinv0, double rinv1, double rinv2) {
  typedef double T;
  constexpr int N = 3;
  const double rinv[3] = {rinv0, rinv1, rinv2};
  const size_t nthreads = (size_t)gridDim.x * BLOCK;
  for (size_t i0 = (size_t)blockIdx.x * BLOCK + threadIdx.x; i0 < a.npts; i0 += nthreads * U) {
    T x[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * nthreads;
#pragma unroll
      for (int d = 0; d < N; ++d) {
        if (MODE & 4) {
          unsigned long long z = (i * 3 + d) * 0x9E3779B97F4A7C15ull;
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
          x[u][d] = -1.0 + 2.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0);
        }
        else x[u][d] = (i < a.npts) ? a.obs[d][i] : a.start[d];
      }
    }
    T t[U][N];
    unsigned base[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      base[u] = 0;
#pragma unroll
      for (int d = 0; d < N; ++d) {
        T floc;
        if (MODE & 2) floc = __builtin_floor((x[u][d] - a.start[d]) * rinv[d]);
        else floc = __builtin_floor((x[u][d] - a.start[d]) / a.step[d]);
        const int loc = clamp_loc<T>(floc, a.n[d] - 2);
        const T izl = __builtin_fma(a.step[d], (T)loc, a.start[d]);
        if (MODE & 2) t[u][d] = (x[u][d] - izl) * rinv[d];
        else t[u][d] = (x[u][d] - izl) / a.step[d];
        base[u] += (unsigned)loc * a.stride[d];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = i0 + (size_t)u * nthreads;
      T res;
      if (MODE & 1) {
        res = t[u][0] + t[u][1] * t[u][2] + (double)base[u];
      } else {
        Leaf<T, 2> r = LinearTree<T, unsigned, N - 1, true>::run(a.vals, base[u], a.stride, t[u]);
        const T y0 = r.v[0];
        const T dy = r.v[1] - y0;
        res = __builtin_fma(t[u][N - 1], dy, y0);
      }
      if (MODE & 8) { if (res == 123.456) a.out[i] = res; }
      else if (i < a.npts) a.out[i] = res;
    }
  }
}

// plain streaming reference: out = x + y + z
__global__ void __launch_bounds__(256) k_stream(const double* x, const double* y, const double* z, double* o, size_t n) {
  const size_t nthreads = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += nthreads) o[i] = x[i] + y[i] + z[i];
}

// the same stream with the product kernels' access pattern: 16 B per lane, non-temporal
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_stream16nt(const d2* x, const d2* y, const d2* z, d2* o, size_t n2) {
  const size_t nthreads = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += nthreads) {
    d2 a = __builtin_nontemporal_load(x + i), b = __builtin_nontemporal_load(y + i), c = __builtin_nontemporal_load(z + i);
    __builtin_nontemporal_store(a + b + c, o + i);
  }
}

static double time_it(const char* name, std::function<void()> fn, size_t P, int reps = 7) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  fn(); CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
  std::sort(ms.begin(), ms.end());
  double med = ms[ms.size() / 2];
  printf("%-44s  med %7.3f ms  min %7.3f ms  %7.1f Gpts/s  %6.0f GB/s (32B/pt)\n", name, med, ms[0], P / med / 1e6, P * 32.0 / med / 1e6);
  fflush(stdout);
  return med;
}

int main(int argc, char** argv) {
  size_t P = argc > 1 ? (size_t)atof(argv[1]) : 100000000;
  int n = argc > 2 ? atoi(argv[2]) : 64;
  size_t G = (size_t)n * n * n;
  std::vector<double> hv(G), hx(P);
  srand(1);
  for (auto& v : hv) v = rand() / (double)RAND_MAX * 2 - 1;
  double *dv, *dx[3], *dout;
  CK(hipMalloc(&dv, G * 8)); CK(hipMemcpy(dv, hv.data(), G * 8, hipMemcpyHostToDevice));
  unsigned long long* fb; CK(hipMalloc(&fb, 8)); CK(hipMemset(fb, 0xFF, 8));
  for (int d = 0; d < 3; ++d) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (d + 1);
    for (size_t i = 0; i < P; ++i) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; hx[i] = -1.0 + 2.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0); }
    CK(hipMalloc(&dx[d], P * 8)); CK(hipMemcpy(dx[d], hx.data(), P * 8, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&dout, P * 8));
  RegularArgs<double, 3> a;
  a.vals = dv; a.out = dout; a.first_bad = fb; a.npts = P; a.linearize = 0;
  double step = 2.0 / (n - 1);
  unsigned acc = 1;
  for (int d = 2; d >= 0; --d) { a.obs[d] = dx[d]; a.start[d] = -1.0; a.step[d] = step; a.n[d] = n; a.stride[d] = acc; acc *= n; }
  double ri = 1.0 / step;
  printf("P=%zu grid=%d^3\n", P, n);

  time_it("stream x+y+z (2048 blk)", [&] { hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, 0, dx[0], dx[1], dx[2], dout, P); }, P);
  time_it("stream x+y+z (8192 blk)", [&] { hipLaunchKernelGGL(k_stream, dim3(8192), dim3(256), 0, 0, dx[0], dx[1], dx[2], dout, P); }, P);

  time_it("stream16nt x+y+z (2048 blk)", [&] { hipLaunchKernelGGL(k_stream16nt, dim3(2048), dim3(256), 0, 0, (const d2*)dx[0], (const d2*)dx[1], (const d2*)dx[2], (d2*)dout, P / 2); }, P);
  if (argc > 3) return 0;  // calibration only

#define RUN(U, MODE, BLOCK, BLOCKS, label) time_it(label, [&] { hipLaunchKernelGGL((k_var<U, MODE, BLOCK>), dim3(BLOCKS), dim3(BLOCK), 0, 0, a, ri, ri, ri); }, P)
  RUN(2, 0, 256, 2048, "U2 full            2048x256");
  RUN(1, 0, 256, 2048, "U1 full            2048x256");
  RUN(4, 0, 256, 2048, "U4 full            2048x256");
  RUN(1, 0, 256, 4096, "U1 full            4096x256");
  RUN(2, 0, 256, 4096, "U2 full            4096x256");
  RUN(2, 0, 256, 1024, "U2 full            1024x256");
  RUN(1, 0, 256, 8192, "U1 full            8192x256");
  RUN(1, 0, 256, (unsigned)((P + 255) / 256), "U1 full            one-shot grid");
  RUN(2, 0, 512, 1024, "U2 full            1024x512");
  RUN(2, 0, 1024, 512, "U2 full             512x1024");
  RUN(2, 0, 64, 8192, "U2 full            8192x64");
  RUN(2, 1, 256, 2048, "U2 no-gather       2048x256");
  RUN(2, 2, 256, 2048, "U2 rcp-mul         2048x256");
  RUN(2, 3, 256, 2048, "U2 no-gather+rcp   2048x256");
  RUN(2, 4, 256, 2048, "U2 no-obs-load     2048x256");
  RUN(2, 8, 256, 2048, "U2 no-store        2048x256");
  RUN(2, 12, 256, 2048, "U2 no-obs,no-store 2048x256");
  RUN(2, 14, 256, 2048, "U2 gather only(rcp) 2048x256");
  RUN(4, 14, 256, 2048, "U4 gather only(rcp) 2048x256");
  RUN(1, 14, 256, 2048, "U1 gather only(rcp) 2048x256");
  return 0;
}
