// FP64 vector (VALU, not MFMA) issue rates on gfx950: v_fma_f64 / v_add_f64 / v_mul_f64 per SIMD.
// The microarchitecture guide gives no FP64 vector peak; bench.py's cfg4 rows quote the one
// measured here.    hipcc --offload-arch=gfx950 -O3 -o tools/fp64_rate tools/fp64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int OP>
__global__ void __launch_bounds__(256) k_rate(double* out, double seed, int iters) {
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x * 1e-9;
  const double m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) a[i] = __builtin_fma(a[i], m, c);
        else if (OP == 1) a[i] = a[i] + c;
        else a[i] = a[i] * m;
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678) out[0] = s;
}

template <int OP>
double run(int waves_per_simd, int iters, double* d) {
  int dev = 0;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, dev);
  const int cus = p.multiProcessorCount;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 * iters * 64.0;  // wave-instructions
  const double per_simd_per_us = insts / (cus * 4) / (ms * 1e3);
  printf("op %s  waves/SIMD %d  %.3f ms  %.1f wave-instr/us/SIMD  -> %.2f TFLOP/s-equivalent (x%d flop)  CUs %d\n",
         OP == 0 ? "fma" : (OP == 1 ? "add" : "mul"), waves_per_simd, ms, per_simd_per_us,
         insts * 64 * (OP == 0 ? 2 : 1) / (ms * 1e-3) / 1e12, OP == 0 ? 2 : 1, cus);
  return ms;
}

int main() {
  double* d;
  hipMalloc(&d, 64);
  for (int w : {1, 2, 4, 8}) {
    run<0>(w, 20000, d);
    run<1>(w, 20000, d);
    run<2>(w, 20000, d);
  }
  return 0;
}
