// Ablation harness for the headline kernel (3-D multilinear-regular, f64, fma flavour):
// instantiates the PRODUCT kernel source (interpn_amd/csrc/linear_brick.h) with its measurement
// flag ABL = 0 (unmodified), 1 (stream-only) and 2 (gather-only) on the brick layouts the library
// can choose.  Loaded only by bench.py (`roofline.ablation`): it times kernels, it is not on the
// product path and its ABL != 0 outputs are meaningless by construction.
//
//   void* ablate_create(const double* vals_dev, int n, int si, int sj, double step)
//   int   ablate_launch(void* h, int mode, x, y, z, out, npts, hipStream_t)   -> hipError_t
//   void  ablate_destroy(void* h)
#include <hip/hip_runtime.h>

#include <new>

#include "../interpn_amd/csrc/linear_brick.h"
#include "../interpn_amd/csrc/linear_sweep.h"

using namespace interpn;

namespace {

struct Ablate {
  double* bricks = nullptr;
  unsigned long long* first_bad = nullptr;
  SweepWork* work = nullptr;
  int n = 0, si = 0, sj = 0;
  unsigned nb[3] = {0, 0, 0};
  double step = 0;
};

unsigned along(int n, int step) {  // k_linear_brick.hip::bricks_along
  if (step == 1) return (unsigned)(n - 1);
  if (step == 2) return (unsigned)((n - 1) / 2 + 1);
  return (unsigned)((n - 2) / step + 1);
}

template <int SI, int SJ, int ABL>
hipError_t go(const Ablate& h, const BrickArgs<double, 3>& a, unsigned blocks, size_t lds, hipStream_t s) {
  hipLaunchKernelGGL((k_linear_brick<double, 3, false, true, SI, SJ, 2, 0, ABL>), dim3(blocks), dim3(kBlock), lds, s, a);
  return hipGetLastError();
}

template <int ABL>
hipError_t go_steps(const Ablate& h, const BrickArgs<double, 3>& a, unsigned blocks, size_t lds, hipStream_t s) {
  if (h.si == 1 && h.sj == 1) return go<1, 1, ABL>(h, a, blocks, lds, s);
  if (h.si == 1 && h.sj == 2) return go<1, 2, ABL>(h, a, blocks, lds, s);
  return go<2, 2, ABL>(h, a, blocks, lds, s);
}

}  // namespace

static size_t g_extra_lds = 0;  // occupancy probe: extra dynamic LDS per workgroup (bytes)

extern "C" {

void ablate_set_extra_lds(size_t bytes) { g_extra_lds = bytes; }

void* ablate_create(const double* vals_dev, int n, int si, int sj, double step) {
  if (!vals_dev || n < 2 || !((si == 1 || si == 2) && (sj == 1 || sj == 2)) || (si == 2 && sj == 1)) return nullptr;
  Ablate* h = new (std::nothrow) Ablate();
  if (!h) return nullptr;
  h->n = n; h->si = si; h->sj = sj; h->step = step;
  h->nb[0] = along(n, si); h->nb[1] = along(n, sj); h->nb[2] = along(n, 3);
  const size_t elems = (size_t)h->nb[0] * h->nb[1] * h->nb[2] * 16;
  if (elems >= 0xFFFFFFFFull || hipMalloc((void**)&h->bricks, elems * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&h->first_bad, 8) != hipSuccess || hipMemset(h->first_bad, 0xFF, 8) != hipSuccess ||
      hipMalloc((void**)&h->work, sizeof(SweepWork)) != hipSuccess || hipMemset(h->work, 0, sizeof(SweepWork)) != hipSuccess) {
    (void)hipFree(h->bricks); (void)hipFree(h->first_bad);
    delete h;
    return nullptr;
  }
  size_t blocks = (elems + kBlock - 1) / kBlock;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(k_build_bricks<double>, dim3((unsigned)blocks), dim3(kBlock), 0, nullptr, vals_dev, h->bricks, (size_t)1,
                     n, n, n, si, sj, h->nb[0], h->nb[1], h->nb[2]);
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(h->bricks); (void)hipFree(h->first_bad);
    delete h;
    return nullptr;
  }
  return h;
}

int ablate_launch(void* handle, int mode, const double* x, const double* y, const double* z, double* out, size_t npts,
                  void* stream) {
  const Ablate* h = static_cast<const Ablate*>(handle);
  if (!h || !x || !y || !z || !out || npts == 0 || (npts & 1) || mode < 0 || mode > 4) return (int)hipErrorInvalidValue;
  for (const void* p : {(const void*)x, (const void*)y, (const void*)z, (const void*)out})
    if (reinterpret_cast<uintptr_t>(p) % 16) return (int)hipErrorInvalidValue;  // the two-points-per-lane form
  BrickArgs<double, 3> a;
  a.gate = nullptr;
  a.bricks = h->bricks;
  a.obs[0] = x; a.obs[1] = y; a.obs[2] = z;
  a.out = out;
  a.first_bad = h->first_bad;
  a.npts = npts;
  for (int d = 0; d < 3; ++d) { a.start[d] = -1.0; a.step[d] = h->step; a.n[d] = h->n; }
  a.nbj = h->nb[1];
  a.nbk = h->nb[2];
  a.lead_stride[0] = 0;
  a.ax.use_lds = 0; a.ax.image = nullptr; a.ax.image_bytes = 0;
  a.iters = 1;  // the library's launch shape for regular grids: one 256-lane row per workgroup
  typedef LeafVec<double, 2>::type P;
  const size_t lds = (size_t)kBlock * kPieceRow * sizeof(P) + (size_t)kBlock * 16 + g_extra_lds;
  const size_t nslots = (npts + 1) / 2;
  const unsigned blocks = (unsigned)((nslots + kBlock - 1) / kBlock);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == 0) return (int)go_steps<0>(*h, a, blocks, lds, s);
  if (mode == 1) return (int)go_steps<1>(*h, a, blocks, lds, s);
  if (mode == 2) return (int)go_steps<2>(*h, a, blocks, lds, s);
  if (mode == 3) return (int)go_steps<3>(*h, a, blocks, lds, s);
  if (npts % 512) return (int)hipErrorInvalidValue;  // the LDS-DMA variant handles whole rows only
  return (int)go_steps<4>(*h, a, blocks, lds + 4 * 3 * 1024, s);
}

// The sweep evaluation (linear_sweep.h) on the same table: K rows of 64 points per wave and round,
// `threads` per workgroup, `wgs` persistent workgroups per CU (0: 1024 / threads).  Returns hipError_t; -1: shape not instantiated.
static unsigned long long* g_sweep_stamps = nullptr;  // device buffer, 8 words per workgroup (STAMPS build of the kernel)
void ablate_set_sweep_stamps(unsigned long long* p) { g_sweep_stamps = p; }
static int g_sweep_parked = 0;  // rows per wave and round parked in LDS beside the K in registers (linear_sweep.h KL): 0 or 2
void ablate_set_sweep_parked(int kl) { g_sweep_parked = kl; }
static int g_sweep_abl = 0;  // linear_sweep.h ABL: 1 no table access, 2 no streams (K = 12 x 768 + 2 parked rows only)
void ablate_set_sweep_abl(int abl) { g_sweep_abl = abl; }
static unsigned g_sweep_fastdiv = 1;  // 0: the divide sequences for every row (the form before step_cell_fast)
void ablate_set_sweep_fastdiv(unsigned on) { g_sweep_fastdiv = on; }
static unsigned g_sweep_clock = 0;  // ticks of 10 ns per sweep of the leading index (0: measured by the previous launch, 1: rows in sorted order)
void ablate_set_sweep_clock(unsigned ticks) { g_sweep_clock = ticks; }

int ablate_launch_sweep(void* handle, const double* x, const double* y, const double* z, double* out, size_t npts, int K,
                        int threads, int wgs, void* stream) {
  const Ablate* h = static_cast<const Ablate*>(handle);
  if (!h || !x || !y || !z || !out || npts == 0) return (int)hipErrorInvalidValue;
  for (const void* p : {(const void*)x, (const void*)y, (const void*)z, (const void*)out})
    if (reinterpret_cast<uintptr_t>(p) % 16) return (int)hipErrorInvalidValue;
  SweepArgs<double> s;
  BrickArgs<double, 3>& a = s.b;
  a.gate = nullptr;
  s.gated = 0;
  a.bricks = h->bricks;
  a.obs[0] = x; a.obs[1] = y; a.obs[2] = z;
  a.out = out;
  a.first_bad = h->first_bad;
  a.npts = npts;
  for (int d = 0; d < 3; ++d) { a.start[d] = -1.0; a.step[d] = h->step; a.n[d] = h->n; }
  a.nbj = h->nb[1];
  a.nbk = h->nb[2];
  a.lead_stride[0] = 0;
  a.ax.use_lds = 0; a.ax.image = nullptr; a.ax.image_bytes = 0;
  a.iters = 1;
  s.key_start = -1.0;
  s.key_scale = 1.0 / h->step;
  s.key_shift = 0;
  while (((h->n - 2) >> s.key_shift) >= 64) ++s.key_shift;
  { const volatile double one = 1.0; for (int d = 0; d < 3; ++d) s.rstep[d] = one / h->step; }
  s.fastdiv = g_sweep_fastdiv;
  const size_t chunk = (size_t)64 * (K + g_sweep_parked);
  s.rounds = (unsigned)((npts + chunk - 1) / chunk);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  s.per_shard = (s.rounds + 7u) / 8u;
  s.period = g_sweep_clock;  // 0: what the previous launch measured; 1: no clock
  s.period_default = 2000;
  s.work = h->work;
  s.stamps = g_sweep_stamps;
  hipStream_t st = static_cast<hipStream_t>(stream);
#define SWEEP(SI_, SJ_, K_, TH_) SWEEP_KL(SI_, SJ_, K_, TH_, 0)
#define SWEEP_KL(SI_, SJ_, K_, TH_, KL_) SWEEP_ABL(SI_, SJ_, K_, TH_, KL_, 0)
#define SWEEP_ABL(SI_, SJ_, K_, TH_, KL_, ABL_)                                                                           \
  if (h->si == SI_ && h->sj == SJ_ && K == K_ && threads == TH_ && g_sweep_parked == KL_ && g_sweep_abl == ABL_) {         \
    auto kern = g_sweep_stamps ? k_linear_sweep<double, false, true, SI_, SJ_, K_, TH_, 0, true, 0, KL_, ABL_> : k_linear_sweep<double, false, true, SI_, SJ_, K_, TH_, 0, false, 0, KL_, ABL_>; \
    const size_t lds = (size_t)SweepLds<double, K_, KL_>::kWave * (TH_ / 64) + SweepLds<double, K_, KL_>::kWorkgroup;                                                   \
    if (lds > 64 * 1024) {                                                                                                \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      if (e != hipSuccess) return (int)e;                                                                                 \
    }                                                                                                                     \
    unsigned blocks = (unsigned)cus * (unsigned)(wgs > 0 ? wgs : (TH_ >= 1024 ? 1 : 1024 / TH_));                         \
    if (blocks > (s.rounds + (TH_ / 64) - 1) / (TH_ / 64)) blocks = (s.rounds + (TH_ / 64) - 1) / (TH_ / 64);             \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(TH_), lds, st, s);                                                        \
    return (int)hipGetLastError();                                                                                        \
  }
#define SWEEP_LAYOUTS(K_, TH_) SWEEP(1, 1, K_, TH_) SWEEP(1, 2, K_, TH_)
  SWEEP_LAYOUTS(8, 1024)
  SWEEP_KL(1, 1, 12, 768, 2)
  SWEEP_KL(1, 1, 12, 768, 4)
  SWEEP_ABL(1, 1, 12, 768, 4, 1)
  SWEEP_ABL(1, 1, 12, 768, 4, 2)
  SWEEP_ABL(1, 1, 12, 768, 2, 1)
  SWEEP_ABL(1, 1, 12, 768, 2, 2)
  SWEEP_ABL(1, 1, 12, 768, 2, 3)
  SWEEP_KL(1, 1, 14, 768, 2)
  SWEEP_KL(1, 2, 12, 768, 2)
  SWEEP_KL(1, 1, 8, 1024, 2)
  SWEEP_LAYOUTS(8, 768)
  SWEEP_LAYOUTS(12, 768)
  SWEEP_LAYOUTS(14, 768)
  SWEEP_LAYOUTS(16, 768)
  SWEEP_LAYOUTS(20, 512)
  SWEEP_LAYOUTS(24, 512)
  SWEEP_LAYOUTS(8, 512)
  SWEEP_LAYOUTS(16, 512)
#undef SWEEP_LAYOUTS
#undef SWEEP
#undef SWEEP_KL
#undef SWEEP_ABL
  return -1;
}

// Nothing but visits of random 128-byte lines, the way the product kernels make them (the four lanes of a quad read
// 4 x 16 bytes spread over one line, 16 lines per wave instruction; tools/tune_sector.hip, mode B): what `visits` of them
// cost from a table of `table_bytes` — the floor under any kernel that reads one line per point.  `table` >= table_bytes.
namespace {
__device__ __forceinline__ unsigned long long visit_mix(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) k_line_visits(const double2* __restrict__ tab, size_t nlines, double* sink, size_t nquads, int reps) {
  const size_t nthreads = (size_t)gridDim.x * 256;
  double acc = 0;
  for (size_t s = (size_t)blockIdx.x * 256 + threadIdx.x; s < nquads * 4; s += nthreads) {
    const size_t quad = s >> 2;
    const unsigned q = (unsigned)(s & 3);
    for (int r = 0; r < reps; ++r) {
      const size_t line = visit_mix(quad * 16 + (unsigned)r) % nlines;
      const double2 v = tab[line * 8 + 2 * q];
      acc += v.x + v.y;
    }
  }
  if (acc == 1.2345) sink[0] = acc;  // (never: the table holds finite table values)
}
}  // namespace
int ablate_line_visits(const void* table, size_t table_bytes, size_t visits, double* sink, void* stream) {
  if (!table || !sink || table_bytes < 128 || visits < 4) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(k_line_visits, dim3(2048), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const double2*>(table), table_bytes / 128,
                     sink, visits / 4, 4);
  return (int)hipGetLastError();
}

// The division-free forms of interpn_device.h against the hardware's IEEE divide sequences, on the GPU itself: every
// thread draws (a, b) pairs over the admitted exponent ranges (random significands, both signs; every fourth pair with a
// quotient next to an integer or a power of two) and compares divide_fast(a, b, RN(1 / b)) with a / b bit for bit, and
// floor_quotient_fast's floor with floor(a / b) wherever it reports its result as exact.  Returns the number of
// differing pairs in `*mismatches` (tests/test_gpu_parity.py::test_division_free_forms_against_the_divide_sequences).
extern "C++" {
namespace {
template <typename T> struct SelfTestBits;
template <> struct SelfTestBits<double> {
  static __device__ double make(unsigned long long r, int elo, int ehi) {
    const int e = elo + (int)((r >> 53) % (unsigned)(ehi - elo));
    const double m = 1.0 + (double)(r & ((1ull << 52) - 1)) * 0x1p-52;
    return ldexp((r >> 63) ? -m : m, e);
  }
  static __device__ bool same(double x, double y) { return __double_as_longlong(x) == __double_as_longlong(y); }
  static constexpr int alo = -256, ahi = 256, blo = -128, bhi = 128;
};
template <> struct SelfTestBits<float> {
  static __device__ float make(unsigned long long r, int elo, int ehi) {
    const int e = elo + (int)((r >> 40) % (unsigned)(ehi - elo));
    const float m = 1.0f + (float)(r & ((1u << 23) - 1)) * 0x1p-23f;
    return ldexpf((r >> 63) ? -m : m, e);
  }
  static __device__ bool same(float x, float y) { return __float_as_uint(x) == __float_as_uint(y); }
  static constexpr int alo = -24, ahi = 24, blo = -16, bhi = 16;
};
template <typename T>
__global__ void __launch_bounds__(256) k_division_selftest(unsigned long long seed, unsigned per_thread, unsigned long long* mismatches, unsigned long long* checked_floor) {
  typedef SelfTestBits<T> B;
  unsigned long long z = seed + ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  unsigned long long bad = 0, fl = 0;
  for (unsigned i = 0; i < per_thread; ++i) {
    const unsigned long long r1 = visit_mix(z), r2 = visit_mix(z + 1), r3 = visit_mix(z + 2);
    z += 3;
    const T b = B::make(r1, B::blo, B::bhi);
    T a = B::make(r2, B::alo, B::ahi);
    if ((i & 3u) == 3u) {  // quotients next to integers / powers of two: a = b * (k + tiny) in working precision
      const T k = (T)(long long)((r3 >> 8) % 4096ull) + (T)1;
      const T eps = (T)((r3 & 255ull)) * (sizeof(T) == 8 ? (T)0x1p-50 : (T)0x1p-21);
      a = b * (k + ((r3 >> 62) & 1 ? eps : -eps));
    }
    const volatile T one = (T)1;
    const T rb = one / b;
    T q;
    const bool in_range = divide_fast(a, b, rb, &q);
    if (in_range && !B::same(q, a / b)) ++bad;
    T floc;
    if (floor_quotient_fast(a, rb, &floc)) {
      ++fl;
      const T ref = sizeof(T) == 8 ? (T)floor((double)(a / b)) : (T)floorf((float)(a / b));
      if (!B::same(floc, ref) && !(floc == ref)) ++bad;  // (floors of +0 / -0 compare equal)
    }
  }
  if (bad) atomicAdd(mismatches, bad);
  atomicAdd(checked_floor, fl);
}
}  // namespace
}  // extern "C++"
int ablate_division_selftest(int f32, unsigned long long seed, unsigned per_thread, unsigned long long* mismatches, unsigned long long* floors_checked) {
  unsigned long long* d = nullptr;
  if (hipMalloc(&d, 16) != hipSuccess) return (int)hipGetLastError();
  (void)hipMemset(d, 0, 16);
  if (f32) hipLaunchKernelGGL(k_division_selftest<float>, dim3(4096), dim3(256), 0, 0, seed, per_thread, d, d + 1);
  else hipLaunchKernelGGL(k_division_selftest<double>, dim3(4096), dim3(256), 0, 0, seed, per_thread, d, d + 1);
  hipError_t e = hipDeviceSynchronize();
  unsigned long long h[2] = {~0ull, 0};
  if (e == hipSuccess) e = hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (mismatches) *mismatches = h[0];
  if (floors_checked) *floors_checked = h[1];
  return (int)e;
}

void ablate_destroy(void* handle) {
  Ablate* h = static_cast<Ablate*>(handle);
  if (!h) return;
  (void)hipDeviceSynchronize();
  (void)hipFree(h->bricks);
  (void)hipFree(h->first_bad);
  (void)hipFree(h->work);
  delete h;
}

}  // extern "C"
