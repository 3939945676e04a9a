// Tuning harness 2 (not part of the library): grid memory layouts and load cache policies for the
// 3-D multilinear-regular gather.  Bricks are 2(i) x 2(j) x 4(k) f64 = one 128-B L2/L1 line;
// SI/SJ/SK are the brick steps (step < extent => overlapping bricks = duplicated storage).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

struct Args {
  const double* vals; const double* obs[3]; double* out; size_t npts;
  double start, step, rinv; int n;
  unsigned nbj, nbk;  // bricks along j, k
  size_t in_mask, out_mask;  // index masks (all ones = normal; small = keep obs/out cache-resident)
};

enum { LD_PLAIN = 0, LD_NT = 1 };

template <int LD> __device__ __forceinline__ double ld1(const double* p) {
  if (LD == LD_NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <int LD> __device__ __forceinline__ d2u ld2(const double* p) {
  if (LD == LD_NT) return __builtin_nontemporal_load((const d2u*)p);
  return *(const d2u*)p;
}

// corner address in a bricked table
template <int SI, int SJ, int SK>
__device__ __forceinline__ unsigned brick_addr(const Args& a, int i, int j, int k, int di, int dj, int dk) {
  int bi = i / SI, oi = i - bi * SI + di; if (oi >= 2) { bi += 1; oi -= SI; }
  int bj = j / SJ, oj = j - bj * SJ + dj; if (oj >= 2) { bj += 1; oj -= SJ; }
  int bk = k / SK, ok = k - bk * SK + dk; if (ok >= 4) { bk += 1; ok -= SK; }
  return ((unsigned)(bi * a.nbj + bj) * a.nbk + bk) * 16u + (unsigned)((oi * 2 + oj) * 4 + ok);
}

// 4(j) x 4(k) tiles of one plane i, stepped 3 along j and k (a (j,j+1) x (k,k+1) patch never leaves a
// tile): 1.78x the grid, always exactly 2 lines per cell (planes i and i+1).  SI = 0 selects it.
template <>
__device__ __forceinline__ unsigned brick_addr<0, 0, 0>(const Args& a, int i, int j, int k, int di, int dj, int dk) {
  const int tj = j / 3, oj = j - tj * 3 + dj;
  const int tk = k / 3, ok = k - tk * 3 + dk;
  return ((unsigned)((i + di) * a.nbj + tj) * a.nbk + tk) * 16u + (unsigned)(oj * 4 + ok);
}

// LAYOUT 0 = row-major; 1 = bricked with steps SI,SJ,SK.  NOSTREAM: synthesise obs, skip store.
template <int LAYOUT, int SI, int SJ, int SK, int LD, bool NOSTREAM>
__global__ void __launch_bounds__(256) k_lay(const Args a) {
  const size_t nthreads = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < a.npts; i0 += nthreads) {
    double x[3], t[3]; int loc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (NOSTREAM) {
        unsigned long long z = (i0 * 3 + d) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        x[d] = -1.0 + 2.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0);
      } else x[d] = a.obs[d][i0 & a.in_mask];
      double floc = __builtin_floor((x[d] - a.start) / a.step);
      floc = floc > 0 ? floc : 0; int l = (int)floc; l = l < a.n - 2 ? l : a.n - 2; loc[d] = l;
      t[d] = (x[d] - __builtin_fma(a.step, (double)l, a.start)) / a.step;
    }
    double v[2][2][2];
    if (LAYOUT == 0) {
      unsigned base = ((unsigned)loc[0] * a.n + loc[1]) * a.n + loc[2];
#pragma unroll
      for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) { d2u p = ld2<LD>(a.vals + base + (di * a.n + dj) * a.n); v[di][dj][0] = p.x; v[di][dj][1] = p.y; }
    } else {
      // pair loads along k when both k corners are in the same brick row (always true when SK == 3)
#pragma unroll
      for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
          unsigned a0 = brick_addr<SI, SJ, SK>(a, loc[0], loc[1], loc[2], di, dj, 0);
          if (SK == 3) { d2u p = ld2<LD>(a.vals + a0); v[di][dj][0] = p.x; v[di][dj][1] = p.y; }
          else {
            unsigned a1 = brick_addr<SI, SJ, SK>(a, loc[0], loc[1], loc[2], di, dj, 1);
            v[di][dj][0] = ld1<LD>(a.vals + a0); v[di][dj][1] = ld1<LD>(a.vals + a1);
          }
        }
    }
    // reference order: dim 0 first, dim 2 last
    double r[2];
#pragma unroll
    for (int dk = 0; dk < 2; ++dk) {
      double c0 = __builtin_fma(t[0], v[1][0][dk] - v[0][0][dk], v[0][0][dk]);
      double c1 = __builtin_fma(t[0], v[1][1][dk] - v[0][1][dk], v[0][1][dk]);
      r[dk] = __builtin_fma(t[1], c1 - c0, c0);
    }
    double res = __builtin_fma(t[2], r[1] - r[0], r[0]);
    if (NOSTREAM) { if (res == 123.456) a.out[i0] = res; } else a.out[i0] = res;
  }
}


// Cooperative variant: the 4 lanes of a quad fetch the 4 16-B pieces of ONE point per load
// instruction (pieces that share a 128-B line are merged by the TCP into one L2 request), then the
// pieces are transposed back through LDS so that every lane finishes its own point.
template <int SI, int SJ, int SK, bool NOSTREAM, bool PF = false, int SP = 0>
__global__ void __launch_bounds__(256) k_coop(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[256 * 80 + 256 * 16];
  const unsigned lane = threadIdx.x;            // 0..255
  const unsigned q = lane & 3;                  // piece index / position in quad
  const unsigned quad = lane >> 2;              // 0..63
  d2u* lds_piece = reinterpret_cast<d2u*>(lds_raw);                       // [quad][r][q] padded: quad stride 80*4 B
  unsigned* lds_off = reinterpret_cast<unsigned*>(lds_raw + 256 * 80);    // [quad][q][r]
  const size_t nthreads = (size_t)gridDim.x * 256;
  const size_t niter = (a.npts + nthreads - 1) / nthreads;
  double xn[3] = {a.start, a.start, a.start};
  if (PF) {
    const size_t i1 = (size_t)blockIdx.x * 256 + lane;
#pragma unroll
    for (int d = 0; d < 3; ++d) if (i1 < a.npts) xn[d] = a.obs[d][i1];
  }
  for (size_t it = 0; it < niter; ++it) {
    const size_t i0 = it * nthreads + (size_t)blockIdx.x * 256 + lane;
    const bool live = i0 < a.npts;
    double x[3], t[3]; int loc[3];
    if (PF) {
#pragma unroll
      for (int d = 0; d < 3; ++d) x[d] = xn[d];
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (PF) {
      } else if (NOSTREAM) {
        unsigned long long z = (i0 * 3 + d) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        x[d] = -1.0 + 2.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0);
      } else if (SP >= 4) {
        x[d] = a.start;
        if (live) {
          const double* ptr = &a.obs[d][i0];
          if (SP == 5) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x[d]) : "v"(ptr) : "memory");
          else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x[d]) : "v"(ptr) : "memory");
        }
      } else x[d] = live ? ((SP == 1 || SP == 2) ? __builtin_nontemporal_load(&a.obs[d][i0]) : a.obs[d][i0]) : a.start;
      double floc = __builtin_floor((x[d] - a.start) / a.step);
      floc = floc > 0 ? floc : 0; int l = (int)floc; l = l < a.n - 2 ? l : a.n - 2; loc[d] = l;
      t[d] = (x[d] - __builtin_fma(a.step, (double)l, a.start)) / a.step;
    }
    // my point's 4 piece offsets -> LDS transposed so that lane q can read piece q of points r=0..3
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      unsigned o = brick_addr<SI, SJ, SK>(a, loc[0], loc[1], loc[2], p >> 1, p & 1, 0);
      lds_off[(quad * 4 + p) * 4 + q] = o;     // row p (piece), column q (= point index r of this lane)
    }
    __builtin_amdgcn_wave_barrier();
    uint4 toff = *reinterpret_cast<uint4*>(&lds_off[(quad * 4 + q) * 4]);  // piece q of points 0..3
    d2u pc[4];
    pc[0] = *(const d2u*)(a.vals + toff.x);
    pc[1] = *(const d2u*)(a.vals + toff.y);
    pc[2] = *(const d2u*)(a.vals + toff.z);
    pc[3] = *(const d2u*)(a.vals + toff.w);
    if (PF) {  // next iteration's coordinates, issued behind the gathers (VMEM returns in order)
      const size_t i1 = i0 + nthreads;
#pragma unroll
      for (int d = 0; d < 3; ++d) xn[d] = (i1 < a.npts) ? a.obs[d][i1] : a.start;
    }
    // write piece q of point r at [quad][r][q]; quad stride 5*4 d2u (80 B per point row -> padding)
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_piece[(quad * 4 + r) * 5 + q] = pc[r];
    __builtin_amdgcn_wave_barrier();
    double v[2][2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) { d2u w = lds_piece[(quad * 4 + q) * 5 + p]; v[p >> 1][p & 1][0] = w.x; v[p >> 1][p & 1][1] = w.y; }
    __builtin_amdgcn_wave_barrier();
    double r[2];
#pragma unroll
    for (int dk = 0; dk < 2; ++dk) {
      double c0 = __builtin_fma(t[0], v[1][0][dk] - v[0][0][dk], v[0][0][dk]);
      double c1 = __builtin_fma(t[0], v[1][1][dk] - v[0][1][dk], v[0][1][dk]);
      r[dk] = __builtin_fma(t[1], c1 - c0, c0);
    }
    double res = __builtin_fma(t[2], r[1] - r[0], r[0]);
    if (NOSTREAM) { if (res == 123.456) a.out[i0] = res; } else if (live) {
      if (SP == 6) { double* op = &a.out[i0]; asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(op), "v"(res) : "memory"); }
      else if (SP == 2 || SP == 3) __builtin_nontemporal_store(res, &a.out[i0]); else a.out[i0] = res; }
  }
}

// streaming kernel for the concurrency experiment: out = x + y + z with 16-B accesses
__global__ void __launch_bounds__(1024) k_stream16(const double2* x, const double2* y, const double2* z, double2* o, size_t n2) {
  extern __shared__ unsigned char pad[];
  const size_t nthreads = (size_t)gridDim.x * 1024;
  for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n2; i += nthreads) {
    double2 a = x[i], b = y[i], c = z[i];
    double2 r; r.x = a.x + b.x + c.x; r.y = a.y + b.y + c.y; o[i] = r;
  }
}


// Two points per lane: coordinates and results move as 16-B vectors (dwordx4); the quad-cooperative
// gather is run once per point of the lane.
template <int SI, int SJ, int SK>
__global__ void __launch_bounds__(256) k_coop2(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[256 * 80 + 256 * 16];
  const unsigned lane = threadIdx.x, q = lane & 3, quad = lane >> 2;
  d2u* lds_piece = reinterpret_cast<d2u*>(lds_raw);
  unsigned* lds_off = reinterpret_cast<unsigned*>(lds_raw + 256 * 80);
  const size_t nthreads = (size_t)gridDim.x * 256;
  const size_t npairs = a.npts / 2;
  const size_t niter = (npairs + nthreads - 1) / nthreads;
  typedef double d2a __attribute__((ext_vector_type(2)));
  for (size_t it = 0; it < niter; ++it) {
    const size_t j0 = it * nthreads + (size_t)blockIdx.x * 256 + lane;  // pair index
    const bool live = j0 < npairs;
    d2a xv[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) { xv[d].x = a.start; xv[d].y = a.start; if (live) xv[d] = reinterpret_cast<const d2a*>(a.obs[d])[j0]; }
    d2a resv;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double t[3]; int loc[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const double x = h == 0 ? xv[d].x : xv[d].y;
        double floc = __builtin_floor((x - a.start) / a.step);
        floc = floc > 0 ? floc : 0; int l = (int)floc; l = l < a.n - 2 ? l : a.n - 2; loc[d] = l;
        t[d] = (x - __builtin_fma(a.step, (double)l, a.start)) / a.step;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) lds_off[(quad * 4 + p) * 4 + q] = brick_addr<SI, SJ, SK>(a, loc[0], loc[1], loc[2], p >> 1, p & 1, 0);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      uint4 toff = *reinterpret_cast<uint4*>(&lds_off[(quad * 4 + q) * 4]);
      d2u pc[4];
      pc[0] = *(const d2u*)(a.vals + toff.x); pc[1] = *(const d2u*)(a.vals + toff.y);
      pc[2] = *(const d2u*)(a.vals + toff.z); pc[3] = *(const d2u*)(a.vals + toff.w);
#pragma unroll
      for (int r = 0; r < 4; ++r) lds_piece[(quad * 4 + r) * 5 + q] = pc[r];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double v[2][2][2];
#pragma unroll
      for (int p = 0; p < 4; ++p) { d2u w = lds_piece[(quad * 4 + q) * 5 + p]; v[p >> 1][p & 1][0] = w.x; v[p >> 1][p & 1][1] = w.y; }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      double r[2];
#pragma unroll
      for (int dk = 0; dk < 2; ++dk) {
        double c0 = __builtin_fma(t[0], v[1][0][dk] - v[0][0][dk], v[0][0][dk]);
        double c1 = __builtin_fma(t[0], v[1][1][dk] - v[0][1][dk], v[0][1][dk]);
        r[dk] = __builtin_fma(t[1], c1 - c0, c0);
      }
      const double res = __builtin_fma(t[2], r[1] - r[0], r[0]);
      if (h == 0) resv.x = res; else resv.y = res;
    }
    if (live) reinterpret_cast<d2a*>(a.out)[j0] = resv;
  }
}

// Cell-major layout: every cell's 8 corners as 64 contiguous, 64-B aligned bytes (8x the grid):
// piece p = (di,dj) at cell*8 + p*2.  Quad-cooperative gather as in k_coop.
template <bool NOSTREAM>
__global__ void __launch_bounds__(256) k_coop_cell(const Args a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[256 * 80 + 256 * 16];
  const unsigned lane = threadIdx.x, q = lane & 3, quad = lane >> 2;
  d2u* lds_piece = reinterpret_cast<d2u*>(lds_raw);
  unsigned* lds_off = reinterpret_cast<unsigned*>(lds_raw + 256 * 80);
  const size_t nthreads = (size_t)gridDim.x * 256;
  const size_t niter = (a.npts + nthreads - 1) / nthreads;
  for (size_t it = 0; it < niter; ++it) {
    const size_t i0 = it * nthreads + (size_t)blockIdx.x * 256 + lane;
    const bool live = i0 < a.npts;
    double x[3], t[3]; int loc[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (NOSTREAM) {
        unsigned long long z = (i0 * 3 + d) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        x[d] = -1.0 + 2.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0);
      } else x[d] = live ? a.obs[d][i0] : a.start;
      double floc = __builtin_floor((x[d] - a.start) / a.step);
      floc = floc > 0 ? floc : 0; int l = (int)floc; l = l < a.n - 2 ? l : a.n - 2; loc[d] = l;
      t[d] = (x[d] - __builtin_fma(a.step, (double)l, a.start)) / a.step;
    }
    const unsigned cell = ((unsigned)loc[0] * (a.n - 1) + loc[1]) * (a.n - 1) + loc[2];
    lds_off[quad * 4 + q] = cell * 8u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const uint4 cb = *reinterpret_cast<uint4*>(&lds_off[quad * 4]);
    d2u pc[4];
    pc[0] = *(const d2u*)(a.vals + cb.x + q * 2);
    pc[1] = *(const d2u*)(a.vals + cb.y + q * 2);
    pc[2] = *(const d2u*)(a.vals + cb.z + q * 2);
    pc[3] = *(const d2u*)(a.vals + cb.w + q * 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_piece[(quad * 4 + r) * 5 + q] = pc[r];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double v[2][2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) { d2u w = lds_piece[(quad * 4 + q) * 5 + p]; v[p >> 1][p & 1][0] = w.x; v[p >> 1][p & 1][1] = w.y; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    double r[2];
#pragma unroll
    for (int dk = 0; dk < 2; ++dk) {
      double c0 = __builtin_fma(t[0], v[1][0][dk] - v[0][0][dk], v[0][0][dk]);
      double c1 = __builtin_fma(t[0], v[1][1][dk] - v[0][1][dk], v[0][1][dk]);
      r[dk] = __builtin_fma(t[1], c1 - c0, c0);
    }
    double res = __builtin_fma(t[2], r[1] - r[0], r[0]);
    if (NOSTREAM) { if (res == 123.456) a.out[i0] = res; } else if (live) a.out[i0] = res;
  }
}

static std::vector<double> make_bricks(const std::vector<double>& v, int n, int SI, int SJ, int SK, unsigned& nbi, unsigned& nbj, unsigned& nbk) {
  nbi = (n - 2) / SI + 2; nbj = (n - 2) / SJ + 2; nbk = (n - 2) / SK + 2;
  std::vector<double> b((size_t)nbi * nbj * nbk * 16, 0.0);
  for (unsigned bi = 0; bi < nbi; ++bi) for (unsigned bj = 0; bj < nbj; ++bj) for (unsigned bk = 0; bk < nbk; ++bk)
    for (int oi = 0; oi < 2; ++oi) for (int oj = 0; oj < 2; ++oj) for (int ok = 0; ok < 4; ++ok) {
      int i = bi * SI + oi, j = bj * SJ + oj, k = bk * SK + ok;
      if (i < n && j < n && k < n) b[(((size_t)bi * nbj + bj) * nbk + bk) * 16 + (oi * 2 + oj) * 4 + ok] = v[((size_t)i * n + j) * n + k];
    }
  return b;
}

static std::vector<double> make_tiles(const std::vector<double>& v, int n, unsigned& ntj, unsigned& ntk) {
  ntj = (n - 2) / 3 + 1; ntk = (n - 2) / 3 + 1;
  std::vector<double> b((size_t)n * ntj * ntk * 16, 0.0);
  for (int i = 0; i < n; ++i) for (unsigned tj = 0; tj < ntj; ++tj) for (unsigned tk = 0; tk < ntk; ++tk)
    for (int oj = 0; oj < 4; ++oj) for (int ok = 0; ok < 4; ++ok) {
      int j = tj * 3 + oj, k = tk * 3 + ok;
      if (j < n && k < n) b[(((size_t)i * ntj + tj) * ntk + tk) * 16 + oj * 4 + ok] = v[((size_t)i * n + j) * n + k];
    }
  return b;
}

static double time_it(const char* name, std::function<void()> fn, size_t P, int reps = 7) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  fn(); CK(hipDeviceSynchronize());
  std::vector<float> ms;
  for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float m; CK(hipEventElapsedTime(&m, a, b)); ms.push_back(m); }
  std::sort(ms.begin(), ms.end());
  printf("%-46s med %7.3f ms  %7.1f Gpts/s\n", name, ms[ms.size() / 2], P / ms[ms.size() / 2] / 1e6); fflush(stdout);
  return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
  size_t P = argc > 1 ? (size_t)atof(argv[1]) : 100000000;
  int n = argc > 2 ? atoi(argv[2]) : 64;
  size_t G = (size_t)n * n * n;
  std::vector<double> hv(G), hx(P);
  srand(1);
  for (auto& v : hv) v = rand() / (double)RAND_MAX * 2 - 1;
  double *dv, *dx[3], *dout, *dref;
  CK(hipMalloc(&dv, G * 8)); CK(hipMemcpy(dv, hv.data(), G * 8, hipMemcpyHostToDevice));
  for (int d = 0; d < 3; ++d) {
    uint64_t s = 0x9E3779B97F4A7C15ull * (d + 1);
    for (size_t i = 0; i < P; ++i) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; hx[i] = -1.0 + 2.0 * (double)(z >> 11) * (1.0 / 9007199254740992.0); }
    if (getenv("UNCACHED_OBS")) { CK(hipExtMallocWithFlags((void**)&dx[d], P * 8, hipDeviceMallocUncached)); }
    else CK(hipMalloc(&dx[d], P * 8));
    CK(hipMemcpy(dx[d], hx.data(), P * 8, hipMemcpyHostToDevice));
  }
  if (getenv("UNCACHED_OUT")) { CK(hipExtMallocWithFlags((void**)&dout, P * 8, hipDeviceMallocUncached)); }
  else CK(hipMalloc(&dout, P * 8));
  CK(hipMalloc(&dref, P * 8));
  Args a; a.vals = dv; a.out = dref; a.npts = P; a.start = -1.0; a.step = 2.0 / (n - 1); a.rinv = 1.0 / a.step; a.n = n; a.nbj = a.nbk = 0; a.in_mask = ~(size_t)0; a.out_mask = ~(size_t)0;
  for (int d = 0; d < 3; ++d) a.obs[d] = dx[d];
  printf("P=%zu grid=%d^3 (%.1f MiB row-major)\n", P, n, G * 8 / 1048576.0);
  const unsigned BLK = 2048 * 2;
  if (argc > 3 && argv[3][0] == 'w') {  // wide (16-B) coordinate/result accesses, two points per lane
    a.out = dref;
    hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, a); CK(hipDeviceSynchronize());
    a.out = dout;
    unsigned nbi, nbj, nbk;
    std::vector<double> b = make_bricks(hv, n, 1, 2, 3, nbi, nbj, nbk);
    double* db; CK(hipMalloc(&db, b.size() * 8)); CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));
    Args c = a; c.vals = db; c.nbj = nbj; c.nbk = nbk;
    for (unsigned blk : {1024u, 2048u, 4096u}) {
      char name[128];
      snprintf(name, sizeof name, "coop(1,2,3) 1 pt/lane   %u blocks", blk);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<1, 2, 3, false>), dim3(blk), dim3(256), 0, 0, c); }, P);
      CK(hipMemset(dout, 0, P * 8));
      snprintf(name, sizeof name, "coop(1,2,3) 2 pts/lane  %u blocks", blk);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop2<1, 2, 3>), dim3(blk), dim3(256), 0, 0, c); }, P);
    }
    std::vector<double> ref(1 << 20), got(1 << 20);
    CK(hipMemcpy(ref.data(), dref, ref.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t q = 0; q < got.size(); ++q) bad += got[q] != ref[q];
    printf("2 pts/lane mismatches: %zu\n", bad);
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'e') {  // cell-major 64-B layout vs fully overlapped bricks
    a.out = dref;
    hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, a); CK(hipDeviceSynchronize());
    a.out = dout;
    {
      unsigned nbi, nbj, nbk;
      std::vector<double> b = make_bricks(hv, n, 1, 1, 3, nbi, nbj, nbk);
      double* db; CK(hipMalloc(&db, b.size() * 8)); CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));
      Args c = a; c.vals = db; c.nbj = nbj; c.nbk = nbk;
      char name[128]; snprintf(name, sizeof name, "coop brick(1,1,3) %.1f MiB full", b.size() * 8 / 1048576.0);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<1, 1, 3, false>), dim3(BLK), dim3(256), 0, 0, c); }, P);
      time_it("coop brick(1,1,3) gather-only", [&] { hipLaunchKernelGGL((k_coop<1, 1, 3, true>), dim3(BLK), dim3(256), 0, 0, c); }, P);
      CK(hipFree(db));
    }
    {
      size_t nc = (size_t)(n - 1) * (n - 1) * (n - 1);
      std::vector<double> cm(nc * 8);
      for (int i = 0; i < n - 1; ++i) for (int j = 0; j < n - 1; ++j) for (int k = 0; k < n - 1; ++k)
        for (int p = 0; p < 4; ++p) for (int dk = 0; dk < 2; ++dk)
          cm[(((size_t)i * (n - 1) + j) * (n - 1) + k) * 8 + p * 2 + dk] = hv[((size_t)(i + (p >> 1)) * n + (j + (p & 1))) * n + k + dk];
      double* dc; CK(hipMalloc(&dc, cm.size() * 8)); CK(hipMemcpy(dc, cm.data(), cm.size() * 8, hipMemcpyHostToDevice));
      Args c = a; c.vals = dc;
      CK(hipMemset(dout, 0, P * 8));
      char name[128]; snprintf(name, sizeof name, "coop cell-major 64B %.1f MiB full", cm.size() * 8 / 1048576.0);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop_cell<false>), dim3(BLK), dim3(256), 0, 0, c); }, P);
      std::vector<double> ref(1 << 20), got(1 << 20);
      CK(hipMemcpy(ref.data(), dref, ref.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t q = 0; q < got.size(); ++q) bad += got[q] != ref[q];
      snprintf(name, sizeof name, "coop cell-major 64B gather-only [mismatch %zu]", bad);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop_cell<true>), dim3(BLK), dim3(256), 0, 0, c); }, P);
      CK(hipFree(dc));
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 't') {  // tile layout (2 lines, 1.78x) against bricks (1,2,3) and (2,2,3)
    a.out = dout;
    unsigned nbi, nbj, nbk, ntj, ntk;
    std::vector<double> t = make_tiles(hv, n, ntj, ntk);
    double* dt; CK(hipMalloc(&dt, t.size() * 8)); CK(hipMemcpy(dt, t.data(), t.size() * 8, hipMemcpyHostToDevice));
    Args ct = a; ct.vals = dt; ct.nbj = ntj; ct.nbk = ntk; ct.out = dout;
    std::vector<double> b12 = make_bricks(hv, n, 1, 2, 3, nbi, nbj, nbk);
    double* d12; CK(hipMalloc(&d12, b12.size() * 8)); CK(hipMemcpy(d12, b12.data(), b12.size() * 8, hipMemcpyHostToDevice));
    Args c12 = a; c12.vals = d12; c12.nbj = nbj; c12.nbk = nbk; c12.out = dout;
    std::vector<double> b22 = make_bricks(hv, n, 2, 2, 3, nbi, nbj, nbk);
    double* d22; CK(hipMalloc(&d22, b22.size() * 8)); CK(hipMemcpy(d22, b22.data(), b22.size() * 8, hipMemcpyHostToDevice));
    Args c22 = a; c22.vals = d22; c22.nbj = nbj; c22.nbk = nbk; c22.out = dout;
    printf("tables: tiles %.2f MiB, bricks(1,2,3) %.2f MiB, bricks(2,2,3) %.2f MiB\n", t.size() * 8 / 1048576.0, b12.size() * 8 / 1048576.0, b22.size() * 8 / 1048576.0);
    // reference result for a correctness check of the tile kernel
    hipLaunchKernelGGL((k_coop<1, 2, 3, false, false, 2>), dim3(BLK), dim3(256), 0, 0, c12); CK(hipDeviceSynchronize());
    std::vector<double> r1(1 << 20), r2(1 << 20);
    CK(hipMemcpy(r1.data(), dout, r1.size() * 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL((k_coop<0, 0, 0, false, false, 2>), dim3(BLK), dim3(256), 0, 0, ct); CK(hipDeviceSynchronize());
    CK(hipMemcpy(r2.data(), dout, r2.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t q = 0; q < r1.size(); ++q) bad += r1[q] != r2[q];
    printf("tile kernel vs bricks(1,2,3): %zu mismatches in the first 2^20 results\n", bad);
    for (int blk : {2048, 8192, 65536}) {
      char name[128];
      snprintf(name, sizeof name, "bricks(1,2,3) nt full      %6d blk", blk);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<1, 2, 3, false, false, 2>), dim3(blk), dim3(256), 0, 0, c12); }, P);
      snprintf(name, sizeof name, "bricks(2,2,3) nt full      %6d blk", blk);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<2, 2, 3, false, false, 2>), dim3(blk), dim3(256), 0, 0, c22); }, P);
      snprintf(name, sizeof name, "tiles 4x4(j,k) nt full     %6d blk", blk);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<0, 0, 0, false, false, 2>), dim3(blk), dim3(256), 0, 0, ct); }, P);
    }
    time_it("bricks(1,2,3) gather-only     2048 blk", [&] { hipLaunchKernelGGL((k_coop<1, 2, 3, true>), dim3(2048), dim3(256), 0, 0, c12); }, P);
    time_it("tiles 4x4(j,k) gather-only    2048 blk", [&] { hipLaunchKernelGGL((k_coop<0, 0, 0, true>), dim3(2048), dim3(256), 0, 0, ct); }, P);
    return 0;
  }
  if (argc > 3 && argv[3][0] == 's') {  // streaming-rate calibration
    for (int blk : {1024, 2048, 4096, 8192}) {
      char name[128];
      snprintf(name, sizeof name, "stream 16B/lane x+y+z->o  %d blocks x 1024", blk);
      time_it(name, [&] { hipLaunchKernelGGL(k_stream16, dim3(blk), dim3(1024), 0, 0, (const double2*)dx[0], (const double2*)dx[1], (const double2*)dx[2], (double2*)dref, P / 2); }, P);
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'c') {  // concurrency: gather-only on GC CUs + stream on SC CUs
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    a.out = dout;
    const int LDSB = 100 * 1024;  // forces one workgroup per CU
    for (int sc : {32, 64, 96}) {
      int gc = 256 - sc;
      char name[128];
      snprintf(name, sizeof name, "gather-only alone on %d CUs", gc);
      time_it(name, [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, true>), dim3(gc), dim3(256), LDSB, s1, a); CK(hipStreamSynchronize(s1)); }, P);
      snprintf(name, sizeof name, "stream16 alone on %d CUs", sc);
      time_it(name, [&] { hipLaunchKernelGGL(k_stream16, dim3(sc), dim3(1024), LDSB, s2, (const double2*)dx[0], (const double2*)dx[1], (const double2*)dx[2], (double2*)dref, P / 2); CK(hipStreamSynchronize(s2)); }, P);
      snprintf(name, sizeof name, "both concurrently (%d + %d CUs)", gc, sc);
      time_it(name, [&] {
        hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, true>), dim3(gc), dim3(256), LDSB, s1, a);
        hipLaunchKernelGGL(k_stream16, dim3(sc), dim3(1024), LDSB, s2, (const double2*)dx[0], (const double2*)dx[1], (const double2*)dx[2], (double2*)dref, P / 2);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); }, P);
    }
    return 0;
  }
  if (argc > 3 && argv[3][0] == 'o') {  // occupancy sweep: extra dynamic LDS limits resident blocks per CU
    unsigned nbi, nbj, nbk;
    std::vector<double> b = make_bricks(hv, n, 2, 2, 3, nbi, nbj, nbk);
    double* db; CK(hipMalloc(&db, b.size() * 8)); CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));
    Args c = a; c.vals = db; c.nbj = nbj; c.nbk = nbk; c.out = dout; a.out = dout;
    int lds_kb[] = {0, 16, 24, 36, 60, 100};
    for (int kb : lds_kb) {
      char name[128];
      snprintf(name, sizeof name, "row-major full  +%3d KB dyn LDS", kb);
      time_it(name, [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), kb * 1024, 0, a); }, P);
      snprintf(name, sizeof name, "row-major gather-only +%3d KB dyn LDS", kb);
      time_it(name, [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, true>), dim3(BLK), dim3(256), kb * 1024, 0, a); }, P);
      snprintf(name, sizeof name, "coop(2,2,3) full +%3d KB dyn LDS", kb);
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<2, 2, 3, false>), dim3(BLK), dim3(256), kb * 1024, 0, c); }, P);
    }
    return 0;
  }
  if (argc > 3) {  // profiling mode: a handful of kernels, two launches each
    unsigned nbi, nbj, nbk;
    std::vector<double> b = make_bricks(hv, n, 2, 2, 3, nbi, nbj, nbk);
    double* db; CK(hipMalloc(&db, b.size() * 8)); CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice));
    Args c = a; c.vals = db; c.nbj = nbj; c.nbk = nbk; c.out = dout; a.out = dout;
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, a);
      hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, true>), dim3(BLK), dim3(256), 0, 0, a);
      hipLaunchKernelGGL((k_lay<1, 2, 2, 3, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, c);
      hipLaunchKernelGGL((k_coop<2, 2, 3, false>), dim3(BLK), dim3(256), 0, 0, c);
      hipLaunchKernelGGL((k_coop<2, 2, 3, true>), dim3(BLK), dim3(256), 0, 0, c);
      CK(hipDeviceSynchronize());
    }
    return 0;
  }
  hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, a); CK(hipDeviceSynchronize());
  a.out = dout;
  time_it("row-major plain            full", [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, a); }, P);
  time_it("row-major plain            gather-only", [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, true>), dim3(BLK), dim3(256), 0, 0, a); }, P);
  { Args w = a; w.in_mask = 0xFFFF; time_it("row-major plain  obs wrapped 64K pts", [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, w); }, P); }
  { Args w = a; w.out_mask = 0xFFFF; time_it("row-major plain  out wrapped 64K pts", [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, w); }, P); }
  { Args w = a; w.in_mask = 0xFFFF; w.out_mask = 0xFFFF; time_it("row-major plain  both wrapped 64K pts", [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, w); }, P); }
  { Args w = a; w.in_mask = 0xFFFFFF; w.out_mask = 0xFFFFFF; time_it("row-major plain  both wrapped 16M pts", [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, w); }, P); }
  time_it("row-major nt               full", [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_NT, false>), dim3(BLK), dim3(256), 0, 0, a); }, P);
  time_it("row-major nt               gather-only", [&] { hipLaunchKernelGGL((k_lay<0, 2, 2, 4, LD_NT, true>), dim3(BLK), dim3(256), 0, 0, a); }, P);

  std::vector<double> ref(1 << 20), got(1 << 20);
  CK(hipMemcpy(ref.data(), dref, ref.size() * 8, hipMemcpyDeviceToHost));
#define LAYOUT_RUN(SI, SJ, SK)                                                                                   \
  {                                                                                                              \
    unsigned nbi, nbj, nbk;                                                                                      \
    std::vector<double> b = make_bricks(hv, n, SI, SJ, SK, nbi, nbj, nbk);                                       \
    double* db; CK(hipMalloc(&db, b.size() * 8)); CK(hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice)); \
    Args c = a; c.vals = db; c.nbj = nbj; c.nbk = nbk;                                                           \
    char name[128];                                                                                              \
    CK(hipMemset(dout, 0, P * 8));                                                                               \
    snprintf(name, sizeof name, "brick step(%d,%d,%d) %.1f MiB plain full", SI, SJ, SK, b.size() * 8 / 1048576.0); \
    time_it(name, [&] { hipLaunchKernelGGL((k_lay<1, SI, SJ, SK, LD_PLAIN, false>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
    CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));                                      \
    size_t bad = 0; for (size_t q = 0; q < got.size(); ++q) bad += got[q] != ref[q];                            \
    snprintf(name, sizeof name, "brick step(%d,%d,%d)          gather-only [mismatch %zu]", SI, SJ, SK, bad);    \
    time_it(name, [&] { hipLaunchKernelGGL((k_lay<1, SI, SJ, SK, LD_PLAIN, true>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
    if (SK == 3) {                                                                                               \
      CK(hipMemset(dout, 0, P * 8));                                                                             \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    full", SI, SJ, SK);                              \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, false>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
      CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));                                    \
      bad = 0; for (size_t q = 0; q < got.size(); ++q) bad += got[q] != ref[q];                                  \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    full nt-loads", SI, SJ, SK);   \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, false, false, 1>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    full nt-loads+nt-stores", SI, SJ, SK);   \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, false, false, 2>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    full sc0sc1-loads(serial)", SI, SJ, SK);   \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, false, false, 4>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    full sc1-loads(serial)", SI, SJ, SK);   \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, false, false, 5>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    full sc0sc1 ld+st", SI, SJ, SK);   \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, false, false, 6>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    full nt-stores", SI, SJ, SK);   \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, false, false, 3>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    full+prefetch [mismatch %zu]", SI, SJ, SK, bad);   \
      CK(hipMemset(dout, 0, P * 8));                                                                             \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, false, true>), dim3(BLK), dim3(256), 0, 0, c); }, P); \
      CK(hipMemcpy(got.data(), dout, got.size() * 8, hipMemcpyDeviceToHost));                                    \
      bad = 0; for (size_t q = 0; q < got.size(); ++q) bad += got[q] != ref[q];                                  \
      snprintf(name, sizeof name, "  coop quad step(%d,%d,%d)    gather-only [pf mismatch %zu]", SI, SJ, SK, bad);   \
      time_it(name, [&] { hipLaunchKernelGGL((k_coop<SI, SJ, SK, true>), dim3(BLK), dim3(256), 0, 0, c); }, P);  \
    }                                                                                                            \
    CK(hipFree(db));                                                                                             \
  }
  LAYOUT_RUN(2, 2, 4)
  LAYOUT_RUN(2, 2, 3)
  LAYOUT_RUN(1, 2, 4)
  LAYOUT_RUN(1, 2, 3)
  LAYOUT_RUN(1, 1, 4)
  LAYOUT_RUN(1, 1, 3)
  return 0;
}
