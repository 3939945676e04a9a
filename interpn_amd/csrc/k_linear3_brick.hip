// 3-D multilinear (regular and rectilinear), f64, bricked grid + quad-cooperative gather.
//
// Why: with the grid in C order a point's 8 corners lie on 4 different 128-B lines, and on
// MI355X the per-XCD L2 -> L1 line rate (16 lines/clk/XCD, ~2.7e11 lines/s chip-wide), not HBM,
// bounds the kernel (profiles/r01_tune_*: TCP_TCC_READ_REQ = 4.3 lines/point, all L2 hits).
// Here the handle keeps a second copy of the grid in 2(i) x 2(j) x 4(k) bricks of one line each,
// overlapped so that a cell's corners span fewer bricks (step 3 along k: a k-pair never leaves
// a brick row; step 1 along i and/or j duplicates planes/rows), and the four lanes of a quad
// fetch the four 16-B pieces of ONE point per load instruction, so pieces on the same line
// become a single L2 request.  Pieces are transposed back through LDS and every lane finishes
// its own point with the reference's arithmetic, so results are bit-identical to the C-order
// kernels (same values, same operation order).
#include "rect_args.h"

namespace interpn {

struct Brick3Args {
  const double* bricks;
  const double* obs[3];
  double* out;
  unsigned long long* first_bad;
  size_t npts;
  double start[3];
  double step[3];
  int n[3];
  AxisArgs<double, 3> ax;
  unsigned nbj, nbk;
};

typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

template <int SI, int SJ>
__device__ __forceinline__ unsigned brick_piece(const Brick3Args& a, int i, int j, unsigned kpart, int di, int dj) {
  int bi, oi, bj, oj;
  if (SI == 1) { bi = i + 0; oi = di; bi = i; }
  else { bi = i >> 1; oi = (i & 1) + di; if (oi == 2) { bi += 1; oi = 0; } }
  if (SJ == 1) { bj = j; oj = dj; }
  else { bj = j >> 1; oj = (j & 1) + dj; if (oj == 2) { bj += 1; oj = 0; } }
  return ((unsigned)(bi * (int)a.nbj + bj) * a.nbk) * 16u + (unsigned)((oi * 2 + oj) * 4) + kpart;
}

constexpr int kPieceRow = 5;  // d2u slots per point row in LDS (4 used + 1 pad against bank conflicts)

template <bool RECT, bool FMA, int SI, int SJ>
__global__ void __launch_bounds__(kBlock) k_linear3_brick(const Brick3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  d2u* lds_piece = reinterpret_cast<d2u*>(smem_raw);                                   // [quad][r][kPieceRow]
  unsigned* lds_off = reinterpret_cast<unsigned*>(smem_raw + kBlock * kPieceRow * 16);  // [quad][piece][r]
  unsigned char* lds_axes = smem_raw + kBlock * kPieceRow * 16 + kBlock * 16;
  if (RECT && a.ax.use_lds) stage_axes<double, 3>(a.ax, lds_axes);
  const unsigned char* axis_base = (RECT && a.ax.use_lds) ? lds_axes : a.ax.image;
  const unsigned lane = threadIdx.x;
  const unsigned q = lane & 3;
  const unsigned quad = lane >> 2;
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  const size_t niter = (a.npts + nthreads - 1) / nthreads;
  for (size_t it = 0; it < niter; ++it) {
    const size_t i0 = it * nthreads + (size_t)blockIdx.x * kBlock + lane;
    const bool live = i0 < a.npts;
    double t[3];
    int loc[3];
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (RECT) {
        const double x = live ? a.obs[d][i0] : 0.0;
        const Axis<double> ax = make_axis<double, 3>(a.ax, axis_base, d);
        int l = axis_partition_point<double>(ax, x) - 1;       // multilinear/rectilinear.rs:363
        l = l > 0 ? l : 0;
        l = l < a.n[d] - 2 ? l : a.n[d] - 2;                   // rectilinear.rs:365-367
        const double x0 = ax.g[l];
        const double x1 = ax.g[l + 1];
        const double step = x1 - x0;
        t[d] = (x - x0) / step;                                // rectilinear.rs:310-313
        loc[d] = l;
      } else {
        const double x = live ? a.obs[d][i0] : a.start[d];
        double floc;
        ok &= regular_floc<double>(x, a.start[d], a.step[d], &floc);  // multilinear/regular.rs:415-418
        const int l = clamp_loc<double>(floc, a.n[d] - 2);            // regular.rs:420-422
        const double izl = mul_add<FMA>(a.step[d], (double)l, a.start[d]);  // regular.rs:334-337
        t[d] = (x - izl) / a.step[d];                                  // regular.rs:339
        loc[d] = l;
      }
    }
    if (!RECT && !ok && live) atomicMin(a.first_bad, (unsigned long long)i0);
    // Offsets of my point's four pieces -> LDS, transposed: lane q reads piece q of points 0..3.
    const unsigned bk = (unsigned)loc[2] / 3u;
    const unsigned kpart = bk * 16u + ((unsigned)loc[2] - bk * 3u);
#pragma unroll
    for (int p = 0; p < 4; ++p)
      lds_off[(quad * 4 + p) * 4 + q] = brick_piece<SI, SJ>(a, loc[0], loc[1], kpart, p >> 1, p & 1);
    __builtin_amdgcn_wave_barrier();
    const uint4 toff = *reinterpret_cast<const uint4*>(&lds_off[(quad * 4 + q) * 4]);
    d2u pc[4];
    pc[0] = *reinterpret_cast<const d2u*>(a.bricks + toff.x);
    pc[1] = *reinterpret_cast<const d2u*>(a.bricks + toff.y);
    pc[2] = *reinterpret_cast<const d2u*>(a.bricks + toff.z);
    pc[3] = *reinterpret_cast<const d2u*>(a.bricks + toff.w);
#pragma unroll
    for (int r = 0; r < 4; ++r) lds_piece[(quad * 4 + r) * kPieceRow + q] = pc[r];
    __builtin_amdgcn_wave_barrier();
    double v[2][2][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const d2u w = lds_piece[(quad * 4 + q) * kPieceRow + p];
      v[p >> 1][p & 1][0] = w.x;
      v[p >> 1][p & 1][1] = w.y;
    }
    __builtin_amdgcn_wave_barrier();
    // Reference tree (multilinear/regular.rs:347-403): dim 0 first, dim 2 last.
    double r[2];
#pragma unroll
    for (int dk = 0; dk < 2; ++dk) {
      const double c0 = mul_add<FMA>(t[0], v[1][0][dk] - v[0][0][dk], v[0][0][dk]);
      const double c1 = mul_add<FMA>(t[0], v[1][1][dk] - v[0][1][dk], v[0][1][dk]);
      r[dk] = mul_add<FMA>(t[1], c1 - c0, c0);
    }
    const double res = mul_add<FMA>(t[2], r[1] - r[0], r[0]);
    if (live) a.out[i0] = res;
  }
}

// Brick table builder: one thread per brick element.
__global__ void __launch_bounds__(kBlock) k_build_bricks3(const double* __restrict__ vals, double* __restrict__ bricks,
                                                          int n0, int n1, int n2, int si, int sj, int sk,
                                                          unsigned nbi, unsigned nbj, unsigned nbk) {
  const size_t total = (size_t)nbi * nbj * nbk * 16;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const unsigned within = (unsigned)(e & 15);
    size_t b = e >> 4;
    const unsigned bk = (unsigned)(b % nbk); b /= nbk;
    const unsigned bj = (unsigned)(b % nbj); b /= nbj;
    const unsigned bi = (unsigned)b;
    const int i = (int)bi * si + (int)(within >> 3);
    const int j = (int)bj * sj + (int)((within >> 2) & 1);
    const int k = (int)bk * sk + (int)(within & 3);
    double v = 0.0;
    if (i < n0 && j < n1 && k < n2) v = vals[((size_t)i * n1 + j) * n2 + k];
    bricks[e] = v;
  }
}

void brick3_geometry(const int n[3], int si, int sj, unsigned nb[3], size_t* bytes) {
  nb[0] = (unsigned)((n[0] - 2) / si + 2);
  nb[1] = (unsigned)((n[1] - 2) / sj + 2);
  nb[2] = (unsigned)((n[2] - 2) / 3 + 2);
  *bytes = (size_t)nb[0] * nb[1] * nb[2] * 16 * sizeof(double);
}

hipError_t build_bricks3(const GridDesc& g, void* bricks, hipStream_t stream) {
  const size_t total = (size_t)g.brick_nb[0] * g.brick_nb[1] * g.brick_nb[2] * 16;
  size_t blocks = (total + kBlock - 1) / kBlock;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(k_build_bricks3, dim3((unsigned)blocks), dim3(kBlock), 0, stream, static_cast<const double*>(g.vals),
                     static_cast<double*>(bricks), g.n[0], g.n[1], g.n[2], g.brick_step[0], g.brick_step[1], 3,
                     g.brick_nb[0], g.brick_nb[1], g.brick_nb[2]);
  return hipGetLastError();
}

template <bool RECT, bool FMA>
static hipError_t launch_steps(const GridDesc& g, const Brick3Args& a, size_t lds, unsigned blocks, hipStream_t stream) {
  const int si = g.brick_step[0], sj = g.brick_step[1];
  if (si == 1 && sj == 1) hipLaunchKernelGGL((k_linear3_brick<RECT, FMA, 1, 1>), dim3(blocks), dim3(kBlock), lds, stream, a);
  else if (si == 1 && sj == 2) hipLaunchKernelGGL((k_linear3_brick<RECT, FMA, 1, 2>), dim3(blocks), dim3(kBlock), lds, stream, a);
  else hipLaunchKernelGGL((k_linear3_brick<RECT, FMA, 2, 2>), dim3(blocks), dim3(kBlock), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_linear3_brick(const GridDesc& g, const double* const* obs, double* out, size_t npts,
                                unsigned long long* first_bad, hipStream_t stream) {
  Brick3Args a;
  a.bricks = static_cast<const double*>(g.bricks);
  a.out = out;
  a.first_bad = first_bad;
  a.npts = npts;
  for (int d = 0; d < 3; ++d) {
    a.obs[d] = obs[d];
    a.start[d] = g.start[d];
    a.step[d] = g.step[d];
    a.n[d] = g.n[d];
  }
  a.nbj = g.brick_nb[1];
  a.nbk = g.brick_nb[2];
  size_t lds = (size_t)kBlock * kPieceRow * 16 + (size_t)kBlock * 16;
  a.ax.use_lds = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  if (g.kind == kRectilinear) lds += fill_axis_args<double, 3>(g, a.ax);
  const unsigned blocks = grid_blocks(npts, 1, g.cfg);
  if (g.kind == kRegular)
    return g.fma ? launch_steps<false, true>(g, a, lds, blocks, stream) : launch_steps<false, false>(g, a, lds, blocks, stream);
  return g.fma ? launch_steps<true, true>(g, a, lds, blocks, stream) : launch_steps<true, false>(g, a, lds, blocks, stream);
}

}  // namespace interpn
