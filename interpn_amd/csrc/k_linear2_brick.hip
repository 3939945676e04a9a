// 2-D multilinear (regular and rectilinear, f64 and f32) on a bricked copy of the grid with a
// lane-pair cooperative gather.
//
// In C order the four corners of a 2-D cell lie on two rows = two 128-B lines.  Here the grid is
// re-laid in bricks of 2 rows x KW2 columns (KW2 = 8 for f64, 16 for f32; one 128-B line), stepped
// 1 along i (every row pair is stored) and KW2-1 along j (a j-pair never leaves a brick row), i.e.
// 2 * KW2/(KW2-1) = 2.3x the grid, so that a cell's corners are one line.  The two lanes of a pair
// fetch the two row pieces of ONE point per load instruction (same line => one L2 request), then
// swap the piece they hold for the other point with a quad-permute DPP move — no LDS.  Arithmetic
// and operation order are the reference's (src/multilinear/regular.rs:296-404), results are
// bit-identical to the C-order kernel.
#include "lane_axes.h"
#include "rect_args.h"

namespace interpn {

template <typename T> struct Brick2Geom {
  static constexpr int KW = 64 / (int)sizeof(T);  // columns per brick row (64 B)
  static constexpr int SJ = KW - 1;
  static constexpr int ELEMS = 2 * KW;            // 128 B
};

template <typename T>
struct Brick2Args {
  const T* bricks;
  const T* obs[2];
  T* out;
  unsigned long long* first_bad;
  size_t npts;
  T start[2];
  T step[2];
  int n[2];
  AxisArgs<T, 2> ax;
  unsigned nbj;
};

// swap with the neighbouring lane (lane ^ 1): quad_perm [1,0,3,2]
__device__ __forceinline__ unsigned dpp_swap1(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ float dpp_swap1(float v) { return __uint_as_float(dpp_swap1(__float_as_uint(v))); }
__device__ __forceinline__ double dpp_swap1(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = dpp_swap1((unsigned)b), hi = dpp_swap1((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// AXR != 0 (rectilinear, both axes <= 64 coordinates): axes in lanes, see lane_axes.h.
// PPL = 2: a lane owns two consecutive points, coordinates and results move as 2*sizeof(T)-byte
// vectors (as in the 3-D kernel; needs obs / out aligned to 2*sizeof(T), the launcher checks).
template <typename T, bool RECT, bool FMA, int AXR = 0, int PPL = 1>
__global__ void __launch_bounds__(kBlock) k_linear2_brick(const Brick2Args<T> a) {
  typedef typename LeafVec<T, 2>::type P;
  typedef T T2 __attribute__((ext_vector_type(2)));
  constexpr int KW = Brick2Geom<T>::KW;
  constexpr int SJ = Brick2Geom<T>::SJ;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  LaneAxes<T, 2> la;
  if constexpr (RECT && AXR != 0) la = load_lane_axes<T, 2, AXR>(a.ax);
  else if (RECT && a.ax.use_lds) stage_axes<T, 2>(a.ax, smem_raw);
  const unsigned char* axis_base = (RECT && AXR == 0 && a.ax.use_lds) ? smem_raw : a.ax.image;
  const unsigned lane = threadIdx.x;
  const unsigned q = lane & 1;  // which row piece this lane fetches
  const size_t nthreads = (size_t)gridDim.x * kBlock;
  const size_t nslots = (a.npts + PPL - 1) / PPL;
  const size_t niter = (nslots + nthreads - 1) / nthreads;
  for (size_t it = 0; it < niter; ++it) {
    // every lane runs every iteration (its pair partner may be live)
    const size_t i0 = (it * nthreads + (size_t)blockIdx.x * kBlock + lane) * PPL;
    bool live[PPL];
#pragma unroll
    for (int h = 0; h < PPL; ++h) live[h] = i0 + h < a.npts;
    T xin[PPL][2];
    if (PPL == 2) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        T2 v;
        v.x = RECT ? (T)0 : a.start[d];
        v.y = v.x;
        if (live[PPL - 1]) v = stream_load(reinterpret_cast<const T2*>(a.obs[d] + i0));
        else if (live[0]) v.x = stream_load(a.obs[d] + i0);
        xin[0][d] = v.x;
        xin[PPL - 1][d] = v.y;
      }
    } else {
#pragma unroll
      for (int d = 0; d < 2; ++d) xin[0][d] = live[0] ? stream_load(a.obs[d] + i0) : (RECT ? (T)0 : a.start[d]);
    }
    int cell[PPL][2];
    T x0r[PPL][2], x1r[PPL][2];
    if constexpr (RECT && AXR != 0) lane_axes_locate<T, 2, PPL, AXR>(a.ax, la, xin, cell, x0r, x1r);
    T resv[PPL];
#pragma unroll
    for (int h = 0; h < PPL; ++h) {
      T t[2];
      int loc[2];
      bool ok = true;
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const T x = xin[h][d];
        if (RECT) {
          T x0, x1;
          int l;
          if constexpr (AXR != 0) {
            l = cell[h][d];
            x0 = x0r[h][d];
            x1 = x1r[h][d];
          } else {
            const Axis<T> ax = make_axis<T, 2>(a.ax, axis_base, d);
            l = axis_cell<T>(ax, x, &x0, &x1);  // multilinear/rectilinear.rs:353-370, :310-311
          }
          const T step = x1 - x0;
          t[d] = (x - x0) / step;
          loc[d] = l;
        } else {
          T floc;
          ok &= regular_floc<T>(x, a.start[d], a.step[d], &floc);  // multilinear/regular.rs:415-418
          const int l = clamp_loc<T>(floc, a.n[d] - 2);
          const T izl = mul_add<FMA>(a.step[d], (T)l, a.start[d]);
          t[d] = (x - izl) / a.step[d];
          loc[d] = l;
        }
      }
      if (!RECT && !ok && live[h]) atomicMin(a.first_bad, (unsigned long long)(i0 + h));
      // brick (bi = i, bj = j / SJ); my point's pair starts at column j - bj*SJ of both rows
      const unsigned bj = (unsigned)loc[1] / (unsigned)SJ;
      const unsigned mine = ((unsigned)loc[0] * a.nbj + bj) * (unsigned)Brick2Geom<T>::ELEMS + ((unsigned)loc[1] - bj * (unsigned)SJ);
      const unsigned theirs = dpp_swap1(mine);
      // instruction r fetches point r of the pair (r = 0: even lane's point, r = 1: odd lane's)
      const unsigned off0 = (q == 0 ? mine : theirs) + q * (unsigned)KW;
      const unsigned off1 = (q == 0 ? theirs : mine) + q * (unsigned)KW;
      const P p0 = *reinterpret_cast<const P*>(a.bricks + off0);
      const P p1 = *reinterpret_cast<const P*>(a.bricks + off1);
      // I keep the piece of my own point and trade the other one
      const P keep = q == 0 ? p0 : p1;
      const P send = q == 0 ? p1 : p0;
      P recv;
      recv.x = dpp_swap1(send.x);
      recv.y = dpp_swap1(send.y);
      const P row0 = q == 0 ? keep : recv;  // row i   : v(i, j), v(i, j+1)
      const P row1 = q == 0 ? recv : keep;  // row i+1
      // reference tree: dim 0 first for each j, then dim 1 (multilinear/regular.rs:347-403)
      const T c0 = mul_add<FMA>(t[0], row1.x - row0.x, row0.x);
      const T c1 = mul_add<FMA>(t[0], row1.y - row0.y, row0.y);
      resv[h] = mul_add<FMA>(t[1], c1 - c0, c0);
    }
    if (PPL == 2) {
      if (live[PPL - 1]) {
        T2 v;
        v.x = resv[0];
        v.y = resv[PPL - 1];
        stream_store(reinterpret_cast<T2*>(a.out + i0), v);
      } else if (live[0]) {
        stream_store(a.out + i0, resv[0]);
      }
    } else if (live[0]) {
      stream_store(a.out + i0, resv[0]);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_build_bricks2(const T* __restrict__ vals, T* __restrict__ bricks, int n0, int n1,
                                                          unsigned nbi, unsigned nbj) {
  constexpr int KW = Brick2Geom<T>::KW;
  constexpr int EL = Brick2Geom<T>::ELEMS;
  const size_t total = (size_t)nbi * nbj * EL;
  for (size_t e = (size_t)blockIdx.x * kBlock + threadIdx.x; e < total; e += (size_t)gridDim.x * kBlock) {
    const unsigned within = (unsigned)(e % EL);
    const size_t b = e / EL;
    const unsigned bj = (unsigned)(b % nbj);
    const unsigned bi = (unsigned)(b / nbj);
    const int i = (int)bi + (int)(within / KW);
    const int j = (int)bj * (KW - 1) + (int)(within % KW);
    T v = (T)0;
    if (i < n0 && j < n1) v = vals[(size_t)i * n1 + j];
    bricks[e] = v;
  }
}

void brick2_geometry(const GridDesc& g, unsigned nb[2], size_t* bytes) {
  const int kw = g.dtype == kF64 ? 8 : 16;
  nb[0] = (unsigned)(g.n[0] - 1);
  nb[1] = (unsigned)((g.n[1] - 2) / (kw - 1) + 1);
  *bytes = (size_t)nb[0] * nb[1] * 128;
}

hipError_t build_bricks2(const GridDesc& g, void* bricks, hipStream_t stream) {
  const size_t elems = (size_t)g.brick_nb[0] * g.brick_nb[1] * (g.dtype == kF64 ? 16 : 32);
  size_t blocks = (elems + kBlock - 1) / kBlock;
  if (blocks > 65535) blocks = 65535;
  if (g.dtype == kF64)
    hipLaunchKernelGGL(k_build_bricks2<double>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, static_cast<const double*>(g.vals),
                       static_cast<double*>(bricks), g.n[0], g.n[1], g.brick_nb[0], g.brick_nb[1]);
  else
    hipLaunchKernelGGL(k_build_bricks2<float>, dim3((unsigned)blocks), dim3(kBlock), 0, stream, static_cast<const float*>(g.vals),
                       static_cast<float*>(bricks), g.n[0], g.n[1], g.brick_nb[0], g.brick_nb[1]);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_linear2_brick(const GridDesc& g, const T* const* obs, T* out, size_t npts, unsigned long long* first_bad,
                                hipStream_t stream) {
  Brick2Args<T> a;
  a.bricks = static_cast<const T*>(g.bricks);
  a.out = out;
  a.first_bad = first_bad;
  a.npts = npts;
  for (int d = 0; d < 2; ++d) {
    a.obs[d] = obs[d];
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
  }
  a.nbj = g.brick_nb[1];
  size_t lds = 0;
  a.ax.use_lds = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  if (g.kind == kRectilinear) lds = fill_axis_args<T, 2>(g, a.ax, /*big_lds=*/true, /*records=*/true);
  const int axr = lane_axes_mode(g);  // both axes <= 64 coordinates: searched across lanes, no LDS image
  if (axr) lds = 0;
  // Two points per lane when every stream is aligned to 2*sizeof(T) (the handle's `ppl` option = 1
  // forces the scalar form).
  bool aligned = (reinterpret_cast<uintptr_t>(out) % (2 * sizeof(T))) == 0;
  for (int d = 0; d < 2; ++d) aligned = aligned && (reinterpret_cast<uintptr_t>(obs[d]) % (2 * sizeof(T))) == 0;
  const int ppl = (aligned && g.cfg.ppl != 1) ? 2 : 1;
  const unsigned blocks = (g.kind == kRegular || axr) ? one_pass_blocks(npts, ppl * (axr ? 4 : 1)) : grid_blocks(npts, ppl, g.cfg);
#define GO2(RECT, FMA, AXR, PPL) do { g.tag.set("k_linear2_brick", {RECT, FMA, AXR, PPL}, 0b0011u); hipLaunchKernelGGL((k_linear2_brick<T, RECT, FMA, AXR, PPL>), dim3(blocks), dim3(kBlock), lds, stream, a); } while (0)
#define GO(RECT, FMA, AXR) do { if (ppl == 2) GO2(RECT, FMA, AXR, 2); else GO2(RECT, FMA, AXR, 1); } while (0)
  if (g.kind == kRegular) { if (g.fma) GO(false, true, 0); else GO(false, false, 0); }
  else if (axr == 2) { if (g.fma) GO(true, true, 2); else GO(true, false, 2); }
  else if (axr == 3) { if (g.fma) GO(true, true, 3); else GO(true, false, 3); }
  else if (axr == 1) { if (g.fma) GO(true, true, 1); else GO(true, false, 1); }
  else { if (g.fma) GO(true, true, 0); else GO(true, false, 0); }
#undef GO
#undef GO2
  return hipGetLastError();
}

template hipError_t launch_linear2_brick<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t);
template hipError_t launch_linear2_brick<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t);

}  // namespace interpn
