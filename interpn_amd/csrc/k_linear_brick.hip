// Host side of the bricked multilinear path (N = 3..6): table geometry, the table builder's
// launcher and the kernel launchers.  The device code lives in linear_brick.h.
#include <cstdlib>

#include "linear_brick.h"

namespace interpn {

// Bricks needed per dimension (cell indices run 0 .. n-2):
//   step 1:      brick = cell index                      -> n-1 bricks
//   step 2:      brick = i>>1, +1 when i is odd (i+1 spills into the next brick) -> (n-1)/2 + 1
//   step KW-1 (k): the pair never leaves its brick row   -> (n-2)/(KW-1) + 1
static unsigned bricks_along(int n, int step) {
  if (step == 1) return (unsigned)(n - 1);
  if (step == 2) return (unsigned)((n - 1) / 2 + 1);
  return (unsigned)((n - 2) / step + 1);
}

void brick_geometry(const GridDesc& g, int si, int sj, unsigned nb[3], size_t* bytes) {
  const int N = g.ndims;
  const int kw = g.dtype == kF64 ? 4 : 8;
  nb[0] = bricks_along(g.n[N - 3], si);
  nb[1] = bricks_along(g.n[N - 2], sj);
  nb[2] = bricks_along(g.n[N - 1], kw - 1);
  size_t lead = 1;
  for (int d = 0; d < N - 3; ++d) lead *= (size_t)g.n[d];
  *bytes = lead * nb[0] * nb[1] * nb[2] * 128;
}

// 4-D cell bricks (linear_brick.h, CELL == 1): nb = bricks along (i, j, k, h), all steps 1 except
// KW-1 along k.
void brick_cell_geometry(const GridDesc& g, unsigned nb[4], size_t* bytes) {
  const int N = g.ndims;
  const int kw = g.dtype == kF64 ? 2 : 4;
  nb[0] = bricks_along(g.n[N - 3], 1);
  nb[1] = bricks_along(g.n[N - 2], 1);
  nb[2] = kw == 2 ? bricks_along(g.n[N - 1], 1) : bricks_along(g.n[N - 1], kw - 1);
  nb[3] = bricks_along(g.n[N - 4], 1);
  size_t lead = 1;
  for (int d = 0; d < N - 4; ++d) lead *= (size_t)g.n[d];
  *bytes = lead * nb[3] * nb[0] * nb[1] * nb[2] * 128;
}

// f32 2 x 4 x 4 bricks (linear_brick.h, CELL == 2): nb = bricks along (i, j, k), steps (1, 3, 3).
void brick_j4_geometry(const GridDesc& g, unsigned nb[3], size_t* bytes) {
  const int N = g.ndims;
  nb[0] = bricks_along(g.n[N - 3], 1);
  nb[1] = bricks_along(g.n[N - 2], 3);
  nb[2] = bricks_along(g.n[N - 1], 3);
  size_t lead = 1;
  for (int d = 0; d < N - 3; ++d) lead *= (size_t)g.n[d];
  *bytes = lead * nb[0] * nb[1] * nb[2] * 128;
}

static hipError_t build_cell_bricks(const GridDesc& g, void* bricks, hipStream_t stream) {
  const int N = g.ndims;
  size_t lead = 1;
  for (int d = 0; d < N - 4; ++d) lead *= (size_t)g.n[d];
  const size_t elems = lead * g.brick_nb[3] * g.brick_nb[0] * g.brick_nb[1] * g.brick_nb[2] * (g.dtype == kF64 ? 16 : 32);
  size_t blocks = (elems + kBlock - 1) / kBlock;
  if (blocks > 65535) blocks = 65535;
  if (g.dtype == kF64)
    hipLaunchKernelGGL(k_build_cell_bricks<double>, dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                       static_cast<const double*>(g.vals), static_cast<double*>(bricks), lead, g.n[N - 4], g.n[N - 3],
                       g.n[N - 2], g.n[N - 1], g.brick_nb[3], g.brick_nb[0], g.brick_nb[1], g.brick_nb[2]);
  else
    hipLaunchKernelGGL(k_build_cell_bricks<float>, dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                       static_cast<const float*>(g.vals), static_cast<float*>(bricks), lead, g.n[N - 4], g.n[N - 3],
                       g.n[N - 2], g.n[N - 1], g.brick_nb[3], g.brick_nb[0], g.brick_nb[1], g.brick_nb[2]);
  return hipGetLastError();
}

hipError_t build_bricks(const GridDesc& g, void* bricks, hipStream_t stream) {
  if (g.brick_cell == 1) return build_cell_bricks(g, bricks, stream);
  const int N = g.ndims;
  if (g.brick_cell == 2) {
    if (g.dtype != kF32) return hipErrorInvalidValue;
    size_t lead4 = 1;
    for (int d = 0; d < N - 3; ++d) lead4 *= (size_t)g.n[d];
    const size_t elems4 = lead4 * g.brick_nb[0] * g.brick_nb[1] * g.brick_nb[2] * 32;
    size_t blocks4 = (elems4 + kBlock - 1) / kBlock;
    if (blocks4 > 65535) blocks4 = 65535;
    hipLaunchKernelGGL(k_build_j4_bricks<float>, dim3((unsigned)blocks4), dim3(kBlock), 0, stream,
                       static_cast<const float*>(g.vals), static_cast<float*>(bricks), lead4, g.n[N - 3], g.n[N - 2],
                       g.n[N - 1], g.brick_nb[0], g.brick_nb[1], g.brick_nb[2]);
    return hipGetLastError();
  }
  size_t lead = 1;
  for (int d = 0; d < N - 3; ++d) lead *= (size_t)g.n[d];
  const size_t elems = lead * g.brick_nb[0] * g.brick_nb[1] * g.brick_nb[2] * (g.dtype == kF64 ? 16 : 32);
  size_t blocks = (elems + kBlock - 1) / kBlock;
  if (blocks > 65535) blocks = 65535;
  if (g.dtype == kF64)
    hipLaunchKernelGGL(k_build_bricks<double>, dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                       static_cast<const double*>(g.vals), static_cast<double*>(bricks), lead, g.n[N - 3], g.n[N - 2],
                       g.n[N - 1], g.brick_step[0], g.brick_step[1], g.brick_nb[0], g.brick_nb[1], g.brick_nb[2]);
  else
    hipLaunchKernelGGL(k_build_bricks<float>, dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                       static_cast<const float*>(g.vals), static_cast<float*>(bricks), lead, g.n[N - 3], g.n[N - 2],
                       g.n[N - 1], g.brick_step[0], g.brick_step[1], g.brick_nb[0], g.brick_nb[1], g.brick_nb[2]);
  return hipGetLastError();
}

template <typename T, int N, bool RECT, bool FMA, int PPL, int AXR>
static hipError_t launch_steps(const GridDesc& g, const BrickArgs<T, N>& a, size_t lds, unsigned blocks, hipStream_t stream) {
  const int si = g.brick_step[0], sj = g.brick_step[1];
  if constexpr (sizeof(T) == 4) {
    if (g.brick_cell == 2) {
      g.tag.set("k_linear_brick", {N, RECT, FMA, 1, 1, PPL, AXR, 0, 2}, 0b000000110u);
      hipLaunchKernelGGL((k_linear_brick<T, N, RECT, FMA, 1, 1, PPL, AXR, 0, 2>), dim3(blocks), dim3(kBlock), lds, stream, a);
      return hipGetLastError();
    }
  }
  if constexpr (N >= 4) {
    if (g.brick_cell == 1) {
      g.tag.set("k_linear_brick", {N, RECT, FMA, 1, 1, PPL, AXR, 0, 1}, 0b000000110u);
      hipLaunchKernelGGL((k_linear_brick<T, N, RECT, FMA, 1, 1, PPL, AXR, 0, 1>), dim3(blocks), dim3(kBlock), lds, stream, a);
      return hipGetLastError();
    }
  }
  g.tag.set("k_linear_brick", {N, RECT, FMA, si == 1 ? 1 : 2, (si == 1 && sj == 1) ? 1 : 2, PPL, AXR, 0, 0}, 0b000000110u);
  if (si == 1 && sj == 1) hipLaunchKernelGGL((k_linear_brick<T, N, RECT, FMA, 1, 1, PPL, AXR>), dim3(blocks), dim3(kBlock), lds, stream, a);
  else if (si == 1 && sj == 2) hipLaunchKernelGGL((k_linear_brick<T, N, RECT, FMA, 1, 2, PPL, AXR>), dim3(blocks), dim3(kBlock), lds, stream, a);
  else hipLaunchKernelGGL((k_linear_brick<T, N, RECT, FMA, 2, 2, PPL, AXR>), dim3(blocks), dim3(kBlock), lds, stream, a);
  return hipGetLastError();
}

template <typename T, int N, int PPL>
static hipError_t launch_kind(const GridDesc& g, BrickArgs<T, N>& a, size_t lds, size_t axis_lds, size_t npts, hipStream_t stream) {
  const int axr = lane_axes_mode(g);  // axes in lanes (lane_axes.h) or 0 = LDS / L2 search
  a.iters = brick_iters(g, npts, PPL, /*setup=*/g.kind != kRectilinear ? 0 : (axr == 0 ? 2 : 1));
  // a gated launch mostly returns at once (unordered points: the sweep kernel has the batch): few, fat workgroups
  if ((a.gate || g.launch_fat) && g.cfg.gated_iters > 0 && a.iters < (unsigned)g.cfg.gated_iters) a.iters = (unsigned)g.cfg.gated_iters;
  const size_t nslots = (npts + PPL - 1) / PPL;
  const size_t per_block = (size_t)kBlock * a.iters;
  const unsigned blocks = (unsigned)((nslots + per_block - 1) / per_block);
  if (g.kind == kRegular)
    return g.fma ? launch_steps<T, N, false, true, PPL, 0>(g, a, lds, blocks, stream)
                 : launch_steps<T, N, false, false, PPL, 0>(g, a, lds, blocks, stream);
  if (axr == 2)
    return g.fma ? launch_steps<T, N, true, true, PPL, 2>(g, a, lds, blocks, stream)
                 : launch_steps<T, N, true, false, PPL, 2>(g, a, lds, blocks, stream);
  if (axr == 3)
    return g.fma ? launch_steps<T, N, true, true, PPL, 3>(g, a, lds, blocks, stream)
                 : launch_steps<T, N, true, false, PPL, 3>(g, a, lds, blocks, stream);
  if (axr == 1)
    return g.fma ? launch_steps<T, N, true, true, PPL, 1>(g, a, lds, blocks, stream)
                 : launch_steps<T, N, true, false, PPL, 1>(g, a, lds, blocks, stream);
  return g.fma ? launch_steps<T, N, true, true, PPL, 0>(g, a, lds + axis_lds, blocks, stream)
               : launch_steps<T, N, true, false, PPL, 0>(g, a, lds + axis_lds, blocks, stream);
}

template <typename T, int N>
static hipError_t launch_n(const GridDesc& g, const T* const* obs, T* out, size_t npts, unsigned long long* first_bad,
                           hipStream_t stream) {
  typedef typename LeafVec<T, 2>::type P;
  BrickArgs<T, N> a;
  a.gate = g.launch_gate;
  a.bricks = static_cast<const T*>(g.bricks);
  a.out = out;
  a.first_bad = first_bad;
  a.npts = npts;
  for (int d = 0; d < N; ++d) {
    a.obs[d] = obs[d];
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
  }
  a.nbj = g.brick_nb[1];
  a.nbk = g.brick_nb[2];
  // Table offset per unit of each leading index.  3-D bricks: every leading dimension indexes
  // whole tables of the last three.  4-D cell bricks: dimension N-4 (h) indexes bricks (one brick
  // row per h cell), the dimensions in front of it whole 4-D tables.
  unsigned acc = g.brick_nb[0] * g.brick_nb[1] * g.brick_nb[2] * (g.dtype == kF64 ? 16u : 32u);
  a.lead_stride[0] = 0;
  for (int d = N - 4; d >= 0; --d) {
    a.lead_stride[d] = acc;
    acc *= (g.brick_cell == 1 && d == N - 4) ? g.brick_nb[3] : (unsigned)g.n[d];
  }
  size_t lds = (size_t)kBlock * kPieceRow * sizeof(P) + (size_t)kBlock * 16;
  a.ax.use_lds = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  size_t axis_lds = 0;
  if (g.kind == kRectilinear) axis_lds = fill_axis_args<T, N>(g, a.ax, false, /*records=*/true);
  // Two points per lane (vector coordinate/result accesses) for the 3-D shape when every stream is
  // aligned to 2*sizeof(T); the handle's `ppl` option = 1 forces the scalar form (tuning / testing).
  if constexpr (N == 3) {
    auto aligned_to = [&](size_t bytes) {
      bool al = (reinterpret_cast<uintptr_t>(out) % bytes) == 0;
      for (int d = 0; d < N; ++d) al = al && (reinterpret_cast<uintptr_t>(obs[d]) % bytes) == 0;
      return al;
    };
    // (f32 with FOUR points per lane — the 16-byte stream accesses f64 gets with two — was built and
    // measured in round 3: 64^3 0.666 against 0.661 ms, 72^3 0.790 against 0.769: no gain, not instantiated)
    if (aligned_to(2 * sizeof(T)) && g.cfg.ppl != 1) return launch_kind<T, N, 2>(g, a, lds, axis_lds, npts, stream);
  }
  return launch_kind<T, N, 1>(g, a, lds, axis_lds, npts, stream);
}

template <typename T>
hipError_t launch_linear_brick(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                               unsigned long long* first_bad, hipStream_t stream) {
  switch (g.ndims) {
    case 3: return launch_n<T, 3>(g, obs, out, npts, first_bad, stream);
    case 4: return launch_n<T, 4>(g, obs, out, npts, first_bad, stream);
    case 5: return launch_n<T, 5>(g, obs, out, npts, first_bad, stream);
    case 6: return launch_n<T, 6>(g, obs, out, npts, first_bad, stream);
    default: return hipErrorInvalidValue;
  }
}

template hipError_t launch_linear_brick<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t);
template hipError_t launch_linear_brick<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t);

}  // namespace interpn
