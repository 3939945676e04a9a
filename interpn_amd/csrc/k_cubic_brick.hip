// Launchers of the tiled multicubic kernels (cubic_brick.h).
#include "cubic_brick.h"

namespace interpn {

void cubic_tile_geometry(const GridDesc& g, int si, int sj, unsigned nb[2], size_t* bytes) {
  nb[0] = (unsigned)((g.n[0] - 4) / si + 2);
  nb[1] = (unsigned)((g.n[1] - 4) / sj + 2);
  size_t planes = 1;
  for (int d = 2; d < g.ndims; ++d) planes *= (size_t)g.n[d];
  *bytes = planes * nb[0] * nb[1] * 16 * (g.dtype == kF64 ? 8 : 4);
}

hipError_t build_cubic_tiles(const GridDesc& g, void* tiles, hipStream_t stream) {
  size_t planes = 1;
  for (int d = 2; d < g.ndims; ++d) planes *= (size_t)g.n[d];
  const size_t elems = planes * g.brick_nb[0] * g.brick_nb[1] * 16;
  size_t blocks = (elems + kBlock - 1) / kBlock;
  if (blocks > 65535) blocks = 65535;
  if (g.dtype == kF64)
    hipLaunchKernelGGL(k_build_cubic_tiles<double>, dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                       static_cast<const double*>(g.vals), static_cast<double*>(tiles), planes, g.n[0], g.n[1],
                       g.brick_step[0], g.brick_step[1], g.brick_nb[0], g.brick_nb[1]);
  else
    hipLaunchKernelGGL(k_build_cubic_tiles<float>, dim3((unsigned)blocks), dim3(kBlock), 0, stream,
                       static_cast<const float*>(g.vals), static_cast<float*>(tiles), planes, g.n[0], g.n[1],
                       g.brick_step[0], g.brick_step[1], g.brick_nb[0], g.brick_nb[1]);
  return hipGetLastError();
}

template <typename T, int N, bool RECT, bool FMA>
static hipError_t launch_steps(const GridDesc& g, const CubicBrickArgs<T, N>& a, size_t lds, unsigned blocks, hipStream_t stream) {
  const int si = g.brick_step[0], sj = g.brick_step[1];
#define GO(SI, SJ) do { g.tag.set("k_cubic_brick", {N, RECT, FMA, SI, SJ}, 0b00110u); hipLaunchKernelGGL((k_cubic_brick<T, N, RECT, FMA, SI, SJ>), dim3(blocks), dim3(kBlock), lds, stream, a); } while (0)
  if (si == 4 && sj == 4) GO(4, 4);
  else if (si == 2 && sj == 4) GO(2, 4);
  else if (si == 2 && sj == 2) GO(2, 2);
  else if (si == 1 && sj == 4) GO(1, 4);
  else if (si == 1 && sj == 1) GO(1, 1);
  else return hipErrorInvalidValue;
#undef GO
  return hipGetLastError();
}

template <typename T, int N>
static hipError_t launch_n(const GridDesc& g, const T* const* obs, T* out, size_t npts, unsigned long long* first_bad,
                           hipStream_t stream, const unsigned* scatter, size_t index_base) {
  CubicBrickArgs<T, N> a;
  a.bricks = static_cast<const T*>(g.bricks);
  {
    unsigned nb[2];
    size_t bytes = 0;
    cubic_tile_geometry(g, g.brick_step[0], g.brick_step[1], nb, &bytes);
    a.table_bytes = (unsigned)bytes;  // < 4 GiB by construction (maybe_build_cubic_tiles)
  }
  a.out = out;
  a.first_bad = first_bad;
  a.npts = npts;
  a.scatter = scatter;
  a.index_base = index_base;
  a.linearize = g.linearize;
  for (int d = 0; d < N; ++d) {
    a.obs[d] = obs[d];
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
    a.plane_stride[d] = 0;
  }
  a.nbj = g.brick_nb[1];
  // table[plane index (dims 2..N-1, C order)][bi][bj][16]
  unsigned acc = g.brick_nb[0] * g.brick_nb[1] * 16u;
  for (int d = N - 1; d >= 2; --d) {
    a.plane_stride[d] = acc;
    acc *= (unsigned)g.n[d];
  }
  const bool dma = g.brick_step[0] == 1 && g.brick_step[1] == 1;  // cubic_brick.h::cubic_dma
  size_t lds = dma ? (size_t)(kBlock / 64) * cubic_dma_image<T>() : (size_t)kBlock * kCubRow * (sizeof(T) > 4 ? sizeof(T) : 4);
  a.ax.use_lds = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  if (g.kind == kRectilinear) lds += fill_axis_args<T, N>(g, a.ax);
  for (int d = 0; d < N; ++d)
    a.crec[d] = (g.kind == kRectilinear && g.axis_crec_bytes)
                    ? reinterpret_cast<const CubicCellRecord<T>*>(static_cast<const unsigned char*>(g.axis_image) + g.axis_crec_off[d]) : nullptr;
  unsigned blocks = grid_blocks(npts, 1, g.cfg);
  a.gate = scatter ? nullptr : g.launch_gate;
  a.eighth = 0;
  if (scatter && g.cfg.deal && blocks >= 64) {
    blocks &= ~7u;  // eight equal XCD shares; the grid-stride loop covers what the rounding drops
    a.eighth = ((npts + 7) / 8 + kBlock - 1) / kBlock * kBlock;
  }
  if (g.kind == kRegular)
    return g.fma ? launch_steps<T, N, false, true>(g, a, lds, blocks, stream)
                 : launch_steps<T, N, false, false>(g, a, lds, blocks, stream);
  return g.fma ? launch_steps<T, N, true, true>(g, a, lds, blocks, stream)
               : launch_steps<T, N, true, false>(g, a, lds, blocks, stream);
}

template <typename T>
hipError_t launch_cubic_brick(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                              unsigned long long* first_bad, hipStream_t stream, const unsigned* scatter, size_t index_base) {
  switch (g.ndims) {
    case 2: return launch_n<T, 2>(g, obs, out, npts, first_bad, stream, scatter, index_base);
    case 3: return launch_n<T, 3>(g, obs, out, npts, first_bad, stream, scatter, index_base);
    case 4: return launch_n<T, 4>(g, obs, out, npts, first_bad, stream, scatter, index_base);
    default: return hipErrorInvalidValue;
  }
}

template hipError_t launch_cubic_brick<double>(const GridDesc&, const double* const*, double*, size_t, unsigned long long*, hipStream_t,
                                               const unsigned*, size_t);
template hipError_t launch_cubic_brick<float>(const GridDesc&, const float* const*, float*, size_t, unsigned long long*, hipStream_t,
                                              const unsigned*, size_t);

}  // namespace interpn
