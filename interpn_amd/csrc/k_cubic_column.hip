// Launcher of the LDS-column multicubic kernel for sorted 4-D points (cubic_column.h).
#include "cubic_column.h"

namespace interpn {

namespace {
constexpr size_t kColumnLdsMax = 144 * 1024;  // of the CU's 160 KiB; the kernel's static words and the runtime keep the rest

template <typename T>
size_t column_bytes(const GridDesc& g) { return col_lds_bytes<T>((unsigned)g.n[2] * (unsigned)g.n[3]); }
}  // namespace

bool cubic_column_applies(const GridDesc& g) {
  if (g.method != kCubic || g.kind != kRegular || g.ndims != 4 || !g.bricks) return false;
  // the fully overlapped tile table is what the column is filled from
  const bool main11 = g.brick_step[0] == 1 && g.brick_step[1] == 1;
  if (!main11 && !g.bricks11) return false;
  if ((long long)(g.n[0] - 3) * (g.n[1] - 3) > kMaxBins) return false;  // one bin per (i, j) cell
  const size_t col = g.dtype == kF64 ? column_bytes<double>(g) : column_bytes<float>(g);
  return col <= kColumnLdsMax;
}

template <typename T>
hipError_t launch_cubic_column(const GridDesc& g, const BinPlan& plan, const T* const* sorted_obs, const unsigned* index,
                               const BinExtras& extras, bool unpermute, T* out, size_t npts, size_t max_parts,
                               unsigned long long* first_bad, size_t index_base, hipStream_t stream) {
  CubicColumnArgs<T> a;
  a.tiles = static_cast<const T*>(g.bricks);
  {
    unsigned nb[2];
    size_t bytes = 0;
    cubic_tile_geometry(g, 1, 1, nb, &bytes);
    a.table_bytes = (unsigned)bytes;
    a.nbj = nb[1];
    unsigned acc = nb[0] * nb[1] * 16u;
    a.plane_stride[0] = a.plane_stride[1] = 0;
    a.plane_stride[3] = acc;
    a.plane_stride[2] = acc * (unsigned)g.n[3];
  }
  for (int d = 0; d < 4; ++d) {
    a.obs[d] = sorted_obs[d];
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
  }
  a.index = index;
  a.out = out;
  a.res_sorted = unpermute ? static_cast<T*>(extras.res_sorted) : nullptr;
  a.first_bad = first_bad;
  a.index_base = index_base;
  a.npts = npts;
  a.bin_end = extras.bin_end;
  a.part_prefix = extras.part_prefix;
  a.nbins = plan.nbins;
  a.nb1 = plan.nb1;
  a.inv_mult = (unsigned)plan.inv_mult;
  a.part_points = 0;
  a.linearize = g.linearize;
  const size_t lds = column_bytes<T>(g);
  // More than 64 KiB of dynamic LDS needs the opt-in, once per kernel and device.
  auto prepare = [&](auto kernel) -> hipError_t {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColumnLdsMax);
  };
  hipError_t e;
  if (g.fma) {
    e = prepare(k_cubic_column<T, true>);
    if (e != hipSuccess) return e;
    g.tag.set("k_cubic_column", {1}, 0b1u);
    hipLaunchKernelGGL((k_cubic_column<T, true>), dim3((unsigned)max_parts), dim3(kColThreads), lds, stream, a);
  } else {
    e = prepare(k_cubic_column<T, false>);
    if (e != hipSuccess) return e;
    g.tag.set("k_cubic_column", {0}, 0b1u);
    hipLaunchKernelGGL((k_cubic_column<T, false>), dim3((unsigned)max_parts), dim3(kColThreads), lds, stream, a);
  }
  e = hipGetLastError();
  if (e != hipSuccess || !unpermute) return e;
  hipLaunchKernelGGL(k_unpermute<T>, dim3((unsigned)((npts + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                     static_cast<const T*>(extras.res_sorted), extras.rank, out, npts);
  return hipGetLastError();
}

template hipError_t launch_cubic_column<double>(const GridDesc&, const BinPlan&, const double* const*, const unsigned*, const BinExtras&,
                                                bool, double*, size_t, size_t, unsigned long long*, size_t, hipStream_t);
template hipError_t launch_cubic_column<float>(const GridDesc&, const BinPlan&, const float* const*, const unsigned*, const BinExtras&,
                                               bool, float*, size_t, size_t, unsigned long long*, size_t, hipStream_t);

}  // namespace interpn
