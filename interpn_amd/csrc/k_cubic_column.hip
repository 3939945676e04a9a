// Launcher of the LDS-column multicubic kernel for sorted 4-D points (cubic_column.h).
#include <mutex>

#include "cubic_column.h"

namespace interpn {

namespace {
constexpr size_t kColumnBytesMax = 128 * 1024;                 // the column itself
constexpr size_t kColumnPermBytes = 2 * (size_t)kColumnMaxPart;  // local order of a part (16-bit)
constexpr size_t kColumnLdsMax = kColumnBytesMax + kColumnPermBytes;  // dynamic LDS at most (+ ~4.3 KiB static: 148.3 of the CU's 160 KiB)

template <typename T>
size_t column_bytes(const GridDesc& g) { return col_lds_bytes<T>((unsigned)g.n[2] * (unsigned)g.n[3]); }

// More than 64 KiB of dynamic LDS needs the opt-in, once per kernel and device: remembered here so
// that the launch path does not pay the call (microseconds) every time.
hipError_t column_lds_opt_in(const void* kernel) {
  struct Seen { const void* fn; unsigned long long devices; };
  static std::mutex mu;
  static Seen seen[32] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = -1;
  Seen* slot = nullptr;
  if (dev >= 0) {
    std::lock_guard<std::mutex> lk(mu);
    for (Seen& s : seen) {
      if (s.fn == kernel) { slot = &s; break; }
      if (!s.fn) { s.fn = kernel; slot = &s; break; }
    }
    if (slot && ((slot->devices >> dev) & 1ull)) return hipSuccess;
  }
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColumnLdsMax);
  if (e == hipSuccess && slot) {
    std::lock_guard<std::mutex> lk(mu);
    slot->devices |= 1ull << dev;
  }
  return e;
}
}  // namespace

bool cubic_column_applies(const GridDesc& g) {
  if (g.method != kCubic || g.kind != kRegular || g.ndims != 4 || !g.bricks) return false;
  // the fully overlapped tile table is what the column is filled from
  const bool main11 = g.brick_step[0] == 1 && g.brick_step[1] == 1;
  if (!main11 && !g.bricks11) return false;
  if ((long long)(g.n[0] - 1) * (g.n[1] - 1) > kMaxBins) return false;  // one bin per class pair of dims 0, 1
  if ((long long)(g.n[2] - 1) * (g.n[3] - 1) > 1024) return false;      // the workgroup's local sort: class pairs of dims 2, 3
  const size_t col = g.dtype == kF64 ? column_bytes<double>(g) : column_bytes<float>(g);
  return col <= kColumnBytesMax;
}

template <typename T>
hipError_t launch_cubic_column(const GridDesc& g, const BinPlan& plan, const BinExtras& extras, const unsigned* index,
                               T* out, size_t npts, size_t max_parts, unsigned long long* first_bad, size_t index_base,
                               hipStream_t stream) {
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
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.rstep[d] = (T)(1.0 / g.step[d]);
    a.n[d] = g.n[d];
  }
  a.records = static_cast<const T*>(extras.records);
  a.index = index;
  a.out = out;
  a.first_bad = first_bad;
  a.index_base = index_base;
  a.npts = npts;
  a.bin_end = extras.bin_end;
  a.part_prefix = extras.part_prefix;
  a.nbins = plan.nbins;
  a.nb1 = plan.nb1;
  a.inv_mult = (unsigned)plan.inv_mult;
  a.linearize = g.linearize;
  const size_t lds = column_bytes<T>(g) + kColumnPermBytes;
  auto prepare = [&](auto kernel) -> hipError_t { return column_lds_opt_in(reinterpret_cast<const void*>(kernel)); };
  hipError_t e = hipSuccess;
  const int threads = g.cfg.column_threads == 1024 ? 1024 : (g.cfg.column_threads == 768 ? 768 : 512);
#define GO(FMA, TH)                                                                                              \
  do {                                                                                                           \
    e = prepare(k_cubic_column<T, FMA, TH>);                                                                     \
    if (e != hipSuccess) return e;                                                                               \
    g.tag.set("k_cubic_column", {FMA, TH}, 0b01u);                                                               \
    hipLaunchKernelGGL((k_cubic_column<T, FMA, TH>), dim3((unsigned)max_parts), dim3(TH), lds, stream, a);       \
  } while (0)
  if (g.fma) { if (threads == 1024) GO(true, 1024); else if (threads == 768) GO(true, 768); else GO(true, 512); }
  else { if (threads == 1024) GO(false, 1024); else if (threads == 768) GO(false, 768); else GO(false, 512); }
#undef GO
  return hipGetLastError();
}

template hipError_t launch_cubic_column<double>(const GridDesc&, const BinPlan&, const BinExtras&, const unsigned*, double*, size_t, size_t,
                                                unsigned long long*, size_t, hipStream_t);
template hipError_t launch_cubic_column<float>(const GridDesc&, const BinPlan&, const BinExtras&, const unsigned*, float*, size_t, size_t,
                                               unsigned long long*, size_t, hipStream_t);

}  // namespace interpn
