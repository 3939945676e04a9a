// Launcher of the LDS-column multicubic kernel for sorted 4-D points (cubic_column.h).
#include <mutex>

#include "cubic_column.h"

namespace interpn {

namespace {
// LDS of a CU and what a workgroup of this kernel declares statically (s_hist + a few words, rounded up)
constexpr size_t kCuLdsBytes = 160 * 1024;
constexpr size_t kColumnStaticLds = 4608;

int column_threads_of(const GridDesc& g) {
  const int t = g.cfg.column_threads;
  return t <= 256 ? 256 : (t <= 384 ? 384 : (t <= 512 ? 512 : 768));
}

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
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kCuLdsBytes - kColumnStaticLds));
  if (e == hipSuccess && slot) {
    std::lock_guard<std::mutex> lk(mu);
    slot->devices |= 1ull << dev;
  }
  return e;
}
}  // namespace

// How the column evaluation cuts this grid's (k, l) column into K-range phases (cubic_column.h):
// `column_wgs` workgroups share a CU's LDS; each keeps the local order of its part (16-bit, 16
// points per thread) and a sub-column of cpp + 3 tile rows.  false: not even one class per phase
// fits, or the local sort's keys do not.
bool cubic_column_plan(const GridDesc& g, ColumnPlan* plan) {
  ColumnPlan p;
  p.threads = column_threads_of(g);
  p.part_points = (unsigned)(kColPerThread * p.threads);
  if (p.part_points > kColumnMaxPart) p.part_points = kColumnMaxPart;
  const size_t elem = g.dtype == kF64 ? 8 : 4;
  const size_t tile = 16 * elem;
  const int wgs = g.cfg.column_wgs >= 1 && g.cfg.column_wgs <= 4 ? g.cfg.column_wgs : 2;
  const size_t per_wg = (kCuLdsBytes / (size_t)wgs) / 256 * 256;
  const size_t perm = 2 * (size_t)p.part_points;
  if (per_wg < kColumnStaticLds + perm + 4 * tile) return false;
  const size_t room = per_wg - kColumnStaticLds - perm;
  const int n2 = g.n[2], n3 = g.n[3];
  auto sub_bytes = [&](int rows) { return (size_t)(((size_t)rows * (size_t)n3 + 15) / 16) * 16 * tile; };
  int rows = n2;  // the whole column if it fits
  while (rows > 4 && sub_bytes(rows) > room) --rows;
  if (sub_bytes(rows) > room) return false;
  const int ncls2 = n2 - 1;
  int cpp = rows >= n2 ? ncls2 : rows - 3;  // a phase of cpp classes spans cpp + 3 rows at most
  if (cpp < 1) return false;
  if (g.cfg.column_cpp > 0 && g.cfg.column_cpp < cpp) cpp = g.cfg.column_cpp;
  int nphase = (ncls2 + cpp - 1) / cpp;
  cpp = (ncls2 + nphase - 1) / nphase;      // even phases
  nphase = (ncls2 + cpp - 1) / cpp;
  p.cpp = cpp;
  p.nphase = nphase;
  const int sub_rows = nphase == 1 ? n2 : (cpp + 3 < n2 ? cpp + 3 : n2);
  p.sub_bytes = (unsigned)sub_bytes(sub_rows);
  // local sort keys: class of dim 2 x (class of dim 3 >> sh3), at most 1024
  p.sh3 = 0;
  auto q3_of = [&](int sh) { return ((n3 - 2) >> sh) + 1; };
  while ((long long)ncls2 * q3_of(p.sh3) > 1024 && p.sh3 < 30) ++p.sh3;
  p.q3 = q3_of(p.sh3);
  if ((long long)ncls2 * p.q3 > 1024) return false;
  p.lds_bytes = (size_t)p.sub_bytes + perm;
  *plan = p;
  return true;
}

bool cubic_column_applies(const GridDesc& g) {
  if (g.method != kCubic || g.kind != kRegular || g.ndims != 4 || !g.bricks) return false;
  // the fully overlapped tile table is what the column is filled from
  const bool main11 = g.brick_step[0] == 1 && g.brick_step[1] == 1;
  if (!main11 && !g.bricks11) return false;
  if ((long long)(g.n[0] - 1) * (g.n[1] - 1) > kMaxBins) return false;  // one bin per class pair of dims 0, 1
  ColumnPlan plan;
  return cubic_column_plan(g, &plan);
}

template <typename T>
hipError_t launch_cubic_column(const GridDesc& g, const BinPlan& plan, const BinExtras& extras, const unsigned* index,
                               T* out, size_t npts, size_t max_parts, unsigned long long* first_bad, size_t index_base,
                               hipStream_t stream) {
  ColumnPlan cp;
  if (!cubic_column_plan(g, &cp)) return hipErrorInvalidValue;
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
  a.cpp = cp.cpp;
  a.nphase = cp.nphase;
  a.q3 = cp.q3;
  a.sh3 = cp.sh3;
  a.sub_bytes = cp.sub_bytes;
  a.stamps = reinterpret_cast<unsigned long long*>((uintptr_t)g.cfg.debug_stamps);
  const size_t lds = cp.lds_bytes;
  auto prepare = [&](auto kernel) -> hipError_t {
    if (lds <= 64 * 1024) return hipSuccess;
    return column_lds_opt_in(reinterpret_cast<const void*>(kernel));
  };
  hipError_t e = hipSuccess;
#define GO(FMA, TH)                                                                                              \
  do {                                                                                                           \
    e = prepare(k_cubic_column<T, FMA, TH>);                                                                     \
    if (e != hipSuccess) return e;                                                                               \
    g.tag.set("k_cubic_column", {FMA, TH}, 0b01u);                                                               \
    hipLaunchKernelGGL((k_cubic_column<T, FMA, TH>), dim3((unsigned)max_parts), dim3(TH), lds, stream, a);       \
  } while (0)
#define GO_T(FMA)                                                                                                \
  do {                                                                                                           \
    if (cp.threads == 256) GO(FMA, 256); else if (cp.threads == 384) GO(FMA, 384);                               \
    else if (cp.threads == 512) GO(FMA, 512); else GO(FMA, 768);                                                 \
  } while (0)
  if (g.fma) GO_T(true); else GO_T(false);
#undef GO_T
#undef GO
  return hipGetLastError();
}

template hipError_t launch_cubic_column<double>(const GridDesc&, const BinPlan&, const BinExtras&, const unsigned*, double*, size_t, size_t,
                                                unsigned long long*, size_t, hipStream_t);
template hipError_t launch_cubic_column<float>(const GridDesc&, const BinPlan&, const BinExtras&, const unsigned*, float*, size_t, size_t,
                                               unsigned long long*, size_t, hipStream_t);

}  // namespace interpn
