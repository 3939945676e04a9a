// Launcher of the LDS-column multicubic kernel for sorted 3-D points (cubic3_column.h).
#include "cubic3_column.h"

namespace interpn {

namespace {
constexpr size_t kCubic3ColumnLds = 60 * 1024;  // dynamic LDS of a workgroup at most (no opt-in): n2 tiles + the rectilinear axis image
// points of a part at most: 16 steps of the workgroup's 256 lanes (a bin of more is cut, k_bin_scan)
constexpr unsigned kCubic3PartPoints = 16u * kBlock;

size_t column3_bytes(const GridDesc& g) {
  const size_t elem = g.dtype == kF64 ? 8 : 4;
  return ((size_t)g.n[2] * (16 * elem + 16) + 15) / 16 * 16;
}
}  // namespace

unsigned cubic3_column_part_points() { return kCubic3PartPoints; }

// `g`: the handle's description with the fully overlapped tile table as `bricks` (brick_step 1,1).
bool cubic3_column_applies(const GridDesc& g) {
  if (g.method != kCubic || g.ndims != 3 || !g.bricks || g.brick_step[0] != 1 || g.brick_step[1] != 1) return false;
  if ((long long)(g.n[0] - 1) * (g.n[1] - 1) > kMaxBins) return false;
  size_t lds = column3_bytes(g);
  if (g.kind == kRectilinear) {
    if (!g.axis_image) return false;
    if (g.axis_image_bytes <= thresholds(g.cfg).axis_lds) lds += g.axis_image_bytes;
  }
  return lds <= kCubic3ColumnLds;
}

template <typename T>
hipError_t launch_cubic3_column(const GridDesc& g, const BinPlan& plan, const BinExtras& extras, const unsigned* index, T* out,
                                size_t npts, size_t max_parts, unsigned long long* first_bad, size_t index_base, hipStream_t stream) {
  if (!cubic3_column_applies(g) || npts == 0) return hipErrorInvalidValue;
  Cubic3ColumnArgs<T> a;
  a.tiles = static_cast<const T*>(g.bricks);
  unsigned nb[2];
  size_t tbytes = 0;
  cubic_tile_geometry(g, 1, 1, nb, &tbytes);
  if (tbytes >= 0xFFFFF000ull) return hipErrorInvalidValue;
  a.table_bytes = (unsigned)tbytes;
  a.records = static_cast<const T*>(extras.records);
  a.index = index;
  a.out = out;
  a.first_bad = first_bad;
  a.index_base = index_base;
  a.bin_end = extras.bin_end;
  a.part_prefix = extras.part_prefix;
  a.work = extras.work;
  a.nbins = plan.nbins;
  a.nb1 = plan.nb1;
  a.inv_mult = (unsigned)plan.inv_mult;
  for (int d = 0; d < 3; ++d) {
    a.start[d] = (T)g.start[d];
    a.step[d] = (T)g.step[d];
    a.n[d] = g.n[d];
  }
  a.linearize = g.linearize;
  a.plane_stride = nb[0] * nb[1] * 16u;
  a.nbj = nb[1];
  a.ax.use_lds = 0;
  a.ax.use_rec = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  size_t lds = column3_bytes(g);
  a.axes_lds_off = (unsigned)lds;
  if (g.kind == kRectilinear) {
    fill_axis_args<T, 3>(g, a.ax);  // offsets, lengths, bucket tables (coordinates + tables: what cubic_rect_locate searches)
    a.ax.use_rec = 0;
    a.ax.image = static_cast<const unsigned char*>(g.axis_image);
    a.ax.image_bytes = g.axis_image_bytes;
    a.ax.use_lds = g.axis_image_bytes <= thresholds(g.cfg).axis_lds ? 1 : 0;
    if (a.ax.use_lds) lds += g.axis_image_bytes;
  }
  // persistent: as many workgroups as stay resident (registers: four waves per SIMD at most), fewer when there are fewer parts
  size_t wgs = (size_t)(g.cfg.num_cus > 0 ? g.cfg.num_cus : 256) * 4;
  if (wgs > max_parts) wgs = max_parts;
  if (wgs < 1) wgs = 1;
#define GO3(RECT, FMA)                                                                                              \
  do {                                                                                                              \
    g.tag.set("k_cubic3_column", {RECT, FMA}, 0b11u);                                                               \
    hipLaunchKernelGGL((k_cubic3_column<T, RECT, FMA>), dim3((unsigned)wgs), dim3(kBlock), lds, stream, a);         \
  } while (0)
  if (g.kind == kRectilinear) { if (g.fma) GO3(true, true); else GO3(true, false); }
  else { if (g.fma) GO3(false, true); else GO3(false, false); }
#undef GO3
  return hipGetLastError();
}

template hipError_t launch_cubic3_column<double>(const GridDesc&, const BinPlan&, const BinExtras&, const unsigned*, double*, size_t,
                                                 size_t, unsigned long long*, size_t, hipStream_t);
template hipError_t launch_cubic3_column<float>(const GridDesc&, const BinPlan&, const BinExtras&, const unsigned*, float*, size_t,
                                                size_t, unsigned long long*, size_t, hipStream_t);

}  // namespace interpn
