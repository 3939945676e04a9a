// Launcher of the LDS-column multicubic kernel for sorted 4-D points (cubic_column.h).
#include <mutex>

#include "cubic_column.h"

namespace interpn {

namespace {
// LDS of a CU, and what a workgroup of this kernel declares statically per wave group (histogram +
// control words, rounded up)
constexpr size_t kColumnStaticPerGroup = 4096 + 256;

// More than 64 KiB of dynamic LDS needs the opt-in, once per kernel and device: remembered here so
// that the launch path does not pay the call (microseconds) every time.
hipError_t column_lds_opt_in(const void* kernel, int groups, size_t cu_lds) {
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
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)(cu_lds - (size_t)groups * kColumnStaticPerGroup));
  if (e == hipSuccess && slot) {
    std::lock_guard<std::mutex> lk(mu);
    slot->devices |= 1ull << dev;
  }
  return e;
}
}  // namespace

// How the column evaluation runs on this grid (cubic_column.h): one persistent workgroup per CU
// of `groups` wave groups; each group keeps the local order of its part (16-bit, 32 points per
// thread at most) and a sub-column of cpp + 3 tile rows in its share of the CU's LDS.  false: not
// even one class per phase fits, or the local sort's keys do not.
static bool column_plan_at_pitch(const GridDesc& g, size_t pitch, ColumnPlan* plan);

// Tiles sit 16 bytes apart in LDS (consecutive tiles on different bank groups: cubic_column.h);
// where only the bare tile lets a part run in fewer phases — cfg4: the whole 32 x 32 column is
// 128 KiB bare and 144 KiB padded, beside 24 KiB of local order — the padding goes on rectilinear
// grids: a phase more costs a fill nothing overlaps, three group barriers and a second ragged end
// of the row loop; the bank conflicts of sorted points on bare tiles cost LDS cycles.
bool cubic_column_plan(const GridDesc& g, ColumnPlan* plan) {
  const size_t elem = g.dtype == kF64 ? 8 : 4;
  ColumnPlan padded, bare;
  const bool okp = column_plan_at_pitch(g, 16 * elem + 16, &padded);
  const bool okb = column_plan_at_pitch(g, 16 * elem, &bare);
  if (g.cfg.column_pad == 1 && okp) { *plan = padded; return true; }
  if (g.cfg.column_pad == 0 && okb) { *plan = bare; return true; }
  // measured (cfg4 shapes, both forms alternating in one process): regular 32^4 0.536 ms padded in
  // two phases against 0.596 bare in one; rectilinear 0.906 against 0.846 (its nodes cost three
  // times the arithmetic, the LDS conflicts hide behind it)
  if (okb && (!okp || (bare.nphase < padded.nphase && g.kind == kRectilinear))) { *plan = bare; return true; }
  if (okp) { *plan = padded; return true; }
  return false;
}

static bool column_plan_at_pitch(const GridDesc& g, size_t pitch, ColumnPlan* plan) {
  ColumnPlan p;
  p.pitch = (unsigned)pitch;
  const size_t elem = g.dtype == kF64 ? 8 : 4;
  // compiled shapes: 768 threads as two groups of six waves (the product shape) or as one group;
  // 384 threads as one group; 256 threads as two groups of two waves (tests)
  const int t = g.cfg.column_threads;
  p.threads = t <= 256 ? 256 : (t <= 384 ? 384 : 768);
  p.groups = p.threads == 384 ? 1 : (p.threads == 256 ? 2 : (g.cfg.column_groups == 1 ? 1 : 2));
  if (g.kind == kRectilinear) {  // compiled for the product shape and the small test shape only
    if (p.threads != 256) { p.threads = 768; p.groups = 1; }
  }
  const size_t cu_lds = thresholds(g.cfg).column_lds;  // the LDS of a CU: 160 KiB on MI355X
  if (cu_lds < (size_t)p.groups * (kColumnStaticPerGroup + 8192)) return false;
  // rectilinear: the axis image (coordinates + tables of the four axes) sits behind the groups' regions
  p.axes_bytes = g.kind == kRectilinear ? ((size_t)g.axis_image_bytes + 15) / 16 * 16 : 0;
  if (cu_lds < (size_t)p.groups * (kColumnStaticPerGroup + 8192) + p.axes_bytes) return false;
  const size_t per_group = ((cu_lds - p.axes_bytes - (size_t)p.groups * kColumnStaticPerGroup) / (size_t)p.groups) / 1024 * 1024;
  const int n2 = g.n[2], n3 = g.n[3];
  auto sub_bytes = [&](int rows) { return ((size_t)rows * (size_t)n3 * pitch + 1023) / 1024 * 1024; };
  const int ncls2 = n2 - 1;
  // the largest part whose local order leaves room for at least a useful sub-column: parts as large
  // as the registers of the local sort allow (fewer fills per point, longer phases), halved until
  // at least a quarter of the classes fit a phase (or down to 2048 points)
  unsigned part = (unsigned)(col_per_thread(p.threads / p.groups) * (p.threads / p.groups));
  if (part > kColumnMaxPart) part = kColumnMaxPart;
  // The local order of a part (16 bits per point) goes into the padding of the sub-column's tiles
  // first (eight entries per tile; none on bare tiles), what is left behind the sub-column.
  const size_t pad_entries_per_tile = pitch > 16 * elem ? 8 : 0;
  auto tail_bytes = [&](int rows, unsigned pts) {
    const size_t in_pad = (size_t)rows * (size_t)n3 * pad_entries_per_tile;
    return in_pad >= pts ? (size_t)0 : ((size_t)2 * (pts - in_pad) + 15) / 16 * 16;
  };
  int rows = 0;
  for (;; part /= 2) {
    rows = n2;
    while (rows >= 4 && sub_bytes(rows) + tail_bytes(rows, part) > per_group) --rows;
    if (rows < 4) rows = 0;
    const bool roomy = rows >= n2 || (rows >= 4 && (rows - 3) * 4 >= ncls2);
    if (roomy || part <= 2048) break;
  }
  if (rows < 4) return false;
  p.part_points = part;
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
  const long long kmax = column_keys_in_index(g) && ncls2 <= 256 ? 256 : kColKeys;  // 8-bit keys ride in the index words
  while ((long long)ncls2 * q3_of(p.sh3) > kmax && p.sh3 < 30) ++p.sh3;
  p.q3 = q3_of(p.sh3);
  if ((long long)ncls2 * p.q3 > kmax) return false;
  p.perm_pad = (unsigned)((size_t)sub_rows * (size_t)n3 * pad_entries_per_tile);
  p.group_bytes = p.sub_bytes + (unsigned)tail_bytes(sub_rows, part);
  p.lds_bytes = (size_t)p.group_bytes * (size_t)p.groups + p.axes_bytes;
  *plan = p;
  return true;
}

bool cubic_column_applies(const GridDesc& g) {
  if (g.method != kCubic || g.ndims != 4 || !g.bricks) return false;
  if (g.kind == kRectilinear) {  // exact classes need every axis' bucket table (strictly increasing, finite axes)
    for (int d = 0; d < 4; ++d)
      if (g.axis_buckets[d] <= 0) return false;
    if (!g.axis_image || g.axis_image_bytes > 32 * 1024) return false;
  }
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
  a.work = extras.work;
  a.bin_flags = g.linearize ? extras.bin_flags : nullptr;
  a.nbins = plan.nbins;
  a.nb1 = plan.nb1;
  a.inv_mult = (unsigned)plan.inv_mult;
  a.linearize = g.linearize;
  a.cpp = cp.cpp;
  a.nphase = cp.nphase;
  a.q3 = cp.q3;
  a.sh3 = cp.sh3;
  a.sub_bytes = cp.sub_bytes;
  a.group_bytes = cp.group_bytes;
  // time stamps (measurement aid): only into a buffer the caller has declared large enough for this launch's parts
  a.stamps = (g.cfg.debug_stamps && (unsigned long long)g.cfg.debug_stamps_bytes >= (unsigned long long)max_parts * 64ull)
                 ? reinterpret_cast<unsigned long long*>((uintptr_t)g.cfg.debug_stamps) : nullptr;
  a.ax.use_lds = 0;
  a.ax.use_rec = 0;
  a.ax.image = nullptr;
  a.ax.image_bytes = 0;
  a.axes_lds_off = (unsigned)((size_t)cp.group_bytes * (size_t)cp.groups);
  a.coef = g.cfg.column_coef;
  a.pitch = cp.pitch;
  a.index_keys = extras.key_q3 > 0 ? 1 : 0;
  a.perm_pad = cp.perm_pad;
#ifdef INTERPN_COLUMN_CREC_VARIANT
  for (int d = 0; d < 4; ++d) a.crec[d] = nullptr;
#endif
  if (g.kind == kRectilinear) {
    fill_axis_args<T, 4>(g, a.ax);  // offsets, lengths, bucket tables; the kernel stages the image itself
    a.ax.use_rec = 0;
    a.ax.image = static_cast<const unsigned char*>(g.axis_image);
    a.ax.image_bytes = g.axis_image_bytes;
  }
  const size_t lds = cp.lds_bytes;
  // persistent: one workgroup per CU (its LDS leaves room for no second one), fewer when there are fewer parts
  size_t wgs = (max_parts + (size_t)cp.groups - 1) / (size_t)cp.groups;
  if (wgs > (size_t)g.cfg.num_cus) wgs = (size_t)g.cfg.num_cus;
  if (wgs < 1) wgs = 1;
  auto prepare = [&](auto kernel) -> hipError_t {
    if (lds <= 64 * 1024) return hipSuccess;
    return column_lds_opt_in(reinterpret_cast<const void*>(kernel), cp.groups, thresholds(g.cfg).column_lds);
  };
  hipError_t e = hipSuccess;
#define GO_R(RECT, FMA, TH, GR)                                                                                  \
  do {                                                                                                           \
    e = prepare(k_cubic_column<T, RECT, FMA, TH, GR>);                                                           \
    if (e != hipSuccess) return e;                                                                               \
    g.tag.set("k_cubic_column", {RECT, FMA, TH, GR, 0}, 0b10011u);                                               \
    hipLaunchKernelGGL((k_cubic_column<T, RECT, FMA, TH, GR>), dim3((unsigned)wgs), dim3(TH), lds, stream, a);   \
  } while (0)
  // rectilinear grids: the product shape and the small test shape
#define GO(FMA, TH, GR)                                                                                          \
  do {                                                                                                           \
    if (g.kind == kRectilinear) { if (TH == 256) GO_R(true, FMA, 256, 2); else GO_R(true, FMA, 768, 1); }        \
    else GO_R(false, FMA, TH, GR);                                                                               \
  } while (0)
  // the measurement build (time stamps): the product shapes in f64 only
  if (a.stamps && sizeof(T) == 8 && g.fma && cp.threads == 768 && g.kind == kRegular) {
    if constexpr (sizeof(T) == 8) {
      if (cp.groups == 1) {
        e = prepare(k_cubic_column<T, false, true, 768, 1, true>);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_cubic_column<T, false, true, 768, 1, true>), dim3((unsigned)wgs), dim3(768), lds, stream, a);
      } else {
        e = prepare(k_cubic_column<T, false, true, 768, 2, true>);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_cubic_column<T, false, true, 768, 2, true>), dim3((unsigned)wgs), dim3(768), lds, stream, a);
      }
      g.tag.set("k_cubic_column", {0, 1, 768, cp.groups, 1}, 0b10011u);
      return hipGetLastError();
    }
  }
#define GO_T(FMA)                                                                                                \
  do {                                                                                                           \
    if (cp.threads == 256) GO(FMA, 256, 2); else if (cp.threads == 384) GO(FMA, 384, 1);                         \
    else if (cp.groups == 1) GO(FMA, 768, 1); else GO(FMA, 768, 2);                                              \
  } while (0)
  if (g.fma) GO_T(true); else GO_T(false);
#undef GO_T
#undef GO
#undef GO_R
  return hipGetLastError();
}

template hipError_t launch_cubic_column<double>(const GridDesc&, const BinPlan&, const BinExtras&, const unsigned*, double*, size_t, size_t,
                                                unsigned long long*, size_t, hipStream_t);
template hipError_t launch_cubic_column<float>(const GridDesc&, const BinPlan&, const BinExtras&, const unsigned*, float*, size_t, size_t,
                                               unsigned long long*, size_t, hipStream_t);

}  // namespace interpn
