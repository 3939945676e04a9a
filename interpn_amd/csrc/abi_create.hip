// Handle creation in the reference's validation order, replication onto another device,
// destruction, status strings.  (C ABI internals, see abi_internal.h.)
#include "abi_internal.h"
#include "cubic_cell_record.h"

using namespace interpn;
using namespace interpn_abi;

namespace interpn_abi {

int finish_create(interpn_hip_interp* h, const void* vals, size_t nvals, size_t elem, int vals_mem) {
  GridDesc& g = h->desc;
  {
    const DeviceProps dp = device_props(h->device);
    g.cfg.num_cus = dp.num_cus;
    g.cfg.num_xcds = dp.num_xcds;
    g.cfg.l2_bytes = dp.l2_bytes;
    g.cfg.lds_per_cu = dp.lds_per_cu;
    g.cfg.lds_per_wg = dp.lds_per_wg;
  }
  latch_env(g.cfg);
  g.nvals = nvals;
  if (vals_mem == INTERPN_HIP_MEM_DEVICE) {
    g.vals = vals;
  } else {
    HIP_TRY(pool_alloc(h->device, &h->vals_owned, nvals * elem));
    HIP_TRY(hipMemcpy(h->vals_owned, vals, nvals * elem, hipMemcpyHostToDevice));
    g.vals = h->vals_owned;
  }
  HIP_TRY(pool_alloc(h->device, (void**)&h->first_bad, sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(h->first_bad, 0xFF, sizeof(unsigned long long), nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  int st = maybe_build_bricks(h);
  if (st) return st;
  return INTERPN_HIP_OK;
}

template <typename T>
int create_regular(int method, const size_t* dims, size_t ndims, const T* starts, size_t nstarts, const T* steps,
                   size_t nsteps, const T* vals, size_t nvals, int vals_mem, int linearize, int device,
                   interpn_hip_interp** handle) {
  if (!handle) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  *handle = nullptr;
  const int flavour = method & (INTERPN_HIP_FLAVOUR_FMA | INTERPN_HIP_FLAVOUR_NO_FMA);
  if (flavour == (INTERPN_HIP_FLAVOUR_FMA | INTERPN_HIP_FLAVOUR_NO_FMA)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  method &= ~(INTERPN_HIP_FLAVOUR_FMA | INTERPN_HIP_FLAVOUR_NO_FMA);
  if (method != kLinear && method != kCubic && method != kNearest) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if (vals_mem != INTERPN_HIP_MEM_HOST && vals_mem != INTERPN_HIP_MEM_DEVICE) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  int st = validate_regular<T>(method, dims, ndims, starts, nstarts, steps, nsteps, nvals);
  if (st) return st;
  if (!vals && nvals) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  int dev;
  st = resolve_device(device, &dev);
  if (st) return st;
  DeviceGuard guard(dev);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  interpn_hip_interp* h = new (std::nothrow) interpn_hip_interp();
  if (!h) return INTERPN_HIP_ERR_OUT_OF_MEMORY;
  h->device = dev;
  GridDesc& g = h->desc;
  g.method = method;
  g.kind = kRegular;
  g.dtype = sizeof(T) == 8 ? kF64 : kF32;
  g.ndims = (int)ndims;
  g.linearize = linearize ? 1 : 0;
  g.fma = flavour == INTERPN_HIP_FLAVOUR_FMA ? 1 : (flavour == INTERPN_HIP_FLAVOUR_NO_FMA ? 0 : g_fma.load());
  for (size_t i = 0; i < ndims; ++i) {
    g.n[i] = (int)dims[i];
    g.start[i] = (double)starts[i];
    g.step[i] = (double)steps[i];
    {
      const T prod = steps[i] * (T)(dims[i] - 1);
      const T last = starts[i] + prod;  // regular.rs:164, not fused
      g.bound_lo[i] = (double)(T)__builtin_fmin((double)starts[i], (double)last);
      g.bound_hi[i] = (double)(T)__builtin_fmax((double)starts[i], (double)last);
    }
    g.grid_total += dims[i];
  }
  st = finish_create(h, vals, nvals, sizeof(T), vals_mem);
  if (st) {
    interpn_hip_destroy(h);
    return st;
  }
  *handle = h;
  return INTERPN_HIP_OK;
}

template <typename T>
int create_rectilinear(int method, const T* const* grids, const size_t* grid_lens, size_t ngrids, const T* vals,
                       size_t nvals, int vals_mem, int linearize, int device, interpn_hip_interp** handle) {
  if (!handle) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  *handle = nullptr;
  const int flavour = method & (INTERPN_HIP_FLAVOUR_FMA | INTERPN_HIP_FLAVOUR_NO_FMA);
  if (flavour == (INTERPN_HIP_FLAVOUR_FMA | INTERPN_HIP_FLAVOUR_NO_FMA)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  method &= ~(INTERPN_HIP_FLAVOUR_FMA | INTERPN_HIP_FLAVOUR_NO_FMA);
  if (method != kLinear && method != kCubic && method != kNearest) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if (vals_mem != INTERPN_HIP_MEM_HOST && vals_mem != INTERPN_HIP_MEM_DEVICE) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  int st = validate_rectilinear<T>(method, grids, grid_lens, ngrids, nvals);
  if (st) return st;
  if (!vals && nvals) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  int dev;
  st = resolve_device(device, &dev);
  if (st) return st;
  DeviceGuard guard(dev);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  interpn_hip_interp* h = new (std::nothrow) interpn_hip_interp();
  if (!h) return INTERPN_HIP_ERR_OUT_OF_MEMORY;
  h->device = dev;
  GridDesc& g = h->desc;
  g.method = method;
  g.kind = kRectilinear;
  g.dtype = sizeof(T) == 8 ? kF64 : kF32;
  g.ndims = (int)ngrids;
  g.linearize = linearize ? 1 : 0;
  g.fma = flavour == INTERPN_HIP_FLAVOUR_FMA ? 1 : (flavour == INTERPN_HIP_FLAVOUR_NO_FMA ? 0 : g_fma.load());
  size_t total = 0;
  for (size_t i = 0; i < ngrids; ++i) {
    g.n[i] = (int)grid_lens[i];
    total += grid_lens[i];
    g.bound_lo[i] = (double)grids[i][0];                  // rectilinear.rs:121-123
    g.bound_hi[i] = (double)grids[i][grid_lens[i] - 1];
  }
  g.grid_total = total;
  // Axis image: per axis the coordinates (16-byte aligned) followed by the bucket table
  // ((M+1) x u32, M = 2n) when the axis is strictly increasing and finite; otherwise M = 0 and the
  // kernels bisect with the reference's probe sequence (its `new` only checks g[1] > g[0],
  // multilinear/rectilinear.rs:195, so unsorted axes are legal input).
  size_t bytes = 0;
  for (size_t i = 0; i < ngrids; ++i) {
    const size_t n = grid_lens[i];
    bool sorted = true;
    for (size_t k = 0; k + 1 < n && sorted; ++k) sorted = grids[i][k + 1] > grids[i][k];
    sorted = sorted && std::isfinite((double)grids[i][0]) && std::isfinite((double)grids[i][n - 1]);
    const double span = (double)grids[i][n - 1] - (double)grids[i][0];
    int M = 0;
    if (sorted && span > 0 && std::isfinite(span) && n <= ((size_t)1 << 28)) M = (int)(2 * n);
    g.axis_buckets[i] = M;
    g.axis_g0[i] = (double)grids[i][0];
    g.axis_scale[i] = M ? (double)(T)((double)M / span) : 0.0;
    if (M && !(g.axis_scale[i] > 0 && std::isfinite(g.axis_scale[i]))) { g.axis_buckets[i] = 0; M = 0; }
    g.axis_g_off[i] = (unsigned)bytes;
    bytes += (n * sizeof(T) + 15) & ~(size_t)15;
    g.axis_tab_off[i] = (unsigned)bytes;
    bytes += (((size_t)M + 1) * sizeof(unsigned) + 15) & ~(size_t)15;
    g.axis_ltab_off[i] = 0;
    g.axis_lscale[i] = 0.0;
    if (M && n <= 64) {
      const double ls = (double)(T)(255.0 / span);
      if (ls > 0 && std::isfinite(ls)) {
        g.axis_lscale[i] = ls;
        g.axis_ltab_off[i] = (unsigned)bytes;
        bytes += 65 * sizeof(unsigned) + 12;  // 64 packed words + the scan length, 16-byte multiple
      }
    }
    if (bytes > 0xFFFFFF00ull) {
      interpn_hip_destroy(h);
      return INTERPN_HIP_ERR_UNSUPPORTED;
    }
  }
  // Per-bucket search records (interpn_host.h: axis_rec_*; multilinear only — the nearest kernel
  // measured slower with them and never reads them): built on the host with the arithmetic
  // bucket_of() uses on the device (same type, same operations, no contraction), for axes whose
  // buckets hold at most one coordinate each.  All axes or none.  They sit BEHIND the image
  // (coordinates + tables), which keeps its own size: a kernel that searches without records
  // stages the image alone.  The form is chosen for the LDS budget of the kernel that will run
  // (rect_args.h): full records {g[k-1], g[k], g[k+1], k} when they fit it, else the compact
  // form {g[k], k} + a copy of the coordinates, else none.
  g.axis_image_bytes = (unsigned)bytes;
  std::vector<std::vector<unsigned char>> recs(ngrids);
  {
    bool all = method == kLinear;
    const DeviceProps dp0 = device_props(dev);
    LaunchConfig c0;
    c0.lds_per_cu = dp0.lds_per_cu;
    c0.lds_per_wg = dp0.lds_per_wg;
    const Thresholds th0 = thresholds(c0);
    const size_t cap = ngrids <= 2 ? th0.axis_lds_wide : th0.axis_lds;
    std::vector<std::vector<int>> firsts(ngrids);
    size_t full_bytes = 0, compact_bytes = 0;
    const size_t rsize = sizeof(T) == 8 ? 32 : 16, csize = sizeof(T) == 8 ? 16 : 8;
    for (size_t i = 0; i < ngrids && all; ++i) {
      const int n = (int)grid_lens[i], M = g.axis_buckets[i];
      if (M <= 0) { all = false; break; }
      const T g0 = (T)g.axis_g0[i], scale = (T)g.axis_scale[i];
      std::vector<int>& first = firsts[i];
      first.assign(M + 1, n);  // first[b] = tab[b]: coordinates in buckets < b
      int prev = -1;
      for (int k = 0; k < n && all; ++k) {
        const T u = (grids[i][k] - g0) * scale;
        const int b = u >= (T)(M - 1) ? (M - 1) : (u > (T)0 ? (int)u : 0);
        if (b <= prev) all = false;  // two coordinates in one bucket (or not monotone): no records
        for (int q = prev + 1; q <= b; ++q) first[q] = k;
        prev = b;
      }
      for (int b = 0; b < M && all; ++b)
        if (first[b] >= n) all = false;  // cannot happen: g[n-1] lies in bucket M-1
      full_bytes += (size_t)M * rsize;
      compact_bytes += (((size_t)n * sizeof(T) + 15) & ~(size_t)15) + (size_t)M * csize;
    }
    // INTERPN_HIP_AXIS_REC_FORM = 1 / 2 forces the full / compact form where it fits (tuning, tests)
    const char* form_env = getenv("INTERPN_HIP_AXIS_REC_FORM");
    const int form = form_env ? atoi(form_env) : 0;
    // N <= 2: beyond 32 KiB the full records cost a resident workgroup per CU more than their
    // single access saves (2-D 384^2: 1.05 ms with 49 KiB of full records, see profiles/r04_rect_bucket_records.txt)
    const size_t full_cap = ngrids <= 2 ? (size_t)32 * 1024 : cap;
    const bool full = all && (form == 1 ? full_bytes <= cap : (form == 2 ? false : full_bytes <= full_cap));
    const bool compact = all && !full && compact_bytes <= cap;
    if ((full || compact) && bytes + (full ? full_bytes : compact_bytes) < 0xFFFFFF00ull) {
      g.axis_rec_base = (unsigned)bytes;
      g.axis_rec_compact = compact ? 1 : 0;
      for (size_t i = 0; i < ngrids; ++i) {
        const int n = (int)grid_lens[i], M = g.axis_buckets[i];
        const std::vector<int>& first = firsts[i];
        if (full) {
          recs[i].assign((size_t)M * rsize, 0);
          for (int b = 0; b < M; ++b) {
            const int k = first[b];
            T triple[3] = {k > 0 ? grids[i][k - 1] : (T)0, grids[i][k], k + 1 < n ? grids[i][k + 1] : (T)0};
            unsigned char* r = recs[i].data() + (size_t)b * rsize;
            memcpy(r, triple, 3 * sizeof(T));
            const unsigned ku = (unsigned)k;
            memcpy(r + 3 * sizeof(T), &ku, sizeof(ku));
          }
          g.axis_rec_off[i] = (unsigned)bytes;
        } else {
          const size_t gbytes = ((size_t)n * sizeof(T) + 15) & ~(size_t)15;
          recs[i].assign(gbytes + (size_t)M * csize, 0);
          memcpy(recs[i].data(), grids[i], (size_t)n * sizeof(T));
          for (int b = 0; b < M; ++b) {
            const int k = first[b];
            unsigned char* r = recs[i].data() + gbytes + (size_t)b * csize;
            memcpy(r, &grids[i][k], sizeof(T));
            const unsigned ku = (unsigned)k;
            memcpy(r + sizeof(T), &ku, sizeof(ku));
          }
          g.axis_recg_off[i] = (unsigned)bytes;
          g.axis_rec_off[i] = (unsigned)(bytes + gbytes);
        }
        bytes += recs[i].size();
      }
      g.axis_rec_bytes = (unsigned)(bytes - g.axis_rec_base);
    } else {
      for (auto& r : recs) r.clear();
    }
  }
  // Per-cell records of a multicubic handle (interpn_host.h: axis_crec_*) while they stay L1-sized: the kernels read them
  // per point through the vector caches (profiles/r05_rect_cubic.txt: 3-D 64^3 f64, 18 KiB: 0.83 -> 0.74 ms per 1e7 points,
  // 128^3, 36 KiB: 0.84 -> 0.80; 2-D 512^2, 96 KiB: 0.30 -> 0.34, slower than the divisions they replace).
  // INTERPN_HIP_CUBIC_RECORDS=n: that bound in KiB (default 40; 0: never).
  std::vector<std::vector<CubicCellRecord<T>>> crecs(ngrids);
  g.axis_crec_bytes = 0;
  {
    const char* env = getenv("INTERPN_HIP_CUBIC_RECORDS");
    const long cap_kib = env ? atol(env) : 40;
    bool build = method == kCubic && cap_kib > 0;
    size_t total_bytes = 0;
    for (size_t i = 0; i < ngrids && build; ++i) {
      if (grid_lens[i] < 4) build = false;
      total_bytes += (grid_lens[i] - 1) * sizeof(CubicCellRecord<T>);
    }
    if (build && total_bytes > (size_t)cap_kib * 1024) build = false;
    if (build && bytes + total_bytes < 0xFFFFFF00ull) {
      for (size_t i = 0; i < ngrids; ++i) {
        build_cubic_cell_records<T>(grids[i], (int)grid_lens[i], crecs[i]);
        g.axis_crec_off[i] = (unsigned)bytes;
        bytes += crecs[i].size() * sizeof(CubicCellRecord<T>);
      }
      g.axis_crec_bytes = (unsigned)total_bytes;
    }
  }
  g.axis_alloc_bytes = (unsigned)bytes;
  hipError_t e = pool_alloc(h->device, &h->grids_owned, bytes);
  if (e == hipSuccess) e = hipMemsetAsync(h->grids_owned, 0, bytes, nullptr);
  if (e != hipSuccess) {
    interpn_hip_destroy(h);
    return hip_fail(e);
  }
  g.axis_image = h->grids_owned;
  for (size_t i = 0; i < ngrids; ++i) {
    char* gdev = (char*)h->grids_owned + g.axis_g_off[i];
    e = hipMemcpy(gdev, grids[i], grid_lens[i] * sizeof(T), hipMemcpyHostToDevice);
    if (e == hipSuccess && g.axis_buckets[i])
      e = build_buckets<T>(reinterpret_cast<const T*>(gdev), g.n[i], g.axis_buckets[i], (T)g.axis_g0[i],
                           (T)g.axis_scale[i], reinterpret_cast<unsigned*>((char*)h->grids_owned + g.axis_tab_off[i]),
                           nullptr);
    if (e == hipSuccess && g.axis_ltab_off[i])
      e = build_lane_table<T>(reinterpret_cast<const T*>(gdev), g.n[i], (T)g.axis_g0[i], (T)g.axis_lscale[i],
                              reinterpret_cast<unsigned*>((char*)h->grids_owned + g.axis_ltab_off[i]), nullptr);
    if (e == hipSuccess && g.axis_rec_bytes)
      e = hipMemcpy((char*)h->grids_owned + (g.axis_rec_compact ? g.axis_recg_off[i] : g.axis_rec_off[i]), recs[i].data(), recs[i].size(),
                    hipMemcpyHostToDevice);
    if (e == hipSuccess && g.axis_crec_bytes)
      e = hipMemcpy((char*)h->grids_owned + g.axis_crec_off[i], crecs[i].data(), crecs[i].size() * sizeof(CubicCellRecord<T>), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      interpn_hip_destroy(h);
      return hip_fail(e);
    }
    g.grid[i] = gdev;
  }
  e = hipStreamSynchronize(nullptr);
  // The largest bucket population of each lane table decides which form of the cross-lane search
  // the kernels run (lane_axes.h): 4 bytes per axis, once per handle.
  for (size_t i = 0; i < ngrids && e == hipSuccess; ++i) {
    if (!g.axis_ltab_off[i]) continue;
    unsigned pop = 0;
    e = hipMemcpy(&pop, (const char*)h->grids_owned + g.axis_ltab_off[i] + 64 * sizeof(unsigned), sizeof(unsigned),
                  hipMemcpyDeviceToHost);
    g.axis_lscan[i] = (int)pop;
  }
  if (e != hipSuccess) {
    interpn_hip_destroy(h);
    return hip_fail(e);
  }
  st = finish_create(h, vals, nvals, sizeof(T), vals_mem);
  if (st) {
    interpn_hip_destroy(h);
    return st;
  }
  *handle = h;
  return INTERPN_HIP_OK;
}


template int create_regular<double>(int, const size_t*, size_t, const double*, size_t, const double*, size_t, const double*, size_t, int, int, int, interpn_hip_interp**);
template int create_regular<float>(int, const size_t*, size_t, const float*, size_t, const float*, size_t, const float*, size_t, int, int, int, interpn_hip_interp**);
template int create_rectilinear<double>(int, const double* const*, const size_t*, size_t, const double*, size_t, int, int, int, interpn_hip_interp**);
template int create_rectilinear<float>(int, const float* const*, const size_t*, size_t, const float*, size_t, int, int, int, interpn_hip_interp**);

}  // namespace interpn_abi

// ===========================================================================
extern "C" {

const char* interpn_hip_strerror(int status) {
  switch (status) {
    case INTERPN_HIP_OK: return "";
    case INTERPN_HIP_ERR_DIM_MISMATCH: return "Dimension mismatch";
    case INTERPN_HIP_ERR_MIN_TWO_ENTRIES: return "All grids must have at least two entries";
    case INTERPN_HIP_ERR_MIN_2_ENTRIES: return "All grids must have at least 2 entries";
    case INTERPN_HIP_ERR_MIN_FOUR_ENTRIES: return "All grids must have at least four entries";
    case INTERPN_HIP_ERR_MIN_4_ENTRIES: return "All grids must have at least 4 entries";
    case INTERPN_HIP_ERR_NOT_MONOTONIC: return "All grids must be monotonically increasing";
    case INTERPN_HIP_ERR_UNREPRESENTABLE: return "Unrepresentable coordinate value";
    case INTERPN_HIP_ERR_TOO_MANY_DIMS:
      return "Dimension exceeds maximum (8). Use interpolator struct directly for higher dimensions.";
    case INTERPN_HIP_ERR_TOO_MANY_DIMS_6: return "Dimension exceeds maximum (6).";
    case INTERPN_HIP_ERR_REFERENCE_PANIC: return "the reference implementation panics on this input (slice length mismatch or integer overflow)";
    case INTERPN_HIP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case INTERPN_HIP_ERR_UNSUPPORTED: return "grid axis too long for the device kernels";
    case INTERPN_HIP_ERR_NO_DEVICE: return "no usable HIP device";
    case INTERPN_HIP_ERR_OUT_OF_MEMORY: return "out of device or pinned host memory";
    case INTERPN_HIP_ERR_HIP: return "HIP runtime error";
    default: return "unknown status";
  }
}

const char* interpn_hip_last_hip_error(void) { return t_last_hip_error.c_str(); }
const char* interpn_hip_version(void) { return "0.1.0"; }
int interpn_hip_set_fma(int enabled) { return g_fma.exchange(enabled ? 1 : 0); }

int interpn_hip_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

int interpn_hip_trim(int device, size_t* freed_bytes) {
  if (freed_bytes) *freed_bytes = 0;
  int dev;
  const int st = resolve_device(device, &dev);
  if (st) return st;
  DeviceGuard guard(dev);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  const size_t freed = pool_trim(dev);
  if (freed_bytes) *freed_bytes = freed;
  return INTERPN_HIP_OK;
}

#define DEFINE_CREATE(T, SUFFIX)                                                                              \
  int interpn_hip_create_regular_##SUFFIX(int method, const size_t* dims, size_t ndims, const T* starts,     \
                                          size_t nstarts, const T* steps, size_t nsteps, const T* vals,      \
                                          size_t nvals, int vals_mem, int linearize_extrapolation,           \
                                          int device, interpn_hip_interp** handle) {                         \
    return create_regular<T>(method, dims, ndims, starts, nstarts, steps, nsteps, vals, nvals, vals_mem,     \
                             linearize_extrapolation, device, handle);                                       \
  }                                                                                                           \
  int interpn_hip_create_rectilinear_##SUFFIX(int method, const T* const* grids, const size_t* grid_lens,    \
                                              size_t ngrids, const T* vals, size_t nvals, int vals_mem,      \
                                              int linearize_extrapolation, int device,                       \
                                              interpn_hip_interp** handle) {                                 \
    return create_rectilinear<T>(method, grids, grid_lens, ngrids, vals, nvals, vals_mem,                    \
                                 linearize_extrapolation, device, handle);                                   \
  }
DEFINE_CREATE(double, f64)
DEFINE_CREATE(float, f32)

// Clone an interpolator onto another device of this process.  The grid (`vals`, and the axis image
// of a rectilinear grid: coordinates + search tables) is copied DEVICE TO DEVICE with
// hipMemcpyPeer — between two GPUs of one node that is an xGMI transfer, no host staging and no
// second H2D upload —, the re-laid table is rebuilt on the target device.  This is the
// single-process counterpart of the one RCCL broadcast the multi-process path does
// (interpn_amd/sharded.py): SURVEY.md section 8(e) "grid replicated read-only on every GPU".
int interpn_hip_replicate(const interpn_hip_interp* src, int device, interpn_hip_interp** out) {
  if (!src || !out) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int dev;
  int st = resolve_device(device, &dev);
  if (st) return st;
  DeviceGuard guard(dev);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  interpn_hip_interp* h = new (std::nothrow) interpn_hip_interp();
  if (!h) return INTERPN_HIP_ERR_OUT_OF_MEMORY;
  h->device = dev;
  GridDesc& g = h->desc;
  g = src->desc;  // scalars, axis-image offsets, options; every device pointer is replaced below
  g.vals = nullptr;
  g.bricks = nullptr;
  g.bricks11 = nullptr;
  g.sweep_bricks = nullptr;
  g.brick_cell = 0;
  g.rec1_buckets = 0;
  g.axis_image = nullptr;
  g.tag = KernelTag();
  for (int d = 0; d < 8; ++d) g.grid[d] = nullptr;
  const size_t elem = g.dtype == kF64 ? 8 : 4;
  hipError_t e = pool_alloc(dev, &h->vals_owned, g.nvals * elem);
  if (e == hipSuccess) e = hipMemcpyPeer(h->vals_owned, dev, src->desc.vals, src->device, g.nvals * elem);
  if (e == hipSuccess && src->desc.axis_image && g.axis_alloc_bytes) {
    e = pool_alloc(dev, &h->grids_owned, g.axis_alloc_bytes);
    if (e == hipSuccess) e = hipMemcpyPeer(h->grids_owned, dev, src->desc.axis_image, src->device, g.axis_alloc_bytes);
    if (e == hipSuccess) {
      g.axis_image = h->grids_owned;
      for (int d = 0; d < g.ndims; ++d) g.grid[d] = (const char*)h->grids_owned + g.axis_g_off[d];
    }
  }
  if (e != hipSuccess) {
    interpn_hip_destroy(h);
    return hip_fail(e);
  }
  // finish_create with a device `vals` borrows the pointer; here the clone owns it (vals_owned).
  st = finish_create(h, h->vals_owned, g.nvals, elem, INTERPN_HIP_MEM_DEVICE);
  if (st) {
    interpn_hip_destroy(h);
    return st;
  }
  *out = h;
  return INTERPN_HIP_OK;
}

int interpn_hip_elem_size(const interpn_hip_interp* h) { return h ? (h->desc.dtype == kF64 ? 8 : 4) : 0; }
int interpn_hip_ndims(const interpn_hip_interp* h) { return h ? h->desc.ndims : 0; }
int interpn_hip_device(const interpn_hip_interp* h) { return h ? h->device : -1; }

void interpn_hip_destroy(interpn_hip_interp* h) {
  if (!h) return;
  DeviceGuard guard(h->device);
  // Blocks go back to the pool only once nothing in flight can still touch them.  Wait for the
  // work THIS handle enqueued — the event behind the last launch on every caller stream it was
  // given, and its own lane streams — not for the whole device: unrelated streams (a training
  // step on the same GPU) keep running.  Launches that could not be marked fall back to
  // hipDeviceSynchronize.
  {
    std::lock_guard<std::mutex> lk(h->marks_mu);
    for (auto& m : h->marks) {
      if (hipEventSynchronize(m.event) != hipSuccess) { (void)hipGetLastError(); h->sync_device_at_destroy = true; }
      (void)hipEventDestroy(m.event);
    }
    h->marks.clear();
  }
  for (auto& l : h->lane)
    if (l.stream && hipStreamSynchronize(l.stream) != hipSuccess) { (void)hipGetLastError(); h->sync_device_at_destroy = true; }
  if (h->sync_device_at_destroy) (void)hipDeviceSynchronize();
  for (auto& l : h->lane) {
    if (l.stream) pool_return_kit(h->device, l.stream, l.flag_host);
    pool_free(h->device, l.flag_dev);
    pool_free(h->device, l.obs);
    pool_free(h->device, l.out);
  }
  pool_return_small(h->device, h->small_host);
  for (auto& sl : h->bin_slots) {  // their streams were waited for above (marks)
    if (sl.event) (void)hipEventDestroy(sl.event);
    for (hipEvent_t e : sl.stage)
      if (e) (void)hipEventDestroy(e);
    pool_free(h->device, sl.scratch);
  }
  pool_free(h->device, h->first_bad);
  pool_return_pinned_word(h->device, h->finish_word);
  pool_return_pinned_word(h->device, h->probe_host);
  pool_free(h->device, h->grids_owned);
  pool_free(h->device, h->bricks_owned);
  pool_free(h->device, h->bricks11_owned);
  pool_free(h->device, h->sweep_owned);
  pool_free(h->device, h->vals_owned);
  delete h;
}

}  // extern "C"
