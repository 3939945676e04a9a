// C ABI of libinterpn_hip.so (see include/interpn_hip.h).  Host logic only: argument
// validation in the reference's order, device residency of the grid, the chunked host->device
// pipeline of the host-pointer entry points, and status reporting.  No CPU evaluation path
// exists in this library: if the HIP runtime or a device is missing, calls fail with
// INTERPN_HIP_ERR_NO_DEVICE / INTERPN_HIP_ERR_HIP.
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/interpn_hip.h"
#include "interpn_host.h"

using namespace interpn;

namespace {

std::atomic<int> g_fma{1};
thread_local std::string t_last_hip_error;

int hip_fail(hipError_t e) {
  t_last_hip_error = hipGetErrorString(e);
  if (e == hipErrorOutOfMemory) return INTERPN_HIP_ERR_OUT_OF_MEMORY;
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return INTERPN_HIP_ERR_NO_DEVICE;
  return INTERPN_HIP_ERR_HIP;
}

#define HIP_TRY(expr)                        \
  do {                                       \
    hipError_t _e = (expr);                  \
    if (_e != hipSuccess) return hip_fail(_e); \
  } while (0)

// Axis length limits of the device kernels (cell indices are 32-bit; f32 classifies cells by
// comparing against (float)n, exact only up to 2^24).
template <typename T> constexpr size_t max_axis_len() { return sizeof(T) == 8 ? (size_t)2147483391u : (size_t)16777216u; }

bool checked_product(const size_t* dims, size_t n, size_t* out) {
  size_t acc = 1;
  for (size_t i = 0; i < n; ++i)
    if (__builtin_mul_overflow(acc, dims[i], &acc)) return false;  // Cargo.toml:45 overflow-checks => panic
  *out = acc;
  return true;
}

class DeviceGuard {
 public:
  explicit DeviceGuard(int device) : prev_(-1), ok_(true) {
    if (hipGetDevice(&prev_) != hipSuccess) { ok_ = false; return; }
    if (device >= 0 && device != prev_) {
      if (hipSetDevice(device) != hipSuccess) ok_ = false;
      changed_ = true;
    }
  }
  ~DeviceGuard() {
    if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_);
  }
  bool ok() const { return ok_; }

 private:
  int prev_;
  bool ok_;
  bool changed_ = false;
};

// ---------------------------------------------------------------------------
// Device-memory pool.  The one-shot entry points rebuild the interpolator on every call, as the
// reference does (multilinear/regular.rs:65-71); hipMalloc/hipFree, stream and pinned-memory
// creation would dominate small calls (measured 600 us per call against 50 us with a resident
// handle), so freed blocks, streams and pinned status words are kept per device and reused.
// Blocks are returned only after the device has drained (interpn_hip_destroy synchronises, as
// hipFree would).  INTERPN_HIP_POOL_MB caps the cached bytes per device (default 1024, 0 = off).
constexpr int kMaxPoolDevices = 64;

size_t pool_size_class(size_t bytes) {
  if (bytes < 256) return 256;
  const int top = 63 - __builtin_clzll((unsigned long long)bytes);
  const size_t quantum = (size_t)1 << (top > 3 ? top - 3 : 0);  // 8 classes per octave: <= 12.5 % slack
  return (bytes + quantum - 1) / quantum * quantum;
}

struct DevPool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;                     // block -> class size
  std::unordered_map<size_t, std::vector<void*>> free_blocks;  // class size -> cached blocks
  size_t cached_bytes = 0;
  struct Kit { hipStream_t stream; unsigned long long* flag_host; };
  std::vector<Kit> kits;
  std::vector<unsigned long long*> pinned_words;
  std::vector<void*> small_buffers;  // pinned, device-mapped staging of the small-batch path (kSmallBytes each)
};

// Devices beyond the table are not pooled at all (plain hipMalloc / hipFree): two devices must never
// share cached blocks.
bool pooled_device(int device) { return device >= 0 && device < kMaxPoolDevices; }

DevPool& dev_pool(int device) {
  static DevPool pools[kMaxPoolDevices];
  return pools[pooled_device(device) ? device : 0];
}

size_t pool_cap_bytes() {
  static const size_t cap = [] {
    const char* env = getenv("INTERPN_HIP_POOL_MB");
    const long long mb = env ? atoll(env) : 1024;
    return (size_t)(mb < 0 ? 0 : mb) << 20;
  }();
  return cap;
}

// The current device must be `device`.
hipError_t pool_alloc(int device, void** out, size_t bytes) {
  if (!pooled_device(device)) return hipMalloc(out, bytes < 256 ? 256 : bytes);
  DevPool& pool = dev_pool(device);
  const size_t cls = pool_size_class(bytes);
  {
    std::lock_guard<std::mutex> lk(pool.mu);
    auto it = pool.free_blocks.find(cls);
    if (it != pool.free_blocks.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      pool.cached_bytes -= cls;
      pool.live[*out] = cls;
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, cls);
  if (e != hipSuccess) {
    // out of memory: drop everything cached and retry once
    (void)hipGetLastError();
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> lk(pool.mu);
      for (auto& kv : pool.free_blocks) {
        for (void* b : kv.second) drop.push_back(b);
        kv.second.clear();
      }
      pool.cached_bytes = 0;
    }
    for (void* b : drop) (void)hipFree(b);
    e = hipMalloc(out, cls);
    if (e != hipSuccess) return e;
  }
  std::lock_guard<std::mutex> lk(pool.mu);
  pool.live[*out] = cls;
  return hipSuccess;
}

// Only for blocks no in-flight work still touches.
void pool_free(int device, void* p) {
  if (!p) return;
  if (!pooled_device(device)) { (void)hipFree(p); return; }
  DevPool& pool = dev_pool(device);
  size_t cls = 0;
  {
    std::lock_guard<std::mutex> lk(pool.mu);
    auto it = pool.live.find(p);
    if (it != pool.live.end()) {
      cls = it->second;
      pool.live.erase(it);
      if (pool.cached_bytes + cls <= pool_cap_bytes()) {
        pool.free_blocks[cls].push_back(p);
        pool.cached_bytes += cls;
        return;
      }
    }
  }
  (void)hipFree(p);
}

hipError_t pool_take_kit(int device, hipStream_t* stream, unsigned long long** flag_host) {
  DevPool& pool = dev_pool(device);
  if (pooled_device(device)) {
    std::lock_guard<std::mutex> lk(pool.mu);
    if (!pool.kits.empty()) {
      *stream = pool.kits.back().stream;
      *flag_host = pool.kits.back().flag_host;
      pool.kits.pop_back();
      return hipSuccess;
    }
  }
  hipError_t e = hipHostMalloc((void**)flag_host, sizeof(unsigned long long), hipHostMallocDefault);
  if (e != hipSuccess) return e;
  e = hipStreamCreateWithFlags(stream, hipStreamNonBlocking);
  if (e != hipSuccess) { (void)hipHostFree(*flag_host); *flag_host = nullptr; }
  return e;
}

void pool_return_kit(int device, hipStream_t stream, unsigned long long* flag_host) {
  DevPool& pool = dev_pool(device);
  if (pooled_device(device)) {
    std::lock_guard<std::mutex> lk(pool.mu);
    if (pool.kits.size() < 16 && pool_cap_bytes() > 0) {
      pool.kits.push_back({stream, flag_host});
      return;
    }
  }
  (void)hipStreamDestroy(stream);
  (void)hipHostFree(flag_host);
}

hipError_t pool_take_pinned_word(int device, unsigned long long** word) {
  DevPool& pool = dev_pool(device);
  if (pooled_device(device)) {
    std::lock_guard<std::mutex> lk(pool.mu);
    if (!pool.pinned_words.empty()) {
      *word = pool.pinned_words.back();
      pool.pinned_words.pop_back();
      return hipSuccess;
    }
  }
  return hipHostMalloc((void**)word, sizeof(unsigned long long), hipHostMallocDefault);
}

void pool_return_pinned_word(int device, unsigned long long* word) {
  if (!word) return;
  DevPool& pool = dev_pool(device);
  if (pooled_device(device)) {
    std::lock_guard<std::mutex> lk(pool.mu);
    if (pool.pinned_words.size() < 64 && pool_cap_bytes() > 0) {
      pool.pinned_words.push_back(word);
      return;
    }
  }
  (void)hipHostFree(word);
}

// Staging of the small-batch host path: pinned host memory the kernel reads and writes directly
// over PCIe (zero-copy).  One fixed size serves every interpolator: 8 coordinate arrays + 1 result
// array of kSmallPoints f64 elements.
constexpr size_t kSmallPoints = 8192;
constexpr size_t kSmallBytes = (size_t)(8 + 1) * kSmallPoints * 8;

hipError_t pool_take_small(int device, void** buf) {
  DevPool& pool = dev_pool(device);
  if (pooled_device(device)) {
    std::lock_guard<std::mutex> lk(pool.mu);
    if (!pool.small_buffers.empty()) {
      *buf = pool.small_buffers.back();
      pool.small_buffers.pop_back();
      return hipSuccess;
    }
  }
  return hipHostMalloc(buf, kSmallBytes, hipHostMallocMapped);
}

void pool_return_small(int device, void* buf) {
  if (!buf) return;
  DevPool& pool = dev_pool(device);
  if (pooled_device(device)) {
    std::lock_guard<std::mutex> lk(pool.mu);
    if (pool.small_buffers.size() < 16 && pool_cap_bytes() > 0) {
      pool.small_buffers.push_back(buf);
      return;
    }
  }
  (void)hipHostFree(buf);
}

int device_num_cus(int device) {
  static std::atomic<int> cached[kMaxPoolDevices];
  if (device >= 0 && device < kMaxPoolDevices && cached[device].load() > 0) return cached[device].load();
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    n = 256;
  }
  if (device >= 0 && device < kMaxPoolDevices) cached[device].store(n);
  return n;
}

}  // namespace


struct interpn_hip_interp {
  GridDesc desc;
  int device = 0;
  void* vals_owned = nullptr;   // device copy of vals when created from host memory
  void* grids_owned = nullptr;  // one device allocation holding all rectilinear axes
  void* bricks_owned = nullptr; // bricked copy of vals (3-D multilinear f64)
  void* bricks11_owned = nullptr;  // 4-D multicubic: fully overlapped tiles for binned evaluation when `bricks` is another layout
  unsigned long long* first_bad = nullptr;  // device word, ~0 = no failure
  unsigned long long* finish_word = nullptr;  // pinned landing word of interpn_hip_finish
  std::mutex finish_mu;
  std::mutex host_mu;  // host-pointer evaluations on one handle share its two lanes: serialised
  // Caller streams that device-pointer work was enqueued on, each with an event recorded behind
  // the most recent such launch: interpn_hip_destroy waits for exactly these (and its own lane
  // streams) instead of stalling the whole device.  `sync_device_at_destroy` is set for launches
  // that could not be marked (a stream under capture, more than kMaxMarks streams, event failure).
  static constexpr size_t kMaxMarks = 16;
  struct StreamMark { hipStream_t stream; hipEvent_t event; };
  std::mutex marks_mu;
  std::vector<StreamMark> marks;
  bool sync_device_at_destroy = false;
  // Host-evaluation workspace (lazily allocated, reused across calls): two pipeline lanes so
  // that the upload of one chunk overlaps the download of the previous one.
  struct HostLane {
    size_t points = 0;
    void* obs = nullptr;                       // ndims * points elements (device)
    void* out = nullptr;                       // points elements (device)
    unsigned long long* flag_dev = nullptr;    // first failing index of the chunk in flight
    unsigned long long* flag_host = nullptr;   // pinned
    hipStream_t stream = nullptr;
  } lane[2];
  // Small batches skip the staging copies altogether (eval_host_impl): pinned host buffer that
  // the kernel reads the coordinates from and writes the results to, and its device address.
  void* small_host = nullptr;
  void* small_dev = nullptr;
  // Binned evaluation (interpn_host.h): scratch for one slice of sorted points, shared by every
  // evaluation through this handle.  `bin_event` is recorded behind the last use; the next user
  // makes its stream wait for it, so two streams never work in the scratch at the same time.
  // Up to kMaxBinSlots blocks, one per stream that evaluates concurrently: an evaluation takes
  // the block its stream used last (stream order alone makes the reuse safe), else an idle one,
  // else makes a new one (unless told not to allocate), else waits — on the device, through the
  // block's event — for the least recently used one.  `bin_mu` guards the slot list and the
  // per-handle report fields (desc.last_binned, desc.tag of a binned launch); the launches
  // themselves are enqueued outside it.
  static constexpr size_t kMaxBinSlots = 4;
  struct BinSlot {
    void* scratch = nullptr;
    size_t bytes = 0;
    hipEvent_t event = nullptr;
    bool recorded = false;       // `event` has been recorded at least once
    bool busy = false;           // a host thread is enqueueing into this block right now
    hipStream_t last_stream = nullptr;
    unsigned long long stamp = 0;  // use counter value of the last use (LRU)
    bool totals_clean = false;     // the block's bin counters are zero (left so by the last complete sort's scan)
    hipEvent_t stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // option stage_timing: start | hist | scan | scatter | kernel
    bool staged = false;           // the last use recorded them (single slice)
  };
  std::mutex bin_mu;
  std::vector<BinSlot> bin_slots;  // capacity kMaxBinSlots from the start: evaluations hold pointers to elements outside the lock
  unsigned long long bin_uses = 0;
  interpn_hip_interp() { bin_slots.reserve(kMaxBinSlots); }
  std::atomic<long long> evals_binned{0}, evals_in_place{0}, scratch_allocs{0};
};

namespace {

// ---------------------------------------------------------------------------
// Validation, in the order of the reference's `interpn` + `new` (+ `interp`).
// `nobs`/`obs_lens` may be absent (handle creation): pass check_obs = false.
template <typename T>
int validate_regular(int method, const size_t* dims, size_t ndims, const T* starts, size_t nstarts,
                     const T* steps, size_t nsteps, size_t nvals) {
  if (method == kNearest) {
    if (nstarts != ndims || nsteps != ndims) return INTERPN_HIP_ERR_DIM_MISMATCH;  // nearest/regular.rs:50
    if (ndims < 1 || ndims > 6) return INTERPN_HIP_ERR_TOO_MANY_DIMS_6;            // nearest/regular.rs:97
  } else if (method == kLinear) {
    // multilinear/regular.rs:60 — obs.len() is checked by the caller of this helper
    if (nstarts != ndims || nsteps != ndims) return INTERPN_HIP_ERR_DIM_MISMATCH;
    if (ndims < 1 || ndims > 8) return INTERPN_HIP_ERR_TOO_MANY_DIMS;  // regular.rs:111-113
  } else {
    if (ndims < 1 || ndims > 8) return INTERPN_HIP_ERR_TOO_MANY_DIMS;  // multicubic/regular.rs:130-132
    // multicubic/regular.rs:66-73 — try_into().unwrap() panics in the flattened arm
    if (ndims <= 4 && (nstarts != ndims || nsteps != ndims)) return INTERPN_HIP_ERR_REFERENCE_PANIC;
  }
  if (!dims || !starts || !steps) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  size_t prod;
  if (!checked_product(dims, ndims, &prod)) return INTERPN_HIP_ERR_REFERENCE_PANIC;
  if (method == kCubic && !(nstarts == ndims && nsteps == ndims)) return INTERPN_HIP_ERR_DIM_MISMATCH;
  if (nvals != prod) return INTERPN_HIP_ERR_DIM_MISMATCH;  // regular.rs:239 / multicubic/regular.rs:254
  const size_t minlen = method == kCubic ? 4 : 2;
  for (size_t i = 0; i < ndims; ++i)
    if (dims[i] < minlen) return method == kCubic ? INTERPN_HIP_ERR_MIN_FOUR_ENTRIES : INTERPN_HIP_ERR_MIN_TWO_ENTRIES;
  for (size_t i = 0; i < ndims; ++i)
    if (!(steps[i] > (T)0)) return INTERPN_HIP_ERR_NOT_MONOTONIC;  // regular.rs:248
  for (size_t i = 0; i < ndims; ++i)
    if (dims[i] > max_axis_len<T>()) return INTERPN_HIP_ERR_UNSUPPORTED;
  return INTERPN_HIP_OK;
}

template <typename T>
int validate_rectilinear(int method, const T* const* grids, const size_t* grid_lens, size_t ngrids, size_t nvals) {
  const size_t ndims = ngrids;
  if (method == kNearest) {
    if (ndims < 1 || ndims > 6) return INTERPN_HIP_ERR_TOO_MANY_DIMS_6;  // nearest/rectilinear.rs:59
  } else if (ndims < 1 || ndims > 8) return INTERPN_HIP_ERR_TOO_MANY_DIMS;  // rectilinear.rs:77-79
  if (!grids || !grid_lens) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  size_t prod;
  if (!checked_product(grid_lens, ndims, &prod)) return INTERPN_HIP_ERR_REFERENCE_PANIC;
  if (nvals != prod) return INTERPN_HIP_ERR_DIM_MISMATCH;  // rectilinear.rs:186 / multicubic/rectilinear.rs:208
  const size_t minlen = method == kCubic ? 4 : 2;
  for (size_t i = 0; i < ndims; ++i)
    if (grid_lens[i] < minlen) return method == kCubic ? INTERPN_HIP_ERR_MIN_4_ENTRIES : INTERPN_HIP_ERR_MIN_2_ENTRIES;
  for (size_t i = 0; i < ndims; ++i) {
    if (!grids[i]) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
    if (!(grids[i][1] > grids[i][0])) return INTERPN_HIP_ERR_NOT_MONOTONIC;  // rectilinear.rs:195
  }
  for (size_t i = 0; i < ndims; ++i)
    if (grid_lens[i] > max_axis_len<T>()) return INTERPN_HIP_ERR_UNSUPPORTED;
  return INTERPN_HIP_OK;
}

// `.interp(obs, out)` length checks (multilinear/regular.rs:271, multicubic/regular.rs:301, ...),
// including the flattened cubic arms' `obs.try_into().unwrap()` panic.
int validate_obs(const GridDesc& g, const size_t* obs_lens, size_t nobs, size_t nout) {
  const size_t ndims = (size_t)g.ndims;
  if (nobs != ndims) {
    if (g.method == kCubic && ndims <= 4) return INTERPN_HIP_ERR_REFERENCE_PANIC;
    return INTERPN_HIP_ERR_DIM_MISMATCH;
  }
  if (obs_lens)
    for (size_t i = 0; i < ndims; ++i)
      if (obs_lens[i] != nout) return INTERPN_HIP_ERR_DIM_MISMATCH;
  return INTERPN_HIP_OK;
}

int resolve_device(int device, int* out) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    t_last_hip_error = e != hipSuccess ? hipGetErrorString(e) : "no HIP device";
    return INTERPN_HIP_ERR_NO_DEVICE;
  }
  if (device < 0) {
    HIP_TRY(hipGetDevice(&device));
  }
  if (device >= count) return INTERPN_HIP_ERR_NO_DEVICE;
  *out = device;
  return INTERPN_HIP_OK;
}

// Bricked copy of the grid for the multilinear kernels with 3 <= N <= 6 (k_linear_brick.hip).  The layout
// is chosen by where the table will live: fully overlapped bricks (one line per cell, 5.3x the
// grid) while that still fits the 4 MiB XCD L2 or once the grid is far beyond it anyway (served
// by the 256 MiB Infinity Cache, where fewer lines per point matter most); in between, the
// cheaper overlaps that keep most of the table L2-resident.  INTERPN_HIP_BRICKS=off|11|12|22
// overrides (tuning).
// Tiled copy for the multicubic kernels (k_cubic_brick.hip), N = 2..4: dims 0,1 in 4 x 4 tiles.
// Candidates are ranked by a two-level cost model: lines per point x (L2 hit ? 1/2.7e11 : 1/6.2e10 s),
// with the hit fraction ~ min(1, 3 MiB / table bytes) — the measured L2 and Infinity-Cache line
// rates.  INTERPN_HIP_BRICKS=off|44|24|22|14|11 overrides.
// A grid of at most 16 KiB stays resident in every CU's 32 KiB vector L1, where the plain C-order
// gather beats the cooperative brick gather and its LDS exchange (measured, 1e8 points: 2-D linear
// 45^2 0.53 vs 0.70 ms, 3-D linear 12^3 0.74 vs 0.86 ms, 2-D cubic 45^2 1.16 vs 1.53 ms; from
// 32 KiB on the bricks win: 16^3 0.93 vs 1.38 ms).  An explicit INTERPN_HIP_BRICKS layout still
// applies (tests).
static bool grid_is_l1_resident(const GridDesc& g) {
  return g.nvals * (g.dtype == kF64 ? 8u : 4u) <= 16u * 1024u;
}

int maybe_build_cubic_tiles(interpn_hip_interp* h) {
  GridDesc& g = h->desc;
  const char* env = getenv("INTERPN_HIP_BRICKS");
  if (env && !strcmp(env, "off")) return INTERPN_HIP_OK;
  if (!(env && strlen(env) == 2) && grid_is_l1_resident(g)) return INTERPN_HIP_OK;
  static const int cand[5][2] = {{4, 4}, {2, 4}, {2, 2}, {1, 4}, {1, 1}};
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
  int best = -1;
  double best_cost = 0;
  for (int c = 0; c < 5; ++c) {
    const int si = cand[c][0], sj = cand[c][1];
    if (env && strlen(env) == 2 && !(env[0] - '0' == si && env[1] - '0' == sj)) continue;
    unsigned nb[2];
    size_t bytes;
    cubic_tile_geometry(g, si, sj, nb, &bytes);
    // byte offsets into the table are 32-bit in the kernel (buffer loads, cubic_brick.h)
    if (bytes >= 0xFFFFF000ull || bytes > free_b / 2) continue;
    const double e_i = si == 4 ? 1.75 : (si == 2 ? 1.5 : 1.0);
    const double e_j = sj == 4 ? 1.75 : (sj == 2 ? 1.5 : 1.0);
    double planes = 1;
    for (int d = 2; d < g.ndims; ++d) planes *= 4;
    const double lines = planes * e_i * e_j;
    double hit = (3.5 * 1048576.0) / (double)bytes;  // share of the 4 MiB L2 the table keeps beside the non-temporal streams
    if (hit > 1) hit = 1;
    const double cost = lines * (hit / 2.7e11 + (1 - hit) / 6.2e10);
    if (best < 0 || cost < best_cost) { best = c; best_cost = cost; }
  }
  if (best < 0) return INTERPN_HIP_OK;
  const bool forced = env && strlen(env) == 2;
  unsigned nb11[2];
  size_t bytes11 = 0;
  cubic_tile_geometry(g, 1, 1, nb11, &bytes11);
  const bool fits11 = bytes11 < 0xFFFFF000ull && bytes11 <= free_b / 2;
  // f64 fully overlapped tiles are gathered by LDS-DMA (cubic_brick.h), which the line-rate model
  // above does not know: measured in place on 1e7 points (tools/cubic4_layout_probe.py and the
  // N = 2, 3 probe of the same session), (1,1) beats the model's choice whenever its table is at
  // most 8 MiB (4-D 16^4: 1.11 vs 1.42 ms; 3-D 40^3: 0.35 vs 0.45; 2-D 256^2: 0.143 vs 0.168), for
  // every 2-D grid (512^2: 0.19 vs 0.20) and for every rectilinear grid (VALU-bound kernels:
  // 3-D 64^3 0.91 vs 0.94, 4-D 20^4 2.88 vs 2.96).
  if (!forced && g.dtype == kF64 && fits11 &&
      (bytes11 <= ((size_t)8 << 20) || g.ndims == 2 || g.kind == kRectilinear))
    best = 4;
  size_t bytes;
  g.brick_step[0] = cand[best][0];
  g.brick_step[1] = cand[best][1];
  cubic_tile_geometry(g, g.brick_step[0], g.brick_step[1], g.brick_nb, &bytes);
  g.brick_nb[2] = 1;
  hipError_t e = pool_alloc(h->device, &h->bricks_owned, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); h->bricks_owned = nullptr; return INTERPN_HIP_OK; }
  HIP_TRY(build_cubic_tiles(g, h->bricks_owned, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  g.bricks = h->bricks_owned;
  // 4-D grids in the band where an L2-friendly layout wins for small batches (20^4 .. 26^4 in f64,
  // 20^4 .. 30^4 in f32) also keep the fully overlapped table: large batches are evaluated binned
  // on it (eval_device_binned; f64 24^4 at 1e7 points: 1.38 against 1.79 ms, at 1e6: 0.154 against
  // 0.204; f32 28^4: 0.96 against 1.46 ms).
  g.bricks11 = nullptr;
  if (!forced && g.ndims == 4 && best != 4 && fits11) {
    if (pool_alloc(h->device, &h->bricks11_owned, bytes11) == hipSuccess) {
      GridDesc t = g;
      t.brick_step[0] = t.brick_step[1] = 1;
      t.brick_nb[0] = nb11[0];
      t.brick_nb[1] = nb11[1];
      HIP_TRY(build_cubic_tiles(t, h->bricks11_owned, nullptr));
      HIP_TRY(hipStreamSynchronize(nullptr));
      g.bricks11 = h->bricks11_owned;
      g.bricks11_nb[0] = nb11[0];
      g.bricks11_nb[1] = nb11[1];
    } else {
      (void)hipGetLastError();
      h->bricks11_owned = nullptr;
    }
  }
  return INTERPN_HIP_OK;
}

// 1-D multilinear on a rectilinear axis: one record per search bucket (k_linear1_records.hip).
// M uniform buckets are doubled from 2n until none holds two coordinates; axes that would need
// more than 128 MiB of records (strongly clustered coordinates), unsorted or non-finite axes keep
// the general kernel.  INTERPN_HIP_BRICKS=off disables.
int maybe_build_records1(interpn_hip_interp* h) {
  GridDesc& g = h->desc;
  const char* env = getenv("INTERPN_HIP_BRICKS");
  if (env && !strcmp(env, "off")) return INTERPN_HIP_OK;
  if (!g.axis_buckets[0] || g.n[0] < 2 || !g.grid[0] || !g.vals) return INTERPN_HIP_OK;  // no table: axis not proven sorted
  // An axis image that fits the LDS budget of the 1-D kernel is searched there at the stream
  // rate already (<= 512 points: 0.33 ms per 1e8 points against 0.5-0.67 ms from records); the
  // records serve the longer axes, whose search otherwise goes through L1/L2 (4096 points:
  // 1.25 -> 0.71 ms).  INTERPN_HIP_BRICKS=on builds them regardless (tests).
  if (g.axis_image_bytes <= kMaxGridLdsBytesWide && !(env && !strcmp(env, "on"))) return INTERPN_HIP_OK;
  const double span = g.bound_hi[0] - g.bound_lo[0];
  if (!(span > 0) || !std::isfinite(span)) return INTERPN_HIP_OK;
  unsigned* maxpop_dev = nullptr;
  if (pool_alloc(h->device, (void**)&maxpop_dev, sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return INTERPN_HIP_OK; }
  int st = INTERPN_HIP_OK;
  for (long long M = 2LL * g.n[0]; M <= (1LL << 24) && records1_bytes(g, (int)M) <= ((size_t)128 << 20); M *= 2) {
    double scale = (double)M / span;
    if (g.dtype == kF32) scale = (double)(float)scale;
    if (!(scale > 0) || !std::isfinite(scale)) break;
    void* recs = nullptr;
    if (pool_alloc(h->device, &recs, records1_bytes(g, (int)M)) != hipSuccess) { (void)hipGetLastError(); break; }
    unsigned maxpop = 2;
    hipError_t e = build_records1(g, (int)M, scale, recs, maxpop_dev, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&maxpop, maxpop_dev, sizeof(unsigned), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { pool_free(h->device, recs); st = hip_fail(e); break; }
    if (maxpop <= 1) {
      h->bricks_owned = recs;
      g.bricks = recs;
      g.rec1_buckets = (int)M;
      g.rec1_scale = scale;
      break;
    }
    pool_free(h->device, recs);
  }
  pool_free(h->device, maxpop_dev);
  return st;
}

int maybe_build_bricks(interpn_hip_interp* h) {
  GridDesc& g = h->desc;
  if (g.method == kCubic && g.ndims >= 2 && g.ndims <= 4) return maybe_build_cubic_tiles(h);
  if (g.method == kLinear && g.ndims == 2) {
    const char* env2 = getenv("INTERPN_HIP_BRICKS");
    if (env2 && !strcmp(env2, "off")) return INTERPN_HIP_OK;
    if (!(env2 && !strcmp(env2, "on")) && grid_is_l1_resident(g)) return INTERPN_HIP_OK;
    size_t bytes2;
    brick2_geometry(g, g.brick_nb, &bytes2);
    g.brick_nb[2] = 1;
    size_t free2 = 0, total2 = 0;
    if (hipMemGetInfo(&free2, &total2) != hipSuccess) free2 = (size_t)8 << 30;
    if (bytes2 > free2 / 4 || bytes2 / (g.dtype == kF64 ? 8 : 4) >= 0xFFFFFFFFull) return INTERPN_HIP_OK;
    hipError_t e2 = pool_alloc(h->device, &h->bricks_owned, bytes2);
    if (e2 != hipSuccess) { (void)hipGetLastError(); h->bricks_owned = nullptr; return INTERPN_HIP_OK; }
    HIP_TRY(build_bricks2(g, h->bricks_owned, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));
    g.bricks = h->bricks_owned;
    return INTERPN_HIP_OK;
  }
  if (g.method == kLinear && g.ndims == 1 && g.kind == kRectilinear) return maybe_build_records1(h);
  if (!(g.method == kLinear && g.ndims >= 3 && g.ndims <= 6)) return INTERPN_HIP_OK;
  const char* env = getenv("INTERPN_HIP_BRICKS");
  if (env && !strcmp(env, "off")) return INTERPN_HIP_OK;
  int si = 0, sj = 0;
  bool cell = false;
  const size_t MiB = (size_t)1 << 20;
  const size_t esz = g.dtype == kF64 ? 8 : 4;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
  auto fits = [&](size_t b) { return b <= free_b / 4 && b <= ((size_t)16 << 30) && b / esz < 0xFFFFFFFFull; };
  unsigned nbc[4] = {0, 0, 0, 0};
  size_t bcell = 0;
  if (g.ndims >= 4) brick_cell_geometry(g, nbc, &bcell);
  bool j4 = false;       // f32 2 x 4 x 4 bricks (linear_brick.h, CELL == 2)
  unsigned nbj4[3] = {0, 0, 0};
  size_t bj4 = 0;
  if (g.dtype == kF32) brick_j4_geometry(g, nbj4, &bj4);
  if (env && !strcmp(env, "c4") && g.ndims >= 4) {
    cell = true;  // forced 4-D cell bricks (tests / tuning); INTERPN_HIP_BRICKS=c4 is ignored for N = 3
  } else if (env && !strcmp(env, "j4") && g.dtype == kF32) {
    j4 = true;    // forced (tests / tuning); ignored for f64
  } else if (env && strlen(env) == 2 && (env[0] == '1' || env[0] == '2') && (env[1] == '1' || env[1] == '2')) {
    si = env[0] - '0';
    sj = env[1] - '0';
  } else {
    if (grid_is_l1_resident(g)) return INTERPN_HIP_OK;
    // Measured on MI355X (tools/sweep_layouts.py, 3-D f64, 24^3 .. 384^3, non-temporal streams):
    // the fully overlapped layout (one line per cell) wins while its table is L2-sized (<= 6 MiB)
    // and again once even the (2,2) table is far beyond the 4 MiB L2 (Infinity-Cache- or
    // HBM-resident: random 128-B lines stream at > 5 TB/s, so fewer lines per point is all that
    // counts).  In between (56^3 .. 80^3 in f64) the layouts that stay mostly L2-resident win:
    // (1,2) while it fits, then (2,2).
    // N >= 4: the 4-D cell bricks halve the lines per point again (2^(N-4) instead of 2^(N-3) for
    // the fully overlapped 3-D bricks) at 3x their size; they follow the same rule one level up:
    // taken while L2-sized, and once the 3-D layouts no longer fit the L2 either.
    unsigned nb[3];
    size_t b11, b12, b22;
    brick_geometry(g, 1, 1, nb, &b11);
    brick_geometry(g, 1, 2, nb, &b12);
    brick_geometry(g, 2, 2, nb, &b22);
    // (tools/sweep_linear_nd.py, profiles/r02_sweep_linear_nd.txt: within 6 % of the best forced
    // layout at every size, N = 4..6)
    // f32 (round 3): the 2 x 4 x 4 bricks are the one-line-per-cell layout at 0.78x the size of the
    // fully overlapped 2 x 2 x 8 bricks; they take its place in the rule (tools/sweep_f32.py,
    // profiles/r03_sweep_f32_layouts.txt: never slower than (1,1); 64^3 0.73 -> 0.66 ms, 72^3
    // 0.81 ((1,2)) -> 0.76, 128^3 1.77 -> 1.70).
    const bool f32 = g.dtype == kF32;
    const size_t bone = f32 ? bj4 : b11;  // the one-line-per-cell table of this element type
    if (g.ndims >= 4 && fits(bcell) && (bcell <= 6 * MiB || b22 > 3 * MiB)) {
      cell = true;
    } else if (fits(bone)) {
      if (bone <= 6 * MiB || b22 > 6 * MiB) { si = 1; sj = 1; j4 = f32; }
      else if (b12 <= 6 * MiB) { si = 1; sj = 2; }
      else { si = 2; sj = 2; }
    } else if (fits(b12)) { si = 1; sj = 2; }
    else if (fits(b22)) { si = 2; sj = 2; }
    else return INTERPN_HIP_OK;  // stay on the C-order kernel
  }
  size_t bytes;
  if (j4) {
    bytes = bj4;
    for (int k = 0; k < 3; ++k) g.brick_nb[k] = nbj4[k];
    g.brick_nb[3] = 0;
    si = sj = 1;
  } else if (cell) {
    bytes = bcell;
    for (int k = 0; k < 4; ++k) g.brick_nb[k] = nbc[k];
    si = sj = 1;
  } else {
    brick_geometry(g, si, sj, g.brick_nb, &bytes);
    g.brick_nb[3] = 0;
  }
  // brick element offsets are 32-bit in the kernel
  if (bytes / esz >= 0xFFFFFFFFull) return INTERPN_HIP_OK;
  if (bytes > free_b / 2) return INTERPN_HIP_OK;
  g.brick_step[0] = si;
  g.brick_step[1] = sj;
  g.brick_cell = j4 ? 2 : (cell ? 1 : 0);
  hipError_t e = pool_alloc(h->device, &h->bricks_owned, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); h->bricks_owned = nullptr; g.brick_cell = 0; return INTERPN_HIP_OK; }
  HIP_TRY(build_bricks(g, h->bricks_owned, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  g.bricks = h->bricks_owned;
  return INTERPN_HIP_OK;
}

// Options by name (interpn_hip_set_option / interpn_hip_get_option, and the INTERPN_HIP_<NAME>
// environment variables latched at creation).  Returns false for an unknown name or a value out
// of range; `set == false` reads.
bool option_access(LaunchConfig& c, const char* name, long long* value, bool set) {
  struct Opt { const char* name; int* field; long long lo, hi; };
  const Opt opts[] = {
      {"blocks_per_cu", &c.blocks_per_cu, 1, 65536},
      {"iters_per_block", &c.iters_per_block, 0, 65536},
      {"ppl", &c.ppl, 0, 2},
      {"axis_regs", &c.axis_regs, -1, 2},
      {"force_generic", &c.force_generic, 0, 1},
      {"generic_runtime", &c.generic_runtime, 0, 1},
      {"generic_vec", &c.generic_vec, -1, 1},
      {"persistent", &c.persistent, 0, 1},
      {"axis_lds_kb", &c.axis_lds_kb, -1, 60},
      {"binned", &c.binned, -1, 1},
      {"deal", &c.deal, 0, 1},
      {"bin_slice_log2", &c.bin_slice_log2, 16, 27},
      {"column", &c.column, -1, 1},
      {"column_part", &c.column_part, 0, 1 << 20},
      {"column_threads", &c.column_threads, 256, 1024},
      {"column_groups", &c.column_groups, 1, 2},
      {"column_cpp", &c.column_cpp, 0, 1 << 20},
      {"bin_scramble", &c.bin_scramble, 0, 1},
      {"stage_timing", &c.stage_timing, 0, 1},
      {"axis_records", &c.axis_records, 0, 1},
  };
  if (!name || !value) return false;
  if (!strcmp(name, "host_chunk")) {
    if (set) {
      if (*value < 0) return false;
      c.host_chunk = *value;
    } else {
      *value = c.host_chunk;
    }
    return true;
  }
  if (!strcmp(name, "debug_stamps")) {  // a device address (measurement aid, cubic_column.h)
    if (set) c.debug_stamps = *value;
    else *value = c.debug_stamps;
    return true;
  }
  for (const Opt& o : opts) {
    if (strcmp(name, o.name)) continue;
    if (set) {
      if (*value < o.lo || *value > o.hi) return false;
      *o.field = (int)*value;
    } else {
      *value = *o.field;
    }
    return true;
  }
  return false;
}

// The environment is read here, once per handle, and nowhere on the launch path.
void latch_env(LaunchConfig& c) {
  static const char* const names[] = {"blocks_per_cu", "iters_per_block", "ppl", "axis_regs", "force_generic",
                                      "generic_runtime", "generic_vec", "persistent", "axis_lds_kb", "host_chunk", "binned", "deal",
                                      "bin_slice_log2", "column", "column_part", "column_threads", "column_groups", "column_cpp", "axis_records", "bin_scramble"};
  for (const char* nm : names) {
    char var[64] = "INTERPN_HIP_";
    size_t k = strlen(var);
    for (const char* q = nm; *q && k + 1 < sizeof(var); ++q) var[k++] = (char)toupper((unsigned char)*q);
    var[k] = 0;
    const char* env = getenv(var);
    if (!env || !*env) continue;
    char* end = nullptr;
    long long v = strtoll(env, &end, 10);
    if (end == env) continue;
    (void)option_access(c, nm, &v, true);  // out-of-range values are ignored, as before
  }
}

int finish_create(interpn_hip_interp* h, const void* vals, size_t nvals, size_t elem, int vals_mem) {
  GridDesc& g = h->desc;
  g.cfg.num_cus = device_num_cus(h->device);
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
    const size_t cap = ngrids <= 2 ? kMaxGridLdsBytesWide : kMaxGridLdsBytes;
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

template <typename T>
hipError_t launch(const GridDesc& g, const T* const* obs, T* out, size_t npts, unsigned long long* first_bad,
                  hipStream_t stream) {
  if (npts == 0) return hipSuccess;
  if (g.method == kNearest) return launch_nearest<T>(g, obs, out, npts, first_bad, stream);
  if (g.cfg.force_generic || !fast_path(g)) return launch_generic<T>(g, obs, out, npts, first_bad, stream);
  if (g.method == kLinear)
    return g.kind == kRegular ? launch_linear_regular<T>(g, obs, out, npts, first_bad, stream)
                              : launch_linear_rectilinear<T>(g, obs, out, npts, first_bad, stream);
  return g.kind == kRegular ? launch_cubic_regular<T>(g, obs, out, npts, first_bad, stream)
                            : launch_cubic_rectilinear<T>(g, obs, out, npts, first_bad, stream);
}

hipError_t launch_any(const GridDesc& g, const void* const* obs, void* out, size_t npts,
                      unsigned long long* first_bad, hipStream_t stream) {
  if (g.bricks && npts && g.method == kCubic && !g.cfg.force_generic) {
    if (g.dtype == kF64)
      return launch_cubic_brick<double>(g, reinterpret_cast<const double* const*>(obs), static_cast<double*>(out),
                                        npts, first_bad, stream);
    return launch_cubic_brick<float>(g, reinterpret_cast<const float* const*>(obs), static_cast<float*>(out), npts,
                                     first_bad, stream);
  }
  if (g.bricks && npts && g.ndims == 1 && g.method == kLinear && g.rec1_buckets && !g.cfg.force_generic) {
    if (g.dtype == kF64)
      return launch_linear1_records<double>(g, reinterpret_cast<const double* const*>(obs), static_cast<double*>(out), npts, stream);
    return launch_linear1_records<float>(g, reinterpret_cast<const float* const*>(obs), static_cast<float*>(out), npts, stream);
  }
  if (g.bricks && npts && g.ndims == 2 && !g.cfg.force_generic) {
    if (g.dtype == kF64)
      return launch_linear2_brick<double>(g, reinterpret_cast<const double* const*>(obs), static_cast<double*>(out),
                                          npts, first_bad, stream);
    return launch_linear2_brick<float>(g, reinterpret_cast<const float* const*>(obs), static_cast<float*>(out), npts,
                                       first_bad, stream);
  }
  if (g.bricks && npts && !g.cfg.force_generic) {
    if (g.dtype == kF64)
      return launch_linear_brick<double>(g, reinterpret_cast<const double* const*>(obs), static_cast<double*>(out),
                                         npts, first_bad, stream);
    return launch_linear_brick<float>(g, reinterpret_cast<const float* const*>(obs), static_cast<float*>(out), npts,
                                      first_bad, stream);
  }
  if (g.dtype == kF64)
    return launch<double>(g, reinterpret_cast<const double* const*>(obs), static_cast<double*>(out), npts, first_bad, stream);
  return launch<float>(g, reinterpret_cast<const float* const*>(obs), static_cast<float*>(out), npts, first_bad, stream);
}

// Chunk size of the host pipeline (points).  Bounded so that the workspace stays modest
// (8 dims x 8 B x 4 Mi = 256 MiB worst case) while each kernel launch still fills the chip.
constexpr size_t kHostChunkPoints = (size_t)4 << 20;

// Completion of an 8-byte device-to-pinned-host copy, by watching the landing word instead of
// calling hipStreamSynchronize: the caller stores kWordPending into *word, enqueues the copy on
// `s`, then calls this.  The copy is ordered behind everything enqueued on `s` before it, so once
// the word has changed that work is complete ON THE DEVICE: results in device memory may be used
// by anything enqueued afterwards.  It says nothing about data a kernel wrote into HOST memory
// (the zero-copy small-batch path keeps the runtime's wait for that reason).  Spinning on a
// pinned, host-coherent word costs a few microseconds less per call than the runtime's wait;
// after 200 us without an answer the runtime's wait takes over.
constexpr unsigned long long kWordPending = 0xFFFFFFFFFFFFFFFEull;  // neither "no failure" (~0) nor an index
hipError_t wait_status_word(hipStream_t s, const unsigned long long* word) {
  const volatile unsigned long long* w = word;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    if (*w != kWordPending) return hipSuccess;
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
    if ((spin & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) break;
  }
  return hipStreamSynchronize(s);
}

int ensure_lane(interpn_hip_interp* h, int which, size_t points) {
  interpn_hip_interp::HostLane& l = h->lane[which];
  if (l.points >= points && l.obs) return INTERPN_HIP_OK;
  const size_t elem = h->desc.dtype == kF64 ? 8 : 4;
  if (l.obs) { HIP_TRY(hipStreamSynchronize(l.stream)); pool_free(h->device, l.obs); l.obs = nullptr; }
  if (l.out) { pool_free(h->device, l.out); l.out = nullptr; }
  l.points = 0;
  if (!l.stream) HIP_TRY(pool_take_kit(h->device, &l.stream, &l.flag_host));
  HIP_TRY(pool_alloc(h->device, &l.obs, (size_t)h->desc.ndims * points * elem));
  HIP_TRY(pool_alloc(h->device, &l.out, points * elem));
  if (!l.flag_dev) {
    HIP_TRY(pool_alloc(h->device, (void**)&l.flag_dev, sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(l.flag_dev, 0xFF, sizeof(unsigned long long), l.stream));
  }
  l.points = points;
  return INTERPN_HIP_OK;
}

// check_bounds over host arrays: stream each dimension's coordinates through the device and OR
// the per-point violations (multilinear/regular.rs:168-171).
template <typename T>
int check_bounds_host(const T* lo, const T* hi, size_t ndims, const T* const* obs, const size_t* obs_lens, T atol,
                      uint8_t* out) {
  int dev;
  int st = resolve_device(-1, &dev);
  if (st) return st;
  unsigned* flags = nullptr;
  T* buf = nullptr;
  size_t maxlen = 0;
  for (size_t d = 0; d < ndims; ++d) maxlen = obs_lens[d] > maxlen ? obs_lens[d] : maxlen;
  const size_t chunk = maxlen < kHostChunkPoints ? (maxlen ? maxlen : 1) : kHostChunkPoints;
  hipError_t e = pool_alloc(dev, (void**)&flags, sizeof(unsigned) * (ndims ? ndims : 1));
  if (e == hipSuccess) e = hipMemsetAsync(flags, 0, sizeof(unsigned) * (ndims ? ndims : 1), nullptr);
  if (e == hipSuccess) e = pool_alloc(dev, (void**)&buf, chunk * sizeof(T));
  for (size_t d = 0; d < ndims && e == hipSuccess; ++d) {
    if (obs_lens[d] && !obs[d]) { e = hipErrorInvalidValue; break; }
    for (size_t begin = 0; begin < obs_lens[d] && e == hipSuccess; begin += chunk) {
      const size_t count = obs_lens[d] - begin < chunk ? obs_lens[d] - begin : chunk;
      e = hipMemcpy(buf, obs[d] + begin, count * sizeof(T), hipMemcpyHostToDevice);
      if (e == hipSuccess) e = launch_check_bounds<T>(buf, count, lo[d], hi[d], atol, flags + d, nullptr);
      if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    }
  }
  std::vector<unsigned> host(ndims ? ndims : 1, 0);
  if (e == hipSuccess) e = hipMemcpy(host.data(), flags, sizeof(unsigned) * (ndims ? ndims : 1), hipMemcpyDeviceToHost);
  (void)hipStreamSynchronize(nullptr);
  pool_free(dev, buf);
  pool_free(dev, flags);
  if (e != hipSuccess) return hip_fail(e);
  for (size_t d = 0; d < ndims; ++d) out[d] = host[d] ? 1 : 0;
  return INTERPN_HIP_OK;
}

// Remember that device-pointer work was enqueued on `stream` (see interpn_hip_interp::marks).
// No allocation, copy or synchronisation on the common path (the event of a known stream is
// re-recorded); a stream under capture is never touched, so eval_device stays graph-capturable.
void mark_stream(interpn_hip_interp* h, hipStream_t stream) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) {
    (void)hipGetLastError();
    std::lock_guard<std::mutex> lk(h->marks_mu);
    h->sync_device_at_destroy = true;
    return;
  }
  std::lock_guard<std::mutex> lk(h->marks_mu);
  if (cs != hipStreamCaptureStatusNone) {
    h->sync_device_at_destroy = true;  // the graph may replay this launch at any later time
    return;
  }
  for (auto& m : h->marks)
    if (m.stream == stream) {
      if (hipEventRecord(m.event, stream) != hipSuccess) { (void)hipGetLastError(); h->sync_device_at_destroy = true; }
      return;
    }
  hipEvent_t ev = nullptr;
  if (h->marks.size() >= interpn_hip_interp::kMaxMarks ||
      hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    h->sync_device_at_destroy = true;
    return;
  }
  if (hipEventRecord(ev, stream) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipEventDestroy(ev);
    h->sync_device_at_destroy = true;
    return;
  }
  h->marks.push_back({stream, ev});
}

}  // namespace

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

int interpn_hip_set_blocks_per_cu(interpn_hip_interp* h, int blocks_per_cu) {
  if (!h || blocks_per_cu < 1 || blocks_per_cu > 65536) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  h->desc.cfg.blocks_per_cu = blocks_per_cu;
  return INTERPN_HIP_OK;
}

int interpn_hip_set_option(interpn_hip_interp* h, const char* name, long long value) {
  if (!h || !name) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if (!strcmp(name, "fma")) {  // the flavour is read at every launch; tables do not depend on it
    if (value != 0 && value != 1) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
    h->desc.fma = (int)value;
    return INTERPN_HIP_OK;
  }
  return option_access(h->desc.cfg, name, &value, true) ? INTERPN_HIP_OK : INTERPN_HIP_ERR_INVALID_ARGUMENT;
}

int interpn_hip_get_option(const interpn_hip_interp* h, const char* name, long long* value) {
  if (!h || !name || !value) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if (!strcmp(name, "last_binned")) {  // read-only: did the most recent device-pointer evaluation sort its points first?
    *value = h->desc.last_binned;
    return INTERPN_HIP_OK;
  }
  if (!strcmp(name, "fma")) { *value = h->desc.fma; return INTERPN_HIP_OK; }
  // read-only: how the handle's rectilinear axes will be searched (0 no records, 1 full, 2 compact) and what is staged
  if (!strcmp(name, "axis_rec_mode")) { *value = h->desc.axis_rec_bytes ? (h->desc.axis_rec_compact ? 2 : 1) : 0; return INTERPN_HIP_OK; }
  if (!strcmp(name, "axis_rec_bytes")) { *value = h->desc.axis_rec_bytes; return INTERPN_HIP_OK; }
  if (!strcmp(name, "axis_image_bytes")) { *value = h->desc.axis_image_bytes; return INTERPN_HIP_OK; }
  if (!strcmp(name, "evals_binned")) { *value = h->evals_binned.load(); return INTERPN_HIP_OK; }
  if (!strcmp(name, "evals_in_place")) { *value = h->evals_in_place.load(); return INTERPN_HIP_OK; }
  if (!strcmp(name, "scratch_allocs")) { *value = h->scratch_allocs.load(); return INTERPN_HIP_OK; }
  if (!strcmp(name, "scratch_bytes")) {
    interpn_hip_interp* hm = const_cast<interpn_hip_interp*>(h);
    std::lock_guard<std::mutex> lk(hm->bin_mu);
    long long tot = 0;
    for (const auto& sl : hm->bin_slots) tot += (long long)sl.bytes;
    *value = tot;
    return INTERPN_HIP_OK;
  }
  LaunchConfig c = h->desc.cfg;
  return option_access(c, name, value, false) ? INTERPN_HIP_OK : INTERPN_HIP_ERR_INVALID_ARGUMENT;
}

// "interpn::k_linear_brick<double, 3, false, true, 1, 2, 2, 0>" — rocprofv3's spelling of the
// instantiation, without the return type and the argument list.
int interpn_hip_kernel_name(const interpn_hip_interp* h, char* buf, size_t buflen) {
  if (!h || !buf || buflen == 0) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  const KernelTag t = h->desc.tag;
  if (!t.name) {
    buf[0] = 0;
    return INTERPN_HIP_OK;
  }
  std::string out = std::string("interpn::") + t.name + "<" + (h->desc.dtype == kF64 ? "double" : "float");
  for (int k = 0; k < t.nargs; ++k) {
    out += ", ";
    if (t.bool_mask & (1u << k)) out += t.args[k] ? "true" : "false";
    else out += std::to_string(t.args[k]);
  }
  out += ">";
  snprintf(buf, buflen, "%s", out.c_str());
  return INTERPN_HIP_OK;
}

// Bytes of the re-laid grid copy the handle keeps (0 = kernels read the C-ordered `vals`), and
// its layout steps; for reports.
size_t interpn_hip_table_bytes(const interpn_hip_interp* h, int* step_i, int* step_j) {
  if (!h || !h->desc.bricks) return 0;
  const GridDesc& g = h->desc;
  if (step_i) *step_i = g.brick_step[0];
  if (step_j) *step_j = g.brick_step[1];
  size_t bytes = 0;
  unsigned nb[3];
  unsigned nb4[4];
  if (g.method == kCubic) cubic_tile_geometry(g, g.brick_step[0], g.brick_step[1], nb, &bytes);
  else if (g.ndims == 1) bytes = records1_bytes(g, g.rec1_buckets);
  else if (g.ndims == 2) brick2_geometry(g, nb, &bytes);
  else if (g.brick_cell == 2) brick_j4_geometry(g, nb, &bytes);
  else if (g.brick_cell) brick_cell_geometry(g, nb4, &bytes);
  else brick_geometry(g, g.brick_step[0], g.brick_step[1], nb, &bytes);
  return bytes;
}

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
  pool_free(h->device, h->grids_owned);
  pool_free(h->device, h->bricks_owned);
  pool_free(h->device, h->bricks11_owned);
  pool_free(h->device, h->vals_owned);
  delete h;
}

namespace {

// Binned evaluation of the tiled multicubic kernels on device-resident points (interpn_host.h).
// Returns -1 when the path does not apply or cannot be taken right now (the caller then launches
// the kernel on the points as they are), otherwise a status.  Chosen automatically for 4-D grids
// whose tile table is far beyond the L2 and batches large enough to pay for the three sorting
// launches (cfg4: 2.8 -> 1.85 ms per 1e7 points; from about 5e5 points on; 3-D grids lose: 4 lines
// per point are cheaper than sorting them); `binned` = 1 forces it for N = 2..4 (tests),
// 0 turns it off.  Not taken while the stream is being captured into a graph (it may have to
// allocate) or while another thread is inside it with the same handle.
// Slice length of the sorted path (bounds one scratch block): option "bin_slice_log2".
size_t bin_slice_points(const GridDesc& g) {
  const int lg = g.cfg.bin_slice_log2 >= 16 && g.cfg.bin_slice_log2 <= 27 ? g.cfg.bin_slice_log2 : 25;
  const size_t s = (size_t)1 << lg;
  return s < kBinSlicePoints ? s : kBinSlicePoints;
}

// Does the sorted path apply to this handle at all / to a batch of `npoints`?  (No HIP calls.)
// Returns 0 = never for this handle, 1 = not for this batch (too small / switched off), 2 = yes.
int binned_applies(const GridDesc& g, size_t npoints) {
  if (g.method != kCubic || !g.bricks || g.ndims < 2 || g.ndims > 4) return 0;
  if (g.cfg.binned == 0 || g.cfg.force_generic) return 1;
  const bool main11 = g.brick_step[0] == 1 && g.brick_step[1] == 1;
  const bool second = !main11 && g.bricks11 != nullptr;
  if (g.cfg.binned < 0) {
    if (g.ndims != 4) return 0;
    if (!main11 && !second) return 0;
    if (main11) {
      unsigned nb[2];
      size_t table = 0;
      cubic_tile_geometry(g, 1, 1, nb, &table);
      if (table <= ((size_t)8 << 20)) return 0;  // an L2-sized table is gathered at the hit rate anyway
    }
    if (npoints < ((size_t)1 << 19)) return 1;
  }
  return 2;
}

// Take a scratch block of at least `need` bytes for an evaluation on `stream` (see BinSlot).
// On success the block is marked busy and, where another stream used it last, `stream` has been
// made to wait for that use.  `*why` says why not otherwise.
interpn_hip_interp::BinSlot* take_bin_slot(interpn_hip_interp* h, size_t need, hipStream_t stream, bool may_alloc, int* why) {
  using Slot = interpn_hip_interp::BinSlot;
  std::lock_guard<std::mutex> lk(h->bin_mu);
  Slot* pick = nullptr;
  bool wait = false;
  for (auto& sl : h->bin_slots)  // 1. the block this stream used last: stream order is enough
    if (!sl.busy && sl.bytes >= need && sl.recorded && sl.last_stream == stream) { pick = &sl; break; }
  if (!pick)
    for (auto& sl : h->bin_slots) {  // 2. an idle block
      if (sl.busy || sl.bytes < need) continue;
      if (!sl.recorded) { pick = &sl; break; }
      const hipError_t q = hipEventQuery(sl.event);
      if (q == hipSuccess) { pick = &sl; break; }
      if (q != hipErrorNotReady) (void)hipGetLastError();
    }
  if (!pick && may_alloc && h->bin_slots.size() < interpn_hip_interp::kMaxBinSlots) {  // 3. a new block
    Slot sl;
    if (hipEventCreateWithFlags(&sl.event, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); *why = INTERPN_HIP_WHY_ALLOC_FAILED; return nullptr; }
    if (pool_alloc(h->device, &sl.scratch, need) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipEventDestroy(sl.event);
      *why = INTERPN_HIP_WHY_ALLOC_FAILED;
      return nullptr;
    }
    sl.bytes = need;
    h->scratch_allocs.fetch_add(1);
    h->bin_slots.push_back(sl);
    pick = &h->bin_slots.back();
  }
  if (!pick) {  // 4. the least recently used block that is large enough: wait for it on the device
    for (auto& sl : h->bin_slots)
      if (!sl.busy && sl.bytes >= need && (!pick || sl.stamp < pick->stamp)) pick = &sl;
    wait = pick != nullptr;
  }
  if (!pick && may_alloc) {  // 5. grow the least recently used block that nobody is enqueueing into
    for (auto& sl : h->bin_slots)
      if (!sl.busy && (!pick || sl.stamp < pick->stamp)) pick = &sl;
    if (pick) {
      if (pick->recorded && hipEventSynchronize(pick->event) != hipSuccess) { (void)hipGetLastError(); *why = INTERPN_HIP_WHY_ALLOC_FAILED; return nullptr; }
      pool_free(h->device, pick->scratch);
      pick->scratch = nullptr;
      pick->bytes = 0;
      pick->totals_clean = false;
      pick->recorded = false;
      if (pool_alloc(h->device, &pick->scratch, need) != hipSuccess) { (void)hipGetLastError(); pick->scratch = nullptr; *why = INTERPN_HIP_WHY_ALLOC_FAILED; return nullptr; }
      pick->bytes = need;
      h->scratch_allocs.fetch_add(1);
    }
  }
  if (!pick) { *why = INTERPN_HIP_WHY_NO_SCRATCH; return nullptr; }
  if (wait && pick->recorded && pick->last_stream != stream &&
      hipStreamWaitEvent(stream, pick->event, 0) != hipSuccess) { (void)hipGetLastError(); *why = INTERPN_HIP_WHY_NO_SCRATCH; return nullptr; }
  if (!wait && pick->recorded && pick->last_stream != stream) (void)hipStreamWaitEvent(stream, pick->event, 0);  // complete already: free
  pick->busy = true;
  pick->stamp = ++h->bin_uses;
  return pick;
}

// Binned evaluation of the tiled multicubic kernels on device-resident points (interpn_host.h).
// Returns -1 when the path does not apply or cannot be taken right now (`*why` says which; the
// caller then launches the kernel on the points as they are), otherwise a status.  Chosen
// automatically for 4-D grids whose tile table is far beyond the L2 and batches large enough to
// pay for the sorting launches (cfg4: 2.8 -> 1.4 ms per 1e7 points; from about 5e5 points on;
// 3-D grids lose: 4 lines per point are cheaper than sorting them); `binned` = 1 forces it for
// N = 2..4 (tests), 0 turns it off.  Not taken while the stream is being captured into a graph.
int eval_device_binned(interpn_hip_interp* h, const void* const* obs, void* out, size_t npoints, hipStream_t stream,
                       unsigned flags, int* why) {
  const GridDesc& g = h->desc;
  *why = INTERPN_HIP_WHY_NONE;
  const int applies = binned_applies(g, npoints);
  if (applies < 2) { *why = applies ? INTERPN_HIP_WHY_SMALL_OR_OFF : INTERPN_HIP_WHY_NONE; return -1; }
  // The table the sorted points are evaluated on: the handle's own when it is the fully
  // overlapped one (or the evaluation is forced), else the second, fully overlapped table 4-D
  // handles keep for this purpose (maybe_build_cubic_tiles).
  const bool main11 = g.brick_step[0] == 1 && g.brick_step[1] == 1;
  const bool second = !main11 && g.bricks11 != nullptr;
  GridDesc second_desc;
  const GridDesc* use = &g;
  if (second) {
    second_desc = g;
    second_desc.bricks = g.bricks11;
    second_desc.brick_step[0] = second_desc.brick_step[1] = 1;
    second_desc.brick_nb[0] = g.bricks11_nb[0];
    second_desc.brick_nb[1] = g.bricks11_nb[1];
    use = &second_desc;
  }
  BinPlan plan;
  // Column evaluation (cubic_column.h): 4-D regular grids whose (k, l) column of tiles fits the LDS
  // and whose (i, j) cells fit one bin each — cfg4.  The sorted points of a cell are then
  // evaluated out of LDS instead of 16 L2 lines per point.
  bool column = g.cfg.column != 0 && (second || main11) && cubic_column_applies(*use);
  {
    // automatic mode: a workgroup's rows of 768 lanes must be mostly full and its column fill
    // amortised — from about 3000 points per (class pair) bin on (32^4: 4e6 points 0.53 against
    // 0.63 ms, 2e6 points 0.35 against 0.30; profiles/r03_cfg4_column_sizes.txt)
    const size_t slice_max0 = bin_slice_points(g);
    const size_t per_slice = npoints < slice_max0 ? npoints : slice_max0;
    if (g.cfg.column < 0 && per_slice < (size_t)3072 * (size_t)(g.n[0] - 1) * (size_t)(g.n[1] - 1)) column = false;
  }
  {
    unsigned nbt[2];
    size_t tbytes = 0;  // of the table the sorted points will be evaluated on
    if (second || main11) cubic_tile_geometry(g, 1, 1, nbt, &tbytes);
    else cubic_tile_geometry(g, g.brick_step[0], g.brick_step[1], nbt, &tbytes);
    if (column && !make_bin_plan(g, tbytes, &plan, /*classes=*/true)) column = false;
    if (!column && !make_bin_plan(g, tbytes, &plan)) { *why = INTERPN_HIP_WHY_SMALL_OR_OFF; return -1; }
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); *why = INTERPN_HIP_WHY_CAPTURE; return -1; }
  if (cs != hipStreamCaptureStatusNone) { *why = INTERPN_HIP_WHY_CAPTURE; return -1; }
  const size_t slice_max = bin_slice_points(g);
  const size_t slice = npoints < slice_max ? npoints : slice_max;
  interpn_hip_interp::BinSlot* slot =
      take_bin_slot(h, bin_scratch_bytes(g, slice), stream, !(flags & INTERPN_HIP_EVAL_NO_ALLOC), why);
  if (!slot) return -1;
  const size_t elem = g.dtype == kF64 ? 8 : 4;
  hipError_t err = hipSuccess;
  // per-stage timing on request (single-slice evaluations only)
  hipEvent_t* stage = nullptr;
  slot->staged = false;
  if (g.cfg.stage_timing && npoints <= slice) {
    bool okev = true;
    for (hipEvent_t& e : slot->stage)
      if (!e && hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); e = nullptr; okev = false; }
    if (okev) stage = slot->stage;
  }
  for (size_t begin = 0; begin < npoints && err == hipSuccess; begin += slice) {
    const size_t count = npoints - begin < slice ? npoints - begin : slice;
    const void* src[8];
    const void* sorted[8];
    for (int d = 0; d < g.ndims; ++d) src[d] = static_cast<const char*>(obs[d]) + begin * elem;
    const unsigned* index = nullptr;
    char* dst = static_cast<char*>(out) + begin * elem;
    if (column) {
      // a bin is cut into equal parts of at most 16 points per thread of the column workgroup (the
      // registers of its local sort); two such workgroups share a CU, the dispatcher hands parts
      // to whichever frees up
      ColumnPlan cplan;
      if (!cubic_column_plan(*use, &cplan)) { err = hipErrorInvalidValue; break; }
      size_t q = cplan.part_points;
      if (g.cfg.column_part > 0 && (size_t)g.cfg.column_part < q) q = (size_t)g.cfg.column_part;
      const size_t max_parts = 4 * (count / q) + (size_t)plan.nbins + 1;  // upper bound (the scan cuts the last bins finer)
      BinExtras extras;
      err = bin_points(g, plan, src, count, slot->scratch, sorted, &index, stream, &extras, (unsigned)q, stage, slot->totals_clean);
      slot->totals_clean = err == hipSuccess;
      if (err != hipSuccess) break;
      if (g.dtype == kF64)
        err = launch_cubic_column<double>(*use, plan, extras, index, reinterpret_cast<double*>(dst), count, max_parts, h->first_bad, begin, stream);
      else
        err = launch_cubic_column<float>(*use, plan, extras, index, reinterpret_cast<float*>(dst), count, max_parts, h->first_bad, begin, stream);
      continue;
    }
    err = bin_points(g, plan, src, count, slot->scratch, sorted, &index, stream, nullptr, 0, stage, slot->totals_clean);
    slot->totals_clean = err == hipSuccess;
    if (err != hipSuccess) break;
    if (g.dtype == kF64)
      err = launch_cubic_brick<double>(*use, reinterpret_cast<const double* const*>(sorted), reinterpret_cast<double*>(dst), count,
                                       h->first_bad, stream, index, begin);
    else
      err = launch_cubic_brick<float>(*use, reinterpret_cast<const float* const*>(sorted), reinterpret_cast<float*>(dst), count,
                                      h->first_bad, stream, index, begin);
  }
  if (stage && err == hipSuccess && hipEventRecord(stage[4], stream) == hipSuccess) slot->staged = true;
  // Whatever was enqueued — also a sequence cut short by a failure — is followed by the block's
  // event, so that the next user of the block on another stream waits for it.
  {
    std::lock_guard<std::mutex> lk(h->bin_mu);
    if (hipEventRecord(slot->event, stream) == hipSuccess) {
      slot->recorded = true;
    } else {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(stream);  // no event behind the work: make it complete before anyone reuses the block
      slot->recorded = false;
    }
    slot->last_stream = stream;
    slot->busy = false;
    if (err == hipSuccess) {
      h->desc.tag = use->tag;
      h->desc.last_binned = 1;
    }
  }
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return hip_fail(err);
  }
  return INTERPN_HIP_OK;
}

}  // namespace

int interpn_hip_eval_device_ex(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out, size_t npoints,
                               void* stream, unsigned flags, int* path_taken, int* why_out) {
  if (path_taken) *path_taken = INTERPN_HIP_PATH_IN_PLACE;
  if (why_out) *why_out = INTERPN_HIP_WHY_NONE;
  if (!h || (!obs && nobs)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if (flags & ~(unsigned)INTERPN_HIP_EVAL_NO_ALLOC) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  int st = validate_obs(h->desc, nullptr, nobs, npoints);
  if (st) return st;
  if (npoints == 0) return INTERPN_HIP_OK;
  if (!out) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < nobs; ++i)
    if (!obs[i]) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  int why = INTERPN_HIP_WHY_NONE;
  st = eval_device_binned(h, obs, out, npoints, static_cast<hipStream_t>(stream), flags, &why);
  if (why_out) *why_out = why;
  if (st > 0) {
    // part of the sequence may be in flight on `stream` without a mark behind it
    std::lock_guard<std::mutex> lk(h->marks_mu);
    h->sync_device_at_destroy = true;
    return st;
  }
  if (st < 0) {
    HIP_TRY(launch_any(h->desc, obs, out, npoints, h->first_bad, static_cast<hipStream_t>(stream)));
    if (binned_applies(h->desc, npoints)) {  // handles that can sort: keep the report field honest
      std::lock_guard<std::mutex> lk(h->bin_mu);
      h->desc.last_binned = 0;
    } else {
      h->desc.last_binned = 0;
    }
    h->evals_in_place.fetch_add(1);
  } else {
    if (path_taken) *path_taken = INTERPN_HIP_PATH_BINNED;
    h->evals_binned.fetch_add(1);
  }
  mark_stream(h, static_cast<hipStream_t>(stream));
  return INTERPN_HIP_OK;
}

int interpn_hip_eval_device(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out, size_t npoints,
                            void* stream) {
  return interpn_hip_eval_device_ex(h, obs, nobs, out, npoints, stream, 0u, nullptr, nullptr);
}

int interpn_hip_stage_ms(interpn_hip_interp* h, double* ms, size_t n) {
  if (!h || !ms || n < 4) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(h->bin_mu);
  const interpn_hip_interp::BinSlot* best = nullptr;
  for (const auto& sl : h->bin_slots)
    if (sl.staged && !sl.busy && (!best || sl.stamp > best->stamp)) best = &sl;
  if (!best) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipEventSynchronize(best->stage[4]));
  for (int k = 0; k < 4; ++k) {
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, best->stage[k], best->stage[k + 1]));
    ms[k] = (double)f;
  }
  return INTERPN_HIP_OK;
}

int interpn_hip_reserve(interpn_hip_interp* h, size_t npoints, int nstreams) {
  if (!h || nstreams < 0) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if ((size_t)nstreams > interpn_hip_interp::kMaxBinSlots) nstreams = (int)interpn_hip_interp::kMaxBinSlots;
  const GridDesc& g = h->desc;
  if (npoints == 0 || nstreams == 0 || binned_applies(g, npoints) < 2) return INTERPN_HIP_OK;  // nothing to provide
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  const size_t slice_max = bin_slice_points(g);
  const size_t need = bin_scratch_bytes(g, npoints < slice_max ? npoints : slice_max);
  std::lock_guard<std::mutex> lk(h->bin_mu);
  int have = 0;
  for (auto& sl : h->bin_slots)
    if (sl.bytes >= need) ++have;
  // grow blocks that are too small first (idle ones only), then add new ones
  for (auto& sl : h->bin_slots) {
    if (have >= nstreams) break;
    if (sl.bytes >= need || sl.busy) continue;
    if (sl.recorded) HIP_TRY(hipEventSynchronize(sl.event));
    pool_free(h->device, sl.scratch);
    sl.scratch = nullptr;
    sl.bytes = 0;
    sl.totals_clean = false;
    sl.recorded = false;
    hipError_t e = pool_alloc(h->device, &sl.scratch, need);
    if (e != hipSuccess) { (void)hipGetLastError(); sl.scratch = nullptr; return INTERPN_HIP_ERR_OUT_OF_MEMORY; }
    sl.bytes = need;
    h->scratch_allocs.fetch_add(1);
    ++have;
  }
  while (have < nstreams && h->bin_slots.size() < interpn_hip_interp::kMaxBinSlots) {
    interpn_hip_interp::BinSlot sl;
    HIP_TRY(hipEventCreateWithFlags(&sl.event, hipEventDisableTiming));
    hipError_t e = pool_alloc(h->device, &sl.scratch, need);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(sl.event); return INTERPN_HIP_ERR_OUT_OF_MEMORY; }
    sl.bytes = need;
    h->scratch_allocs.fetch_add(1);
    h->bin_slots.push_back(sl);
    ++have;
  }
  return have >= nstreams ? INTERPN_HIP_OK : INTERPN_HIP_ERR_OUT_OF_MEMORY;
}

int interpn_hip_finish(interpn_hip_interp* h, void* stream, uint64_t* first_bad_index) {
  if (!h) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // The status word lands in pinned memory: a plain DMA behind the kernel, no staging copy.
  std::lock_guard<std::mutex> lk(h->finish_mu);
  if (!h->finish_word) HIP_TRY(pool_take_pinned_word(h->device, &h->finish_word));
  *(volatile unsigned long long*)h->finish_word = kWordPending;
  HIP_TRY(hipMemcpyAsync(h->finish_word, h->first_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(wait_status_word(s, h->finish_word));
  const unsigned long long word = *(volatile unsigned long long*)h->finish_word;
  if (word == kNoBadIndexHost) return INTERPN_HIP_OK;
  HIP_TRY(hipMemsetAsync(h->first_bad, 0xFF, sizeof(word), s));
  HIP_TRY(hipStreamSynchronize(s));
  // Read the index only now: the spin above may have seen the word while the copy engine was
  // half-way through it (anything that is neither "pending" nor "clean" ends the spin); behind
  // the synchronisation the 8 bytes are complete.
  const unsigned long long settled = *(volatile unsigned long long*)h->finish_word;
  if (settled == kNoBadIndexHost) return INTERPN_HIP_OK;  // cannot happen for a monotone MIN word; harmless
  if (first_bad_index) *first_bad_index = (uint64_t)settled;
  return INTERPN_HIP_ERR_UNREPRESENTABLE;
}

// Chunked host evaluation.  Chunk c is handled by lane c % 2: upload, kernel, status word,
// download.  With more than one chunk the second lane runs on a helper thread, so that lane A's
// download overlaps lane B's next upload (the two directions use different DMA engines).  The
// reference's loop stops at the first failing point — out[0..i) written, out[i..] untouched
// (multilinear/regular.rs:277-280) — so a lane writes chunk c only once every chunk in front of
// it is known to be clean.  `*bad_index` (optional) receives the index of the first failing point
// when the status is INTERPN_HIP_ERR_UNREPRESENTABLE.
struct HostPipeline {
  interpn_hip_interp* h;
  const void* const* obs;
  void* out;
  size_t nout, chunk, nchunks;
  std::mutex mu;
  std::condition_variable cv;
  size_t checked[2] = {0, 0};      // chunks whose status word has been read, per lane
  size_t fail_chunk = ~(size_t)0;  // lowest failing chunk so far
  size_t fail_index = 0;           // global index of its first failing point
  int error = INTERPN_HIP_OK;      // first HIP failure of any lane

  int run_lane(int which) {
    DeviceGuard guard(h->device);
    if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
    const interpn_hip_interp::HostLane& l = h->lane[which];
    const size_t elem = h->desc.dtype == kF64 ? 8 : 4;
    const int nd = h->desc.ndims;
    const void* dev_obs[8];
    for (size_t c = (size_t)which; c < nchunks; c += 2) {
      {
        std::lock_guard<std::mutex> lk(mu);
        if (fail_chunk < c || error) break;
      }
      const size_t begin = c * chunk;
      const size_t count = (nout - begin) < chunk ? (nout - begin) : chunk;
      for (int d = 0; d < nd; ++d) {
        char* dst = (char*)l.obs + (size_t)d * l.points * elem;
        HIP_TRY(hipMemcpyAsync(dst, (const char*)obs[d] + begin * elem, count * elem, hipMemcpyHostToDevice, l.stream));
        dev_obs[d] = dst;
      }
      HIP_TRY(launch_any(h->desc, dev_obs, l.out, count, l.flag_dev, l.stream));
      HIP_TRY(hipMemcpyAsync(l.flag_host, l.flag_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, l.stream));
      HIP_TRY(hipStreamSynchronize(l.stream));
      const unsigned long long bad = *l.flag_host;
      if (bad != kNoBadIndexHost) HIP_TRY(hipMemsetAsync(l.flag_dev, 0xFF, sizeof(unsigned long long), l.stream));
      size_t good = count;
      bool stop = false;
      {
        std::unique_lock<std::mutex> lk(mu);
        if (bad != kNoBadIndexHost && c < fail_chunk) {
          fail_chunk = c;
          fail_index = begin + (size_t)bad;
        }
        checked[which] = c / 2 + 1;
        cv.notify_all();
        // every chunk in front of c must have reported before c may touch `out`
        const size_t need = (c + 1) / 2;  // chunks of the other lane in front of c
        cv.wait(lk, [&] { return checked[1 - which] >= need || error != INTERPN_HIP_OK; });
        if (error) break;
        if (fail_chunk < c) break;
        if (fail_chunk == c) {
          good = (size_t)bad;
          stop = true;
        }
      }
      if (good) HIP_TRY(hipMemcpyAsync((char*)out + begin * elem, l.out, good * elem, hipMemcpyDeviceToHost, l.stream));
      HIP_TRY(hipStreamSynchronize(l.stream));
      if (stop) break;
    }
    return INTERPN_HIP_OK;
  }

  // A lane that fails (HIP error) must release the other one.
  void lane_main(int which) {
    const int st = run_lane(which);
    std::lock_guard<std::mutex> lk(mu);
    if (st != INTERPN_HIP_OK && error == INTERPN_HIP_OK) error = st;
    checked[which] = ~(size_t)0;
    cv.notify_all();
  }
};

// Small batches (<= kSmallPoints points; BASELINE configs[0] is 1e3): zero-copy.  The CPU copies
// the coordinates into a pinned, device-mapped buffer, the kernel reads them and writes the results
// over PCIe, and the call costs one launch, one 8-byte status copy and ONE stream synchronisation
// instead of N + 2 staged copies and two synchronisations (1e3 points: 51 -> 2x us, see
// profiles/r02_host_path.txt).  Abort semantics as everywhere: only out[0..first_bad) is copied
// to the caller's array.
constexpr int kSmallPathUnavailable = -1;

static int eval_host_small(interpn_hip_interp* h, const void* const* obs, void* out, size_t nout, size_t* bad_index) {
  interpn_hip_interp::HostLane& l = h->lane[0];
  if (!h->small_host) {
    void* buf = nullptr;
    if (pool_take_small(h->device, &buf) != hipSuccess) { (void)hipGetLastError(); return kSmallPathUnavailable; }
    void* dbuf = nullptr;
    if (hipHostGetDevicePointer(&dbuf, buf, 0) != hipSuccess || !dbuf) {
      (void)hipGetLastError();
      pool_return_small(h->device, buf);
      return kSmallPathUnavailable;
    }
    h->small_host = buf;
    h->small_dev = dbuf;
  }
  if (!l.stream) HIP_TRY(pool_take_kit(h->device, &l.stream, &l.flag_host));
  if (!l.flag_dev) {
    HIP_TRY(pool_alloc(h->device, (void**)&l.flag_dev, sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(l.flag_dev, 0xFF, sizeof(unsigned long long), l.stream));
  }
  const size_t elem = h->desc.dtype == kF64 ? 8 : 4;
  const int nd = h->desc.ndims;
  const size_t stride = kSmallPoints * 8;  // bytes between the arrays: keeps every one 16-byte aligned
  const void* dev_obs[8];
  for (int d = 0; d < nd; ++d) {
    memcpy((char*)h->small_host + (size_t)d * stride, obs[d], nout * elem);
    dev_obs[d] = (const char*)h->small_dev + (size_t)d * stride;
  }
  char* host_out = (char*)h->small_host + (size_t)8 * stride;
  void* dev_out = (char*)h->small_dev + (size_t)8 * stride;
  HIP_TRY(launch_any(h->desc, dev_obs, dev_out, nout, l.flag_dev, l.stream));
  HIP_TRY(hipMemcpyAsync(l.flag_host, l.flag_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, l.stream));
  // The runtime's wait, not wait_status_word: here the RESULTS are written by the kernel straight
  // into pinned host memory, and the status word landing (a copy-engine write) does not order
  // those shader writes for the CPU — watching the word alone returned stale results once in
  // 37 739 fuzz cases.
  HIP_TRY(hipStreamSynchronize(l.stream));
  const unsigned long long bad = *l.flag_host;
  size_t good = nout;
  if (bad != kNoBadIndexHost) {
    HIP_TRY(hipMemsetAsync(l.flag_dev, 0xFF, sizeof(unsigned long long), l.stream));
    good = (size_t)bad;
  }
  if (good) memcpy(out, host_out, good * elem);
  if (bad != kNoBadIndexHost) {
    if (bad_index) *bad_index = (size_t)bad;
    return INTERPN_HIP_ERR_UNREPRESENTABLE;
  }
  return INTERPN_HIP_OK;
}

// Points per pipeline chunk: one chunk when the batch is small, else 2 Mi-point chunks.
constexpr size_t kPipelineChunkPoints = (size_t)2 << 20;

static int eval_host_impl(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out, size_t nout,
                          size_t* bad_index) {
  (void)nobs;
  std::lock_guard<std::mutex> host_lock(h->host_mu);
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  if (nout <= kSmallPoints && h->desc.cfg.host_chunk < 1) {
    const int sst = eval_host_small(h, obs, out, nout, bad_index);
    if (sst != kSmallPathUnavailable) return sst;
  }
  HostPipeline p;
  p.h = h;
  p.obs = obs;
  p.out = out;
  p.nout = nout;
  p.chunk = nout <= kPipelineChunkPoints ? nout : kPipelineChunkPoints;
  if (h->desc.cfg.host_chunk >= 1)  // testing: force small chunks
    p.chunk = (size_t)h->desc.cfg.host_chunk < nout ? (size_t)h->desc.cfg.host_chunk : nout;
  p.nchunks = (nout + p.chunk - 1) / p.chunk;
  int st = ensure_lane(h, 0, p.chunk);
  if (st) return st;
  if (p.nchunks > 1) {
    st = ensure_lane(h, 1, p.chunk);
    if (st) return st;
    std::thread helper([&p] { p.lane_main(1); });
    p.lane_main(0);
    helper.join();
  } else {
    p.checked[1] = ~(size_t)0;
    p.lane_main(0);
  }
  if (p.error) return p.error;
  if (p.fail_chunk != ~(size_t)0) {
    if (bad_index) *bad_index = p.fail_index;
    return INTERPN_HIP_ERR_UNREPRESENTABLE;
  }
  return INTERPN_HIP_OK;
}

int interpn_hip_eval_host(interpn_hip_interp* h, const void* const* obs, const size_t* obs_lens, size_t nobs,
                          void* out, size_t nout) {
  if (!h || (!obs && nobs) || (!obs_lens && nobs)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  int st = validate_obs(h->desc, obs_lens, nobs, nout);
  if (st) return st;
  if (nout == 0) return INTERPN_HIP_OK;
  if (!out) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < nobs; ++i)
    if (!obs[i]) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  return eval_host_impl(h, obs, nobs, out, nout, nullptr);
}

// Single-process multi-GPU form of `.interp(obs, out)`: the observation index is cut into
// `nhandles` contiguous ranges (the first nout % nhandles ranges one point longer), range r is
// evaluated by handles[r] on that handle's device from its own host thread.  The handles must
// describe the same interpolator (same method/kind/dtype/ndims); the grid was replicated when
// they were created.  No device-to-device traffic.
int interpn_hip_eval_host_sharded(interpn_hip_interp* const* handles, size_t nhandles, const void* const* obs,
                                  const size_t* obs_lens, size_t nobs, void* out, size_t nout,
                                  uint64_t* first_bad_index) {
  if (!handles || nhandles == 0 || nhandles > 1024 || (!obs && nobs) || (!obs_lens && nobs))
    return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  for (size_t r = 0; r < nhandles; ++r) {
    if (!handles[r]) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
    const GridDesc &a = handles[0]->desc, &b = handles[r]->desc;
    if (a.method != b.method || a.kind != b.kind || a.dtype != b.dtype || a.ndims != b.ndims)
      return INTERPN_HIP_ERR_INVALID_ARGUMENT;
    for (size_t q = 0; q < r; ++q)
      if (handles[q] == handles[r]) return INTERPN_HIP_ERR_INVALID_ARGUMENT;  // a handle has one workspace
  }
  int st = validate_obs(handles[0]->desc, obs_lens, nobs, nout);
  if (st) return st;
  if (nout == 0) return INTERPN_HIP_OK;
  if (!out) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < nobs; ++i)
    if (!obs[i]) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  const size_t elem = handles[0]->desc.dtype == kF64 ? 8 : 4;
  const size_t base = nout / nhandles, extra = nout % nhandles;
  std::vector<int> status(nhandles, INTERPN_HIP_OK);
  std::vector<size_t> bad(nhandles, 0), lo(nhandles, 0), cnt(nhandles, 0);
  std::vector<std::thread> workers;
  workers.reserve(nhandles);
  for (size_t r = 0; r < nhandles; ++r) {
    lo[r] = r * base + (r < extra ? r : extra);
    cnt[r] = base + (r < extra ? 1 : 0);
    if (cnt[r] == 0) continue;
    workers.emplace_back([&, r] {
      const void* sub[8];
      for (size_t d = 0; d < nobs; ++d) sub[d] = (const char*)obs[d] + lo[r] * elem;
      status[r] = eval_host_impl(handles[r], sub, nobs, (char*)out + lo[r] * elem, cnt[r], &bad[r]);
    });
  }
  for (auto& w : workers) w.join();
  // Ranges ascend with r, so the first failing range holds the globally first failing point.
  for (size_t r = 0; r < nhandles; ++r) {
    if (status[r] == INTERPN_HIP_ERR_UNREPRESENTABLE) {
      if (first_bad_index) *first_bad_index = (uint64_t)(lo[r] + bad[r]);
      return status[r];
    }
    if (status[r] != INTERPN_HIP_OK) return status[r];
  }
  return INTERPN_HIP_OK;
}

// One-shot entry points: `interpn(...)` = new(..)? then interp(obs, out) — the struct is rebuilt
// on every call in the reference as well (multilinear/regular.rs:65-71).
#define ONESHOT_TAIL(T)                                                                         \
  st = interpn_hip_eval_host(h, reinterpret_cast<const void* const*>(obs), obs_lens, nobs, out, nout); \
  interpn_hip_destroy(h);                                                                       \
  return st;

// Cheap checks first (grid validation, then the `.interp` length checks) so that a call that
// the reference rejects never touches the device; the order of the checks is the reference's.
#define ONESHOT_PRECHECK(METHOD, NDIMS, VALIDATE)                  \
  {                                                                \
    int pst = (VALIDATE);                                          \
    if (pst) return pst;                                           \
    GridDesc tmp;                                                  \
    tmp.method = (METHOD);                                         \
    tmp.ndims = (int)(NDIMS);                                      \
    if (!obs_lens && nobs) return INTERPN_HIP_ERR_INVALID_ARGUMENT; \
    pst = validate_obs(tmp, obs_lens, nobs, nout);                 \
    if (pst) return pst;                                           \
  }

#define DEFINE_ONESHOT(T, SUFFIX)                                                                             \
  int interpn_hip_linear_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts, size_t nstarts, \
                                          const T* steps, size_t nsteps, const T* vals, size_t nvals,        \
                                          const T* const* obs, const size_t* obs_lens, size_t nobs, T* out,  \
                                          size_t nout) {                                                     \
    /* multilinear/regular.rs:60 */                                                                           \
    if (nstarts != ndims || nsteps != ndims || nobs != ndims) return INTERPN_HIP_ERR_DIM_MISMATCH;           \
    ONESHOT_PRECHECK(kLinear, ndims, validate_regular<T>(kLinear, dims, ndims, starts, nstarts, steps, nsteps, nvals)) \
    interpn_hip_interp* h = nullptr;                                                                         \
    int st = create_regular<T>(kLinear, dims, ndims, starts, nstarts, steps, nsteps, vals, nvals,            \
                               INTERPN_HIP_MEM_HOST, 0, -1, &h);                                             \
    if (st) return st;                                                                                       \
    ONESHOT_TAIL(T)                                                                                          \
  }                                                                                                           \
  int interpn_hip_linear_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens, size_t ngrids, \
                                              const T* vals, size_t nvals, const T* const* obs,              \
                                              const size_t* obs_lens, size_t nobs, T* out, size_t nout) {    \
    /* multilinear/rectilinear.rs:59 */                                                                       \
    if (nobs != ngrids) return INTERPN_HIP_ERR_DIM_MISMATCH;                                                 \
    ONESHOT_PRECHECK(kLinear, ngrids, validate_rectilinear<T>(kLinear, grids, grid_lens, ngrids, nvals))     \
    interpn_hip_interp* h = nullptr;                                                                         \
    int st = create_rectilinear<T>(kLinear, grids, grid_lens, ngrids, vals, nvals, INTERPN_HIP_MEM_HOST, 0,  \
                                   -1, &h);                                                                  \
    if (st) return st;                                                                                       \
    ONESHOT_TAIL(T)                                                                                          \
  }                                                                                                           \
  int interpn_hip_cubic_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts, size_t nstarts,  \
                                         const T* steps, size_t nsteps, const T* vals, size_t nvals,         \
                                         int linearize_extrapolation, const T* const* obs,                   \
                                         const size_t* obs_lens, size_t nobs, T* out, size_t nout) {         \
    ONESHOT_PRECHECK(kCubic, ndims, validate_regular<T>(kCubic, dims, ndims, starts, nstarts, steps, nsteps, nvals)) \
    interpn_hip_interp* h = nullptr;                                                                         \
    int st = create_regular<T>(kCubic, dims, ndims, starts, nstarts, steps, nsteps, vals, nvals,             \
                               INTERPN_HIP_MEM_HOST, linearize_extrapolation, -1, &h);                       \
    if (st) return st;                                                                                       \
    ONESHOT_TAIL(T)                                                                                          \
  }                                                                                                           \
  int interpn_hip_cubic_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens, size_t ngrids,  \
                                             const T* vals, size_t nvals, int linearize_extrapolation,       \
                                             const T* const* obs, const size_t* obs_lens, size_t nobs,       \
                                             T* out, size_t nout) {                                          \
    ONESHOT_PRECHECK(kCubic, ngrids, validate_rectilinear<T>(kCubic, grids, grid_lens, ngrids, nvals))       \
    interpn_hip_interp* h = nullptr;                                                                         \
    int st = create_rectilinear<T>(kCubic, grids, grid_lens, ngrids, vals, nvals, INTERPN_HIP_MEM_HOST,      \
                                   linearize_extrapolation, -1, &h);                                         \
    if (st) return st;                                                                                       \
    ONESHOT_TAIL(T)                                                                                          \
  }
DEFINE_ONESHOT(double, f64)
DEFINE_ONESHOT(float, f32)

#define DEFINE_NEAREST(T, SUFFIX)                                                                             \
  int interpn_hip_nearest_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts, size_t nstarts, \
                                           const T* steps, size_t nsteps, const T* vals, size_t nvals,       \
                                           const T* const* obs, const size_t* obs_lens, size_t nobs, T* out, \
                                           size_t nout) {                                                    \
    /* nearest/regular.rs:50 */                                                                               \
    if (nstarts != ndims || nsteps != ndims || nobs != ndims) return INTERPN_HIP_ERR_DIM_MISMATCH;           \
    ONESHOT_PRECHECK(kNearest, ndims, validate_regular<T>(kNearest, dims, ndims, starts, nstarts, steps, nsteps, nvals)) \
    interpn_hip_interp* h = nullptr;                                                                         \
    int st = create_regular<T>(kNearest, dims, ndims, starts, nstarts, steps, nsteps, vals, nvals,           \
                               INTERPN_HIP_MEM_HOST, 0, -1, &h);                                             \
    if (st) return st;                                                                                       \
    ONESHOT_TAIL(T)                                                                                          \
  }                                                                                                           \
  int interpn_hip_nearest_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens, size_t ngrids, \
                                               const T* vals, size_t nvals, const T* const* obs,             \
                                               const size_t* obs_lens, size_t nobs, T* out, size_t nout) {   \
    /* nearest/rectilinear.rs:43 */                                                                           \
    if (nobs != ngrids) return INTERPN_HIP_ERR_DIM_MISMATCH;                                                 \
    ONESHOT_PRECHECK(kNearest, ngrids, validate_rectilinear<T>(kNearest, grids, grid_lens, ngrids, nvals))   \
    interpn_hip_interp* h = nullptr;                                                                         \
    int st = create_rectilinear<T>(kNearest, grids, grid_lens, ngrids, vals, nvals, INTERPN_HIP_MEM_HOST, 0, \
                                   -1, &h);                                                                  \
    if (st) return st;                                                                                       \
    ONESHOT_TAIL(T)                                                                                          \
  }
DEFINE_NEAREST(double, f64)
DEFINE_NEAREST(float, f32)

// check_bounds on device-resident coordinates, limits from the handle's grid.
int interpn_hip_check_bounds_device(interpn_hip_interp* h, const void* const* obs, size_t nobs, size_t npoints,
                                    double atol, uint8_t* out, size_t nout, void* stream) {
  if (!h || (!obs && nobs)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  const size_t ndims = (size_t)h->desc.ndims;
  if (!(nobs == ndims && nout == ndims)) return INTERPN_HIP_ERR_DIM_MISMATCH;  // regular.rs:153-156
  if (!out) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  for (size_t d = 0; d < nobs; ++d)
    if (!obs[d] && npoints) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned* flags = nullptr;
  HIP_TRY(pool_alloc(h->device, (void**)&flags, sizeof(unsigned) * 8));
  hipError_t e = hipMemsetAsync(flags, 0, sizeof(unsigned) * 8, s);
  for (size_t d = 0; d < ndims && e == hipSuccess; ++d) {
    if (h->desc.dtype == kF64)
      e = launch_check_bounds<double>(static_cast<const double*>(obs[d]), npoints, h->desc.bound_lo[d],
                                      h->desc.bound_hi[d], atol, flags + d, s);
    else
      e = launch_check_bounds<float>(static_cast<const float*>(obs[d]), npoints, (float)h->desc.bound_lo[d],
                                     (float)h->desc.bound_hi[d], (float)atol, flags + d, s);
  }
  unsigned host[8] = {0};
  if (e == hipSuccess) e = hipMemcpyAsync(host, flags, sizeof(host), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  else (void)hipStreamSynchronize(s);
  pool_free(h->device, flags);
  if (e != hipSuccess) return hip_fail(e);
  for (size_t d = 0; d < ndims; ++d) out[d] = host[d] ? 1 : 0;
  return INTERPN_HIP_OK;
}

#define DEFINE_BOUNDS(T, SUFFIX)                                                                              \
  int interpn_hip_check_bounds_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts,           \
                                                size_t nstarts, const T* steps, size_t nsteps,               \
                                                const T* const* obs, const size_t* obs_lens, size_t nobs,    \
                                                T atol, uint8_t* out, size_t nout) {                         \
    /* multilinear/regular.rs:153-156 */                                                                      \
    if (!(nobs == ndims && nout == ndims)) return INTERPN_HIP_ERR_DIM_MISMATCH;                              \
    /* starts[i] / steps[i] index out of range => panic in the reference */                                   \
    if (nstarts < ndims || nsteps < ndims) return INTERPN_HIP_ERR_REFERENCE_PANIC;                           \
    if (ndims && (!dims || !starts || !steps || !obs || !obs_lens || !out)) return INTERPN_HIP_ERR_INVALID_ARGUMENT; \
    std::vector<T> lo(ndims), hi(ndims);                                                                     \
    for (size_t i = 0; i < ndims; ++i) {                                                                     \
      if (dims[i] == 0) return INTERPN_HIP_ERR_REFERENCE_PANIC; /* dims[i] - 1 underflows */                 \
      const T first = starts[i];                                                                             \
      const T prod = steps[i] * (T)(dims[i] - 1);                                                            \
      const T last = starts[i] + prod; /* regular.rs:164, not fused */                                       \
      lo[i] = __builtin_fmin(first, last);                                                                   \
      hi[i] = __builtin_fmax(first, last);                                                                   \
    }                                                                                                        \
    return check_bounds_host<T>(lo.data(), hi.data(), ndims, obs, obs_lens, atol, out);                      \
  }                                                                                                           \
  int interpn_hip_check_bounds_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens,          \
                                                    size_t ngrids, const T* const* obs,                      \
                                                    const size_t* obs_lens, size_t nobs, T atol,             \
                                                    uint8_t* out, size_t nout) {                             \
    const size_t ndims = ngrids;                                                                             \
    if (ndims && (!grids || !grid_lens)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;                            \
    bool nonempty = true;                                                                                    \
    for (size_t i = 0; i < ndims; ++i) nonempty = nonempty && grid_lens[i] > 0;                              \
    /* multilinear/rectilinear.rs:115-118 */                                                                  \
    if (!(nobs == ndims && nout == ndims && nonempty)) return INTERPN_HIP_ERR_DIM_MISMATCH;                  \
    if (ndims && (!obs || !obs_lens || !out)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;                       \
    std::vector<T> lo(ndims), hi(ndims);                                                                     \
    for (size_t i = 0; i < ndims; ++i) {                                                                     \
      lo[i] = grids[i][0];                                                                                   \
      hi[i] = grids[i][grid_lens[i] - 1];                                                                    \
    }                                                                                                        \
    return check_bounds_host<T>(lo.data(), hi.data(), ndims, obs, obs_lens, atol, out);                      \
  }
DEFINE_BOUNDS(double, f64)
DEFINE_BOUNDS(float, f32)

}  // extern "C"
