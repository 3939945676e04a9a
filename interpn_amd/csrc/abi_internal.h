// Internals shared by the translation units of the C ABI (include/interpn_hip.h): the handle, the
// per-device pool, validation in the reference's order, and the entry points the units call in each
// other.  Host logic only.  Split (round 4) from one 2 200-line file into
//   abi_pool.hip      per-device pool of blocks, streams, pinned words; device properties
//   abi_layout.hip    which re-laid copy of the grid a handle keeps (bricks, tiles, 1-D records)
//   abi_options.hip   per-handle options, kernel name, table size
//   abi_create.hip    create / replicate / destroy, status strings
//   abi_launch.hip    kernel dispatch of one evaluation, stream marks, status-word wait
//   abi_binned.hip    sorted (binned / column) evaluation of device-resident batches
//   abi_host.hip      host-pointer pipeline, small-batch path, finish, one-shot entry points
//   abi_bounds.hip    check_bounds
//   abi_sharded.hip   single-process multi-GPU forms
#pragma once
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

namespace interpn_abi {

using namespace interpn;

extern std::atomic<int> g_fma;                        // process default of the `fma` flavour (interpn_hip_set_fma)
extern thread_local std::string t_last_hip_error;

int hip_fail(hipError_t e);

#define HIP_TRY(expr)                        \
  do {                                       \
    hipError_t _e = (expr);                  \
    if (_e != hipSuccess) return hip_fail(_e); \
  } while (0)

// Axis length limits of the device kernels (cell indices are 32-bit; f32 classifies cells by
// comparing against (float)n, exact only up to 2^24).
template <typename T> constexpr size_t max_axis_len() { return sizeof(T) == 8 ? (size_t)2147483391u : (size_t)16777216u; }

bool checked_product(const size_t* dims, size_t n, size_t* out);

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

// ---- per-device pool (abi_pool.hip)
hipError_t pool_alloc(int device, void** out, size_t bytes);  // the current device must be `device`
void pool_free(int device, void* p);                          // only for blocks no in-flight work still touches
size_t pool_trim(int device);                                 // frees what the pool holds for `device` (current device = `device`)
hipError_t pool_take_kit(int device, hipStream_t* stream, unsigned long long** flag_host);
void pool_return_kit(int device, hipStream_t stream, unsigned long long* flag_host);
hipError_t pool_take_pinned_word(int device, unsigned long long** word);
void pool_return_pinned_word(int device, unsigned long long* word);
// Staging of the small-batch host path: pinned host memory the kernel reads and writes directly
// over PCIe (zero-copy).  One fixed size serves every interpolator: 8 coordinate arrays + 1 result
// array of kSmallPoints f64 elements.
constexpr size_t kSmallPoints = 8192;
constexpr size_t kSmallBytes = (size_t)(8 + 1) * kSmallPoints * 8;

hipError_t pool_take_small(int device, void** buf);
void pool_return_small(int device, void* buf);
struct DeviceProps { int num_cus = 256, num_xcds = 8; long long l2_bytes = 4 << 20, lds_per_cu = 160 << 10, lds_per_wg = 64 << 10; };
DeviceProps device_props(int device);
int device_num_cus(int device);

}  // namespace interpn_abi

struct interpn_hip_interp {
  interpn::GridDesc desc;
  int device = 0;
  void* vals_owned = nullptr;   // device copy of vals when created from host memory
  void* grids_owned = nullptr;  // one device allocation holding all rectilinear axes
  void* bricks_owned = nullptr; // bricked copy of vals (3-D multilinear f64)
  void* bricks11_owned = nullptr;  // 4-D multicubic: fully overlapped tiles for binned evaluation when `bricks` is another layout
  void* sweep_owned = nullptr;     // 3-D f64 multilinear: the sweep evaluation's table when `bricks` is another layout (desc.sweep_bricks)
  unsigned long long* first_bad = nullptr;  // device word, ~0 = no failure
  unsigned long long* finish_word = nullptr;  // pinned landing word of interpn_hip_finish
  unsigned long long* finish_word_dev = nullptr;  // ... as the device sees it (the status kernel's target), or null
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
    // the block's first sweep_work_bytes() are a SweepWork a complete sweep launch left behind (round counters zero, the
    // measured period kept for the next launch).  The two invariants exclude each other: the period word lies inside the
    // sort's histogram (totals[291]), so whichever path uses a block clears the other path's flag.
    bool sweep_clean = false;
    hipEvent_t stage[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // option stage_timing: start | hist | scan | scatter | kernel
    bool staged = false;           // the last use recorded them (single slice)
  };
  std::mutex bin_mu;
  std::vector<BinSlot> bin_slots;  // capacity kMaxBinSlots from the start: evaluations hold pointers to elements outside the lock
  unsigned long long bin_uses = 0;
  interpn_hip_interp() { bin_slots.reserve(kMaxBinSlots); }
  std::atomic<long long> evals_binned{0}, evals_in_place{0}, evals_sweep{0}, scratch_allocs{0};
  // thinning the sample out (option sweep_probe = 2; guarded by bin_mu): the sampling kernel also stores (seq << 1 | coherent)
  // into this pinned word; the host looks at it before the next launch — no synchronisation, a verdict that has not landed
  // yet simply is not known yet
  unsigned long long* probe_host = nullptr;
  unsigned* probe_host_dev = nullptr;
  unsigned probe_seq = 0, probe_seen = 0;
  int probe_streak = 0;     // samples in a row that came out unordered
  int probe_streak_coherent = 0;  // ... that came out coherent
  int probe_skipped = 0;    // automatic launches since the last sample
  const void* last_probe_word = nullptr;  // device word holding the verdict of the most recent gated launch's sampling kernel (option "sweep_probe_took_brick"; tests, bench)
};

namespace interpn_abi {

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
inline int validate_obs(const GridDesc& g, const size_t* obs_lens, size_t nobs, size_t nout) {
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

int resolve_device(int device, int* out);

// abi_layout.hip
int maybe_build_bricks(interpn_hip_interp* h);

// abi_options.hip
bool option_access(LaunchConfig& c, const char* name, long long* value, bool set);
void latch_env(LaunchConfig& c);

// abi_create.hip
int finish_create(interpn_hip_interp* h, const void* vals, size_t nvals, size_t elem, int vals_mem);
template <typename T>
int create_regular(int method, const size_t* dims, size_t ndims, const T* starts, size_t nstarts, const T* steps,
                   size_t nsteps, const T* vals, size_t nvals, int vals_mem, int linearize, int device,
                   interpn_hip_interp** handle);
template <typename T>
int create_rectilinear(int method, const T* const* grids, const size_t* grid_lens, size_t ngrids, const T* vals,
                       size_t nvals, int vals_mem, int linearize, int device, interpn_hip_interp** handle);

// abi_launch.hip
hipError_t launch_any(const GridDesc& g, const void* const* obs, void* out, size_t npts,
                      unsigned long long* first_bad, hipStream_t stream);
void mark_stream(interpn_hip_interp* h, hipStream_t stream);
// Chunk size of the host pipeline (points).  Bounded so that the workspace stays modest
// (8 dims x 8 B x 4 Mi = 256 MiB worst case) while each kernel launch still fills the chip.
constexpr size_t kHostChunkPoints = (size_t)4 << 20;

constexpr unsigned long long kWordPending = 0xFFFFFFFFFFFFFFFEull;  // neither "no failure" (~0) nor an index
hipError_t wait_status_word(hipStream_t s, const unsigned long long* word);

// abi_binned.hip
int binned_applies(const GridDesc& g, size_t npoints);
interpn_hip_interp::BinSlot* take_bin_slot(interpn_hip_interp* h, size_t need, hipStream_t stream, bool may_alloc, int* why);

// abi_sweep.hip
int eval_device_sweep(interpn_hip_interp* h, const void* const* obs, void* out, size_t npoints, hipStream_t stream,
                      unsigned flags, int* why);

// abi_host.hip
int ensure_lane(interpn_hip_interp* h, int which, size_t points);
int eval_host_impl(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out, size_t nout, size_t* bad_index);

}  // namespace interpn_abi
