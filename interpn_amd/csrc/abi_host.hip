// Host-pointer evaluation: the chunked double-lane pipeline, the zero-copy small-batch path,
// interpn_hip_finish, and the one-shot entry points that mirror the reference's `interpn(...)`
// functions.  (C ABI internals, see abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

namespace interpn_abi {

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

}  // namespace interpn_abi

namespace {

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

}  // namespace

namespace interpn_abi {

int eval_host_impl(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out, size_t nout,
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

}  // namespace interpn_abi

namespace {
// The status word to the host by a one-lane kernel instead of a copy-engine transfer (round 6): ordered behind the evaluation
// like any kernel on the stream, it stores the 8 bytes into the pinned word with ONE system-scope store (never torn), and
// costs a dispatch (~4 us) where the 8-byte hipMemcpyAsync costs ~10.  What the host may conclude from seeing the word is
// what it concluded from the copy: everything enqueued on the stream before it is complete on the device.
__global__ void k_status_to_host(const unsigned long long* dev_word, unsigned long long* host_word) {
  const unsigned long long v = __hip_atomic_load(dev_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(host_word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

extern "C" {

int interpn_hip_finish(interpn_hip_interp* h, void* stream, uint64_t* first_bad_index) {
  if (!h) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // The status word lands in pinned memory: a plain DMA behind the kernel, no staging copy.
  std::lock_guard<std::mutex> lk(h->finish_mu);
  if (!h->finish_word) {
    HIP_TRY(pool_take_pinned_word(h->device, &h->finish_word));
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, h->finish_word, 0) == hipSuccess) h->finish_word_dev = static_cast<unsigned long long*>(dp);
    else (void)hipGetLastError();
  }
  *(volatile unsigned long long*)h->finish_word = kWordPending;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool by_kernel = h->finish_word_dev && h->desc.cfg.finish_kernel != 0 &&
                         hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone;
  if (by_kernel) {
    hipLaunchKernelGGL(k_status_to_host, dim3(1), dim3(1), 0, s, (const unsigned long long*)h->first_bad, h->finish_word_dev);
    HIP_TRY(hipGetLastError());
  } else {
    (void)hipGetLastError();
    HIP_TRY(hipMemcpyAsync(h->finish_word, h->first_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  }
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

}  // extern "C"
