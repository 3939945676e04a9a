// Kernel dispatch of one evaluation, stream marks (what interpn_hip_destroy waits for), and the
// status-word wait.  (C ABI internals, see abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

namespace interpn_abi {

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

// Completion of an 8-byte device-to-pinned-host copy, by watching the landing word instead of
// calling hipStreamSynchronize: the caller stores kWordPending into *word, enqueues the copy on
// `s`, then calls this.  The copy is ordered behind everything enqueued on `s` before it, so once
// the word has changed that work is complete ON THE DEVICE: results in device memory may be used
// by anything enqueued afterwards.  It says nothing about data a kernel wrote into HOST memory
// (the zero-copy small-batch path keeps the runtime's wait for that reason).  Spinning on a
// pinned, host-coherent word costs a few microseconds less per call than the runtime's wait;
// after 200 us without an answer the runtime's wait takes over.
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

}  // namespace interpn_abi
