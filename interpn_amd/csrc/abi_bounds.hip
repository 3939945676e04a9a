// check_bounds (multilinear/regular.rs:145-182, rectilinear.rs:109-134): host arrays streamed
// through the device, and device arrays with the limits from a handle.  (C ABI internals, see
// abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

namespace {

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

}  // namespace

extern "C" {

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
