// Single-process multi-GPU forms of `.interp(obs, out)`: one handle per device, contiguous ranges
// of the observation index, the global first failing index.  (C ABI internals, see abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

#include <pthread.h>
#include <sched.h>

#include <fstream>
#include <sstream>

namespace {

// Run the calling host thread on the CPUs that are local to `device` (its PCIe root's NUMA node):
// hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<id>/local_cpulist.  The staging copies of a shard
// then come from memory and cores next to its GPU's link instead of crossing the socket
// interconnect.  Best effort: a missing sysfs entry, an empty list or a refused call change nothing.
bool pin_thread_near_device(int device) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return false; }
  std::string id(bus);
  for (char& c : id) c = (char)tolower((unsigned char)c);
  std::ifstream f("/sys/bus/pci/devices/" + id + "/local_cpulist");
  std::string list;
  if (!f || !std::getline(f, list) || list.empty()) return false;
  cpu_set_t set;
  CPU_ZERO(&set);
  int n = 0;
  std::stringstream ss(list);
  std::string item;
  while (std::getline(ss, item, ',')) {  // "0-47,96-143"
    int lo = 0, hi = 0;
    if (sscanf(item.c_str(), "%d-%d", &lo, &hi) == 2) { /* range */ }
    else if (sscanf(item.c_str(), "%d", &lo) == 1) hi = lo;
    else continue;
    for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c) { CPU_SET(c, &set); ++n; }
  }
  if (n == 0) return false;
  return pthread_setaffinity_np(pthread_self(), sizeof(set), &set) == 0;
}

}  // namespace

extern "C" {

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
  bool distinct_devices = false;
  for (size_t r = 1; r < nhandles; ++r) distinct_devices = distinct_devices || handles[r]->device != handles[0]->device;
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
      // a shard's host thread runs next to its GPU (only where the handles sit on different devices:
      // several handles on one device share its CPUs anyway)
      if (distinct_devices) (void)pin_thread_near_device(handles[r]->device);
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

// Device-resident single-process form: shard r = (obs[r][0..nobs), out[r], npoints[r]) lives on the
// device of handles[r] and is evaluated there on streams[r] (NULL array / entry = that device's
// default stream).  Every launch is enqueued before the first status word is waited for, so the
// devices run concurrently; no host or device-to-device traffic besides the 8-byte status words.
// The global first failing index counts points in shard order (shard 0's points first): the
// smallest over the shards of (points in front of shard r) + (its first failing point).
int interpn_hip_eval_device_sharded(interpn_hip_interp* const* handles, size_t nhandles, const void* const* const* obs,
                                    size_t nobs, void* const* out, const size_t* npoints, void* const* streams,
                                    uint64_t* first_bad_index) {
  if (!handles || nhandles == 0 || nhandles > 1024 || !obs || !out || !npoints) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  for (size_t r = 0; r < nhandles; ++r) {
    if (!handles[r] || (!obs[r] && nobs)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
    const GridDesc &a = handles[0]->desc, &b = handles[r]->desc;
    if (a.method != b.method || a.kind != b.kind || a.dtype != b.dtype || a.ndims != b.ndims)
      return INTERPN_HIP_ERR_INVALID_ARGUMENT;
    for (size_t q = 0; q < r; ++q)
      if (handles[q] == handles[r]) return INTERPN_HIP_ERR_INVALID_ARGUMENT;  // a handle has one status word
  }
  int first_error = INTERPN_HIP_OK;
  std::vector<int> launched(nhandles, 0);
  for (size_t r = 0; r < nhandles; ++r) {  // enqueue everything first
    const int st = interpn_hip_eval_device(handles[r], obs[r], nobs, out[r], npoints[r], streams ? streams[r] : nullptr);
    if (st != INTERPN_HIP_OK) { if (first_error == INTERPN_HIP_OK) first_error = st; continue; }
    launched[r] = 1;
  }
  uint64_t best = ~(uint64_t)0;
  size_t offset = 0;
  for (size_t r = 0; r < nhandles; ++r) {  // then wait for every shard that was enqueued (also after a failure)
    if (launched[r]) {
      uint64_t bad = 0;
      const int st = interpn_hip_finish(handles[r], streams ? streams[r] : nullptr, &bad);
      if (st == INTERPN_HIP_ERR_UNREPRESENTABLE) {
        if (offset + bad < best) best = offset + bad;
      } else if (st != INTERPN_HIP_OK && first_error == INTERPN_HIP_OK) {
        first_error = st;
      }
    }
    offset += npoints[r];
  }
  if (first_error != INTERPN_HIP_OK) return first_error;
  if (best != ~(uint64_t)0) {
    if (first_bad_index) *first_bad_index = best;
    return INTERPN_HIP_ERR_UNREPRESENTABLE;
  }
  return INTERPN_HIP_OK;
}

}  // extern "C"
