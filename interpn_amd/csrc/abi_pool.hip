// Per-device pool of device blocks, streams, pinned words and small-batch staging buffers; device
// properties; the thread's last HIP error.  (C ABI internals, see abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

namespace interpn_abi {

std::atomic<int> g_fma{1};
thread_local std::string t_last_hip_error;

int hip_fail(hipError_t e) {
  t_last_hip_error = hipGetErrorString(e);
  if (e == hipErrorOutOfMemory) return INTERPN_HIP_ERR_OUT_OF_MEMORY;
  if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return INTERPN_HIP_ERR_NO_DEVICE;
  return INTERPN_HIP_ERR_HIP;
}

bool checked_product(const size_t* dims, size_t n, size_t* out) {
  size_t acc = 1;
  for (size_t i = 0; i < n; ++i)
    if (__builtin_mul_overflow(acc, dims[i], &acc)) return false;  // Cargo.toml:45 overflow-checks => panic
  *out = acc;
  return true;
}

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
  // Blocks beyond the cap are not freed on the spot — hipFree waits for the WHOLE device, also for
  // streams that have nothing to do with the handle being destroyed — but parked here and freed in
  // one go by the next allocation that misses the pool once a quarter of the cap has collected
  // (or when an allocation fails, or at the cap itself).
  // A parked block is still a good block: an allocation of its size class takes it back.
  std::vector<std::pair<void*, size_t>> parked;  // block, class size
  size_t parked_bytes = 0;
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
    for (size_t k = pool.parked.size(); k-- > 0;)
      if (pool.parked[k].second == cls) {
        *out = pool.parked[k].first;
        pool.parked[k] = pool.parked.back();
        pool.parked.pop_back();
        pool.parked_bytes -= cls;
        pool.live[*out] = cls;
        return hipSuccess;
      }
  }
  {
    // an allocation that misses the pool is a heavyweight moment anyway (handle creation, scratch
    // growth): the place to give back what destroy calls have parked (hipFree waits for the device)
    std::vector<std::pair<void*, size_t>> drop;
    {
      std::lock_guard<std::mutex> lk(pool.mu);
      if (pool.parked_bytes > pool_cap_bytes() / 4) {
        drop.swap(pool.parked);
        pool.parked_bytes = 0;
      }
    }
    for (auto& b : drop) (void)hipFree(b.first);
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
      for (auto& b : pool.parked) drop.push_back(b.first);
      pool.parked.clear();
      pool.parked_bytes = 0;
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
  std::vector<void*> drop;  // freed below, outside the lock
  {
    std::lock_guard<std::mutex> lk(pool.mu);
    auto it = pool.live.find(p);
    if (it == pool.live.end()) {
      drop.push_back(p);  // not ours: plain hipFree
    } else {
      const size_t cls = it->second;
      pool.live.erase(it);
      if (pool.cached_bytes + cls <= pool_cap_bytes()) {
        pool.free_blocks[cls].push_back(p);
        pool.cached_bytes += cls;
      } else if (pool_cap_bytes() == 0) {
        drop.push_back(p);  // pooling switched off
      } else {  // over the cap: park it; the parked blocks are freed together now and then
        pool.parked.push_back({p, cls});
        pool.parked_bytes += cls;
        if (pool.parked_bytes > pool_cap_bytes()) {  // hard limit; normally the next allocation that misses the pool frees them
          for (auto& b : pool.parked) drop.push_back(b.first);
          pool.parked.clear();
          pool.parked_bytes = 0;
        }
      }
    }
  }
  for (void* b : drop) (void)hipFree(b);
}

// Give back every cached and parked block of a device (the current device must be `device`); returns the bytes freed.
size_t pool_trim(int device) {
  if (!pooled_device(device)) return 0;
  DevPool& pool = dev_pool(device);
  std::vector<void*> drop;
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lk(pool.mu);
    for (auto& kv : pool.free_blocks) {
      for (void* b : kv.second) drop.push_back(b);
      kv.second.clear();
    }
    bytes = pool.cached_bytes + pool.parked_bytes;
    pool.cached_bytes = 0;
    for (auto& b : pool.parked) drop.push_back(b.first);
    pool.parked.clear();
    pool.parked_bytes = 0;
  }
  for (void* b : drop) (void)hipFree(b);
  return bytes;
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

// CUs, L2 of one XCD, LDS per CU / per workgroup, XCDs: queried once per device.  Attributes the
// runtime does not report fall back to the MI355X values.
DeviceProps device_props(int device) {
  static std::mutex mu;
  static DeviceProps cached[kMaxPoolDevices];
  static bool have[kMaxPoolDevices] = {};
  const bool cacheable = device >= 0 && device < kMaxPoolDevices;
  if (cacheable) {
    std::lock_guard<std::mutex> lk(mu);
    if (have[device]) return cached[device];
  }
  DeviceProps p;
  auto attr = [&](hipDeviceAttribute_t a, int fallback) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, a, device) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = fallback; }
    return v;
  };
  p.num_cus = attr(hipDeviceAttributeMultiprocessorCount, 256);
  p.l2_bytes = attr(hipDeviceAttributeL2CacheSize, 4 << 20);
  // what a launch gets without hipFuncSetAttribute: the runtime's 64 KiB rule, or less on a smaller part
  // (the attribute reports the opt-in maximum: 160 KiB on MI355X)
  p.lds_per_wg = attr(hipDeviceAttributeMaxSharedMemoryPerBlock, 64 << 10);
  if (p.lds_per_wg > (64 << 10)) p.lds_per_wg = 64 << 10;
  p.lds_per_cu = attr(hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 160 << 10);
  if (p.lds_per_cu < p.lds_per_wg) p.lds_per_cu = p.lds_per_wg;
  // XCDs: gfx94x / gfx95x parts have 32 (MI300A: 38-CU XCDs, 228 CUs -> 6) CUs per L2 domain; the
  // runtime has no attribute for it.  The reported L2 size is that of ONE XCD on these parts; a
  // device that reports more than 16 MiB is taken to report the sum.
  p.num_xcds = p.num_cus >= 64 ? (p.num_cus + 37) / 38 : 1;
  if (p.num_cus % 32 == 0) p.num_xcds = p.num_cus / 32 > 0 ? p.num_cus / 32 : 1;
  if (p.l2_bytes > (16 << 20) && p.num_xcds > 1) p.l2_bytes /= p.num_xcds;
  if (cacheable) {
    std::lock_guard<std::mutex> lk(mu);
    cached[device] = p;
    have[device] = true;
  }
  return p;
}

int device_num_cus(int device) { return device_props(device).num_cus; }

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

}  // namespace interpn_abi
