// Host-side description of a device-resident interpolator and the launcher entry points.
// Internal to the library (the public surface is include/interpn_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

namespace interpn {

enum Method : int { kLinear = 0, kCubic = 1 };
enum Kind : int { kRegular = 0, kRectilinear = 1 };
enum DType : int { kF64 = 0, kF32 = 1 };

struct LaunchConfig {
  int num_cus = 256;       // MI355X: 8 XCDs x 32 CUs
  int blocks_per_cu = 8;   // 256-thread workgroups resident per CU that the grid is sized for
};

struct GridDesc {
  int method = kLinear;
  int kind = kRegular;
  int dtype = kF64;
  int ndims = 0;
  int linearize = 0;
  int fma = 1;                 // the reference's `fma` cargo feature (pyproject.toml:72 => on)
  int n[8] = {0};              // points per axis
  double start[8] = {0};       // regular grids; exactly representable in the element type
  double step[8] = {0};
  const void* vals = nullptr;  // device, C-ordered
  size_t nvals = 0;
  const void* grid[8] = {nullptr};  // device, rectilinear axes
  size_t grid_total = 0;            // sum of n[d]
  LaunchConfig cfg;
};

constexpr unsigned long long kNoBadIndexHost = ~0ull;  // value of the device first-bad-index word when no point failed

// LDS budget for rectilinear axes: keeps 8 workgroups per CU resident (160 KiB / 8).
constexpr size_t kMaxGridLdsBytes = 20 * 1024;

template <typename T>
hipError_t launch_linear_regular(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                                 unsigned long long* first_bad, hipStream_t stream);
template <typename T>
hipError_t launch_linear_rectilinear(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                                     unsigned long long* first_bad, hipStream_t stream);
template <typename T>
hipError_t launch_cubic_regular(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                                unsigned long long* first_bad, hipStream_t stream);
template <typename T>
hipError_t launch_cubic_rectilinear(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                                    unsigned long long* first_bad, hipStream_t stream);
template <typename T>
hipError_t launch_generic(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                          unsigned long long* first_bad, hipStream_t stream);

// check_bounds: OR into flag[0] whether any of x[0..n) violates [lo, hi] by atol or more
// (src/multilinear/regular.rs:168-171).
template <typename T>
hipError_t launch_check_bounds(const T* x, size_t n, T lo, T hi, T atol, unsigned* flag, hipStream_t stream);

// True when the templated (flattened-arm) kernels apply: N within the flattened range of the
// reference's dispatch and the grid indexable with 32 bits.
inline bool fast_path(const GridDesc& g) {
  const int maxn = g.method == kLinear ? 6 : 4;
  return g.ndims >= 1 && g.ndims <= maxn && g.nvals < 0xFFFFFFFFull;
}

}  // namespace interpn
