// interpn_hip.hpp — C++17 host-side mirror of the reference crate's Rust API over the C ABI
// (include/interpn_hip.h, libinterpn_hip.so).  Header-only; needs no HIP header and no hipcc.
//
// The reference is a Rust crate (jlogan03/interpn v0.8.2) and no Rust toolchain exists in this
// image, so the compiled-language host side above the C ABI is written in C++ with the crate's
// own module paths, function names, argument order and error strings; INTEGRATION.md shows the
// Rust `hip_sys` binding of the same entry points.  What maps to what:
//
//   Rust (reference, file:line)                                   C++ (this header)
//   interpn::multilinear::regular::interpn      regular.rs:51     interpn_hip::multilinear::regular::interpn
//   ...::interpn_alloc                          regular.rs:124    ...::interpn_alloc
//   ...::check_bounds                           regular.rs:145    ...::check_bounds
//   MultilinearRegular<'a, T, N>::new           regular.rs:225    MultilinearRegular<T, N>::new_   (`new` is a C++ keyword)
//   MultilinearRegular::interp / interp_one     regular.rs:268 / :296   ::interp / ::interp_one
//   interpn::multilinear::rectilinear::*        rectilinear.rs:49,90,109,175,210,244
//   interpn::multicubic::regular::*             multicubic/regular.rs:52,143,165,239,297,325   (7-argument interpn: linearize_extrapolation)
//   interpn::multicubic::rectilinear::*         multicubic/rectilinear.rs:54,111,123,193,237,265
//   interpn::nearest::{regular,rectilinear}::*  nearest/regular.rs:41,108,163,206,234; rectilinear.rs:39,73,124,159,193
//   interpn::utils::{linspace, meshgrid}        utils.rs:8,17     interpn_hip::utils::{linspace, meshgrid}
//
//   &[T] -> Slice<T> (pointer + length, implicit from std::vector / std::array / C arrays),
//   &mut [T] -> SliceMut<T>, &[&[T]] -> Slice<Slice<T>>,
//   Result<(), &'static str> -> Result<void>, Result<T, &'static str> -> Result<T>: `.is_ok()`,
//   `.is_err()`, `.err()` (the reference's message, byte for byte), `.unwrap()` (throws
//   std::runtime_error with that message where Rust would panic), `.status()` (the ABI code).
//
// Differences a maintainer should know: the interpolator structs own a device-resident copy of
// the grid (the Rust structs borrow host slices), so they are move-only RAII handles and the
// `vals` / `grids` slices may be freed after `new_`; where the reference panics (slice -> array
// `try_into().unwrap()` in the multicubic dispatch, multicubic/regular.rs:66-73) the Result
// carries status INTERPN_HIP_ERR_REFERENCE_PANIC instead of aborting the process; evaluation
// needs a HIP device (there is no CPU path: `.err()` is then "no usable HIP device").
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>

#include "interpn_hip.h"

namespace interpn_hip {

// ---------------------------------------------------------------------------------------------
// &[T] / &mut [T]
template <class T>
struct Slice {
  const T* ptr = nullptr;
  std::size_t len_ = 0;
  constexpr Slice() = default;
  constexpr Slice(const T* p, std::size_t n) : ptr(p), len_(n) {}
  Slice(const std::vector<T>& v) : ptr(v.data()), len_(v.size()) {}
  template <std::size_t K> constexpr Slice(const std::array<T, K>& a) : ptr(a.data()), len_(K) {}
  template <std::size_t K> constexpr Slice(const T (&a)[K]) : ptr(a), len_(K) {}
  constexpr std::size_t len() const { return len_; }
  constexpr bool is_empty() const { return len_ == 0; }
  constexpr const T& operator[](std::size_t i) const { return ptr[i]; }
  constexpr const T* begin() const { return ptr; }
  constexpr const T* end() const { return ptr + len_; }
};

template <class T>
struct SliceMut {
  T* ptr = nullptr;
  std::size_t len_ = 0;
  constexpr SliceMut() = default;
  constexpr SliceMut(T* p, std::size_t n) : ptr(p), len_(n) {}
  SliceMut(std::vector<T>& v) : ptr(v.data()), len_(v.size()) {}
  template <std::size_t K> constexpr SliceMut(std::array<T, K>& a) : ptr(a.data()), len_(K) {}
  template <std::size_t K> constexpr SliceMut(T (&a)[K]) : ptr(a), len_(K) {}
  constexpr std::size_t len() const { return len_; }
  constexpr T& operator[](std::size_t i) const { return ptr[i]; }
};

// `&[&[T]]` from a vector of vectors (what the reference's tests build with `.iter().map(|x| &x[..])`).
template <class T>
inline std::vector<Slice<T>> slices(const std::vector<std::vector<T>>& v) {
  std::vector<Slice<T>> s;
  s.reserve(v.size());
  for (const auto& x : v) s.emplace_back(x);
  return s;
}

// ---------------------------------------------------------------------------------------------
// Result<T, &'static str>
class ResultBase {
 public:
  bool is_ok() const { return status_ == INTERPN_HIP_OK; }
  bool is_err() const { return !is_ok(); }
  int status() const { return status_; }
  // The reference's `&'static str` (interpn_hip_strerror returns the identical text); "" when Ok.
  const char* err() const { return is_ok() ? "" : interpn_hip_strerror(status_); }

 protected:
  explicit ResultBase(int status) : status_(status) {}
  void check() const {
    if (is_err()) throw std::runtime_error(err());  // Rust: panic on `.unwrap()` of an Err
  }
  int status_;
};

template <class T>
class Result : public ResultBase {
 public:
  static Result Ok(T v) { return Result(INTERPN_HIP_OK, std::move(v)); }
  static Result Err(int status) { return Result(status, T()); }
  T& unwrap() & { check(); return value_; }
  T&& unwrap() && { check(); return std::move(value_); }
  T unwrap_or(T fallback) && { return is_ok() ? std::move(value_) : std::move(fallback); }

 private:
  Result(int status, T v) : ResultBase(status), value_(std::move(v)) {}
  T value_;
};

template <>
class Result<void> : public ResultBase {
 public:
  static Result Ok() { return Result(INTERPN_HIP_OK); }
  static Result Err(int status) { return Result(status); }
  static Result from(int status) { return Result(status); }
  void unwrap() const { check(); }

 private:
  explicit Result(int status) : ResultBase(status) {}
};

// ---------------------------------------------------------------------------------------------
namespace detail {

enum Method : int { kLinear = INTERPN_HIP_LINEAR, kCubic = INTERPN_HIP_CUBIC, kNearest = INTERPN_HIP_NEAREST };

template <class T> struct Abi;  // the _f64 / _f32 entry points of include/interpn_hip.h
#define INTERPN_HIP_HPP_ABI(T, S)                                                                                         \
  template <> struct Abi<T> {                                                                                             \
    static constexpr auto linear_regular = &interpn_hip_linear_regular_##S;                                               \
    static constexpr auto linear_rectilinear = &interpn_hip_linear_rectilinear_##S;                                       \
    static constexpr auto cubic_regular = &interpn_hip_cubic_regular_##S;                                                 \
    static constexpr auto cubic_rectilinear = &interpn_hip_cubic_rectilinear_##S;                                         \
    static constexpr auto nearest_regular = &interpn_hip_nearest_regular_##S;                                             \
    static constexpr auto nearest_rectilinear = &interpn_hip_nearest_rectilinear_##S;                                     \
    static constexpr auto create_regular = &interpn_hip_create_regular_##S;                                               \
    static constexpr auto create_rectilinear = &interpn_hip_create_rectilinear_##S;                                       \
    static constexpr auto bounds_regular = &interpn_hip_check_bounds_regular_##S;                                         \
    static constexpr auto bounds_rectilinear = &interpn_hip_check_bounds_rectilinear_##S;                                 \
  };
INTERPN_HIP_HPP_ABI(double, f64)
INTERPN_HIP_HPP_ABI(float, f32)
#undef INTERPN_HIP_HPP_ABI

// &[&[T]] -> (pointer array, length array)
template <class T>
struct Unpacked {
  std::vector<const T*> ptrs;
  std::vector<std::size_t> lens;
  explicit Unpacked(Slice<Slice<T>> s) {
    ptrs.reserve(s.len());
    lens.reserve(s.len());
    for (const auto& x : s) {
      ptrs.push_back(x.ptr);
      lens.push_back(x.len());
    }
  }
  template <std::size_t N>
  explicit Unpacked(const std::array<Slice<T>, N>& s) {
    for (const auto& x : s) {
      ptrs.push_back(x.ptr);
      lens.push_back(x.len());
    }
  }
};

template <class T>
inline Result<void> bounds_out(int st, const std::vector<std::uint8_t>& flags, SliceMut<bool> out) {
  if (st == INTERPN_HIP_OK)
    for (std::size_t i = 0; i < out.len(); ++i) out[i] = flags[i] != 0;
  return Result<void>::from(st);
}

template <class T>
inline Result<void> check_bounds_regular(Slice<std::size_t> dims, Slice<T> starts, Slice<T> steps, Slice<Slice<T>> obs, T atol,
                                         SliceMut<bool> out) {
  Unpacked<T> o(obs);
  std::vector<std::uint8_t> flags(out.len(), 0);
  const int st = Abi<T>::bounds_regular(dims.ptr, dims.len(), starts.ptr, starts.len(), steps.ptr, steps.len(), o.ptrs.data(),
                                        o.lens.data(), o.ptrs.size(), atol, flags.data(), flags.size());
  return bounds_out<T>(st, flags, out);
}

template <class T>
inline Result<void> check_bounds_rectilinear(Slice<Slice<T>> grids, Slice<Slice<T>> obs, T atol, SliceMut<bool> out) {
  Unpacked<T> g(grids), o(obs);
  std::vector<std::uint8_t> flags(out.len(), 0);
  const int st = Abi<T>::bounds_rectilinear(g.ptrs.data(), g.lens.data(), g.ptrs.size(), o.ptrs.data(), o.lens.data(),
                                            o.ptrs.size(), atol, flags.data(), flags.size());
  return bounds_out<T>(st, flags, out);
}

// Persistent interpolator: the device-resident counterpart of the reference's borrowed-slice
// structs.  Move-only; the destructor releases the device copy of the grid.
template <class T, std::size_t N>
class Handle {
 public:
  Handle() = default;
  Handle(const Handle&) = delete;
  Handle& operator=(const Handle&) = delete;
  Handle(Handle&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  Handle& operator=(Handle&& o) noexcept {
    if (this != &o) {
      reset();
      h_ = o.h_;
      o.h_ = nullptr;
    }
    return *this;
  }
  ~Handle() { reset(); }

  // `interp`: x = &[&[T]; N], out = &mut [T]; lengths are checked like the reference's
  // ("Dimension mismatch"), the batch aborts at the first point whose coordinate cannot be
  // converted to an index with out[0..i) written ("Unrepresentable coordinate value").
  Result<void> interp(const std::array<Slice<T>, N>& x, SliceMut<T> out) const {
    Unpacked<T> o(x);
    return Result<void>::from(interpn_hip_eval_host(h_, reinterpret_cast<const void* const*>(o.ptrs.data()), o.lens.data(), N,
                                                    out.ptr, out.len()));
  }

  // `interp_one`: one point (a 1-point batch on the device: about 25 us; batch with `interp`).
  Result<T> interp_one(std::array<T, N> x) const {
    const void* ptrs[N];
    std::size_t lens[N];
    for (std::size_t d = 0; d < N; ++d) {
      ptrs[d] = &x[d];
      lens[d] = 1;
    }
    T out = T();
    const int st = interpn_hip_eval_host(h_, ptrs, lens, N, &out, 1);
    return st == INTERPN_HIP_OK ? Result<T>::Ok(out) : Result<T>::Err(st);
  }

  // Escape hatch to the C ABI (device-pointer evaluation on a stream, options, kernel name ...).
  interpn_hip_interp* raw() const { return h_; }

  // Adopts (and will destroy) a handle obtained from the C ABI.
  explicit Handle(interpn_hip_interp* h) : h_(h) {}

 protected:
  void reset() {
    if (h_) interpn_hip_destroy(h_);
    h_ = nullptr;
  }
  interpn_hip_interp* h_ = nullptr;
};

#if defined(INTERPN_HIP_FEATURE_FMA)
constexpr int kFlavour = INTERPN_HIP_FEATURE_FMA ? INTERPN_HIP_FLAVOUR_FMA : INTERPN_HIP_FLAVOUR_NO_FMA;
// The one-shot `interpn` functions have the crate's signatures — no place for a flavour — and
// follow the process default: a program built with the feature macro sets it once, at start-up
// (a cargo feature is a whole-program choice too).
inline const int kFeatureFmaApplied = (interpn_hip_set_fma(INTERPN_HIP_FEATURE_FMA ? 1 : 0), 0);
#else
constexpr int kFlavour = 0;  // the process default (interpn_hip_set_fma)
#endif

template <class Derived, class T, std::size_t N>
inline Result<Derived> make_regular(int method, const std::array<std::size_t, N>& dims, const std::array<T, N>& starts,
                                    const std::array<T, N>& steps, Slice<T> vals, bool linearize, int device) {
  interpn_hip_interp* h = nullptr;
  const int st = Abi<T>::create_regular(method | kFlavour, dims.data(), N, starts.data(), N, steps.data(), N, vals.ptr, vals.len(),
                                        INTERPN_HIP_MEM_HOST, linearize ? 1 : 0, device, &h);
  return st == INTERPN_HIP_OK ? Result<Derived>::Ok(Derived(h)) : Result<Derived>::Err(st);
}

template <class Derived, class T, std::size_t N>
inline Result<Derived> make_rectilinear(int method, const std::array<Slice<T>, N>& grids, Slice<T> vals, bool linearize, int device) {
  Unpacked<T> g(grids);
  interpn_hip_interp* h = nullptr;
  const int st = Abi<T>::create_rectilinear(method | kFlavour, g.ptrs.data(), g.lens.data(), N, vals.ptr, vals.len(), INTERPN_HIP_MEM_HOST,
                                            linearize ? 1 : 0, device, &h);
  return st == INTERPN_HIP_OK ? Result<Derived>::Ok(Derived(h)) : Result<Derived>::Err(st);
}

template <class T, class F>
inline Result<std::vector<T>> alloc_then(Slice<Slice<T>> obs, F&& eval) {
  // Rust indexes obs[0] and panics on an empty `obs` (regular.rs:131)
  if (obs.is_empty()) return Result<std::vector<T>>::Err(INTERPN_HIP_ERR_REFERENCE_PANIC);
  std::vector<T> out(obs[0].len(), T(0));
  Result<void> r = eval(SliceMut<T>(out));
  return r.is_ok() ? Result<std::vector<T>>::Ok(std::move(out)) : Result<std::vector<T>>::Err(r.status());
}

}  // namespace detail

// ---------------------------------------------------------------------------------------------
namespace multilinear {

namespace regular {

// src/multilinear/regular.rs:51-117.  N = dims.len() from 1 to 8.
template <class T>
inline Result<void> interpn(Slice<std::size_t> dims, Slice<T> starts, Slice<T> steps, Slice<T> vals, Slice<Slice<T>> obs,
                            SliceMut<T> out) {
  detail::Unpacked<T> o(obs);
  return Result<void>::from(detail::Abi<T>::linear_regular(dims.ptr, dims.len(), starts.ptr, starts.len(), steps.ptr, steps.len(),
                                                           vals.ptr, vals.len(), o.ptrs.data(), o.lens.data(), o.ptrs.size(),
                                                           out.ptr, out.len()));
}

// regular.rs:124-134
template <class T>
inline Result<std::vector<T>> interpn_alloc(Slice<std::size_t> dims, Slice<T> starts, Slice<T> steps, Slice<T> vals,
                                            Slice<Slice<T>> obs) {
  return detail::alloc_then<T>(obs, [&](SliceMut<T> out) { return interpn<T>(dims, starts, steps, vals, obs, out); });
}

// regular.rs:145-182
template <class T>
inline Result<void> check_bounds(Slice<std::size_t> dims, Slice<T> starts, Slice<T> steps, Slice<Slice<T>> obs, T atol,
                                 SliceMut<bool> out) {
  return detail::check_bounds_regular<T>(dims, starts, steps, obs, atol, out);
}

// regular.rs:200-425
template <class T, std::size_t N>
class MultilinearRegular : public detail::Handle<T, N> {
 public:
  MultilinearRegular() = default;
  static Result<MultilinearRegular> new_(std::array<std::size_t, N> dims, std::array<T, N> starts, std::array<T, N> steps,
                                         Slice<T> vals, int device = -1) {
    return detail::make_regular<MultilinearRegular, T, N>(detail::kLinear, dims, starts, steps, vals, false, device);
  }

  using detail::Handle<T, N>::Handle;  // adopt a raw ABI handle
};

}  // namespace regular

namespace rectilinear {

// src/multilinear/rectilinear.rs:49-83
template <class T>
inline Result<void> interpn(Slice<Slice<T>> grids, Slice<T> vals, Slice<Slice<T>> obs, SliceMut<T> out) {
  detail::Unpacked<T> g(grids), o(obs);
  return Result<void>::from(detail::Abi<T>::linear_rectilinear(g.ptrs.data(), g.lens.data(), g.ptrs.size(), vals.ptr, vals.len(),
                                                               o.ptrs.data(), o.lens.data(), o.ptrs.size(), out.ptr, out.len()));
}

// rectilinear.rs:90-99
template <class T>
inline Result<std::vector<T>> interpn_alloc(Slice<Slice<T>> grids, Slice<T> vals, Slice<Slice<T>> obs) {
  return detail::alloc_then<T>(obs, [&](SliceMut<T> out) { return interpn<T>(grids, vals, obs, out); });
}

// rectilinear.rs:109-134
template <class T>
inline Result<void> check_bounds(Slice<Slice<T>> grids, Slice<Slice<T>> obs, T atol, SliceMut<bool> out) {
  return detail::check_bounds_rectilinear<T>(grids, obs, atol, out);
}

// rectilinear.rs:153-370
template <class T, std::size_t N>
class MultilinearRectilinear : public detail::Handle<T, N> {
 public:
  MultilinearRectilinear() = default;
  static Result<MultilinearRectilinear> new_(const std::array<Slice<T>, N>& grids, Slice<T> vals, int device = -1) {
    return detail::make_rectilinear<MultilinearRectilinear, T, N>(detail::kLinear, grids, vals, false, device);
  }

  using detail::Handle<T, N>::Handle;  // adopt a raw ABI handle
};

}  // namespace rectilinear

using regular::MultilinearRegular;          // src/multilinear/mod.rs:9-11
using rectilinear::MultilinearRectilinear;

}  // namespace multilinear

// ---------------------------------------------------------------------------------------------
namespace multicubic {

namespace regular {

// src/multicubic/regular.rs:52-136 (7 arguments: linearize_extrapolation sits in front of obs)
template <class T>
inline Result<void> interpn(Slice<std::size_t> dims, Slice<T> starts, Slice<T> steps, Slice<T> vals, bool linearize_extrapolation,
                            Slice<Slice<T>> obs, SliceMut<T> out) {
  detail::Unpacked<T> o(obs);
  return Result<void>::from(detail::Abi<T>::cubic_regular(dims.ptr, dims.len(), starts.ptr, starts.len(), steps.ptr, steps.len(),
                                                          vals.ptr, vals.len(), linearize_extrapolation ? 1 : 0, o.ptrs.data(),
                                                          o.lens.data(), o.ptrs.size(), out.ptr, out.len()));
}

// multicubic/regular.rs:143-163
template <class T>
inline Result<std::vector<T>> interpn_alloc(Slice<std::size_t> dims, Slice<T> starts, Slice<T> steps, Slice<T> vals,
                                            bool linearize_extrapolation, Slice<Slice<T>> obs) {
  return detail::alloc_then<T>(
      obs, [&](SliceMut<T> out) { return interpn<T>(dims, starts, steps, vals, linearize_extrapolation, obs, out); });
}

// multicubic/regular.rs:165 re-exports the multilinear check
using multilinear::regular::check_bounds;

// multicubic/regular.rs:211-469
template <class T, std::size_t N>
class MulticubicRegular : public detail::Handle<T, N> {
 public:
  MulticubicRegular() = default;
  static Result<MulticubicRegular> new_(std::array<std::size_t, N> dims, std::array<T, N> starts, std::array<T, N> steps,
                                        Slice<T> vals, bool linearize_extrapolation, int device = -1) {
    return detail::make_regular<MulticubicRegular, T, N>(detail::kCubic, dims, starts, steps, vals, linearize_extrapolation, device);
  }

  using detail::Handle<T, N>::Handle;  // adopt a raw ABI handle
};

}  // namespace regular

namespace rectilinear {

// src/multicubic/rectilinear.rs:54-104
template <class T>
inline Result<void> interpn(Slice<Slice<T>> grids, Slice<T> vals, bool linearize_extrapolation, Slice<Slice<T>> obs,
                            SliceMut<T> out) {
  detail::Unpacked<T> g(grids), o(obs);
  return Result<void>::from(detail::Abi<T>::cubic_rectilinear(g.ptrs.data(), g.lens.data(), g.ptrs.size(), vals.ptr, vals.len(),
                                                              linearize_extrapolation ? 1 : 0, o.ptrs.data(), o.lens.data(),
                                                              o.ptrs.size(), out.ptr, out.len()));
}

// multicubic/rectilinear.rs:111-121
template <class T>
inline Result<std::vector<T>> interpn_alloc(Slice<Slice<T>> grids, Slice<T> vals, bool linearize_extrapolation,
                                            Slice<Slice<T>> obs) {
  return detail::alloc_then<T>(obs, [&](SliceMut<T> out) { return interpn<T>(grids, vals, linearize_extrapolation, obs, out); });
}

using multilinear::rectilinear::check_bounds;  // multicubic/rectilinear.rs:123

// multicubic/rectilinear.rs:168-408
template <class T, std::size_t N>
class MulticubicRectilinear : public detail::Handle<T, N> {
 public:
  MulticubicRectilinear() = default;
  static Result<MulticubicRectilinear> new_(const std::array<Slice<T>, N>& grids, Slice<T> vals, bool linearize_extrapolation,
                                            int device = -1) {
    return detail::make_rectilinear<MulticubicRectilinear, T, N>(detail::kCubic, grids, vals, linearize_extrapolation, device);
  }

  using detail::Handle<T, N>::Handle;  // adopt a raw ABI handle
};

}  // namespace rectilinear

using regular::MulticubicRegular;            // src/multicubic/mod.rs:54-56
using rectilinear::MulticubicRectilinear;

}  // namespace multicubic

// ---------------------------------------------------------------------------------------------
namespace nearest {

namespace regular {

// src/nearest/regular.rs:41-101.  N from 1 to 6.
template <class T>
inline Result<void> interpn(Slice<std::size_t> dims, Slice<T> starts, Slice<T> steps, Slice<T> vals, Slice<Slice<T>> obs,
                            SliceMut<T> out) {
  detail::Unpacked<T> o(obs);
  return Result<void>::from(detail::Abi<T>::nearest_regular(dims.ptr, dims.len(), starts.ptr, starts.len(), steps.ptr, steps.len(),
                                                            vals.ptr, vals.len(), o.ptrs.data(), o.lens.data(), o.ptrs.size(),
                                                            out.ptr, out.len()));
}

// nearest/regular.rs:108-118
template <class T>
inline Result<std::vector<T>> interpn_alloc(Slice<std::size_t> dims, Slice<T> starts, Slice<T> steps, Slice<T> vals,
                                            Slice<Slice<T>> obs) {
  return detail::alloc_then<T>(obs, [&](SliceMut<T> out) { return interpn<T>(dims, starts, steps, vals, obs, out); });
}

using multilinear::regular::check_bounds;  // nearest/regular.rs:120

// nearest/regular.rs:138-320
template <class T, std::size_t N>
class NearestRegular : public detail::Handle<T, N> {
 public:
  NearestRegular() = default;
  static Result<NearestRegular> new_(std::array<std::size_t, N> dims, std::array<T, N> starts, std::array<T, N> steps,
                                     Slice<T> vals, int device = -1) {
    return detail::make_regular<NearestRegular, T, N>(detail::kNearest, dims, starts, steps, vals, false, device);
  }

  using detail::Handle<T, N>::Handle;  // adopt a raw ABI handle
};

}  // namespace regular

namespace rectilinear {

// src/nearest/rectilinear.rs:39-66
template <class T>
inline Result<void> interpn(Slice<Slice<T>> grids, Slice<T> vals, Slice<Slice<T>> obs, SliceMut<T> out) {
  detail::Unpacked<T> g(grids), o(obs);
  return Result<void>::from(detail::Abi<T>::nearest_rectilinear(g.ptrs.data(), g.lens.data(), g.ptrs.size(), vals.ptr, vals.len(),
                                                                o.ptrs.data(), o.lens.data(), o.ptrs.size(), out.ptr, out.len()));
}

// nearest/rectilinear.rs:73-81
template <class T>
inline Result<std::vector<T>> interpn_alloc(Slice<Slice<T>> grids, Slice<T> vals, Slice<Slice<T>> obs) {
  return detail::alloc_then<T>(obs, [&](SliceMut<T> out) { return interpn<T>(grids, vals, obs, out); });
}

using multilinear::rectilinear::check_bounds;  // nearest/rectilinear.rs:83

// nearest/rectilinear.rs:102-265
template <class T, std::size_t N>
class NearestRectilinear : public detail::Handle<T, N> {
 public:
  NearestRectilinear() = default;
  static Result<NearestRectilinear> new_(const std::array<Slice<T>, N>& grids, Slice<T> vals, int device = -1) {
    return detail::make_rectilinear<NearestRectilinear, T, N>(detail::kNearest, grids, vals, false, device);
  }

  using detail::Handle<T, N>::Handle;  // adopt a raw ABI handle
};

}  // namespace rectilinear

using regular::NearestRegular;               // src/nearest/mod.rs:7-8
using rectilinear::NearestRectilinear;

}  // namespace nearest

// src/lib.rs:94-100 re-exports
using multilinear::MultilinearRectilinear;
using multilinear::MultilinearRegular;
using multicubic::MulticubicRectilinear;
using multicubic::MulticubicRegular;
using nearest::NearestRectilinear;
using nearest::NearestRegular;

// ---------------------------------------------------------------------------------------------
namespace utils {

// src/utils.rs:8-14: start + i * dx with dx = (stop - start) / (n - 1) — NOT numpy's formula; the
// reference's Rust tests build their grids with it, so the mirrored tests do too.
template <class T>
inline std::vector<T> linspace(T start, T stop, std::size_t n) {
  const T dx = (stop - start) / static_cast<T>(n - 1);
  std::vector<T> v(n);
  for (std::size_t i = 0; i < n; ++i) v[i] = start + static_cast<T>(i) * dx;
  return v;
}

// src/utils.rs:17-25: every combination in C order (the last axis varies fastest), one
// N-vector per grid point.
template <class T>
inline std::vector<std::vector<T>> meshgrid(const std::vector<const std::vector<T>*>& x) {
  std::size_t total = 1;
  for (const auto* a : x) total *= a->size();
  std::vector<std::vector<T>> pts(total, std::vector<T>(x.size()));
  for (std::size_t p = 0; p < total; ++p) {
    std::size_t rem = p;
    for (std::size_t d = x.size(); d-- > 0;) {
      pts[p][d] = (*x[d])[rem % x[d]->size()];
      rem /= x[d]->size();
    }
  }
  return pts;
}

}  // namespace utils

// The `fma` cargo feature (Cargo.toml:35; on in the published wheels).  Like the crate's it is
// chosen at COMPILE time, per translation unit: define INTERPN_HIP_FEATURE_FMA as 1 or 0 before
// including this header and every interpolator struct created through it carries that flavour
// (a per-handle property of the C ABI: INTERPN_HIP_FLAVOUR_*).  Left undefined, the structs follow
// the process default, which `set_fma` changes (also what the one-shot `interpn` functions use).
inline void set_fma(bool enabled) { interpn_hip_set_fma(enabled ? 1 : 0); }

}  // namespace interpn_hip
