// Per-cell records of a rectilinear multicubic axis: what interpn_device.h::cubic_rect_dim_setup computes from the cell alone
// (multicubic/rectilinear.rs:413-545 with the non-uniform central difference of mod.rs:103-117), stored per cell k of the axis
//   k = 0: the first interval and everything below it, class Low (footprint g[0..3]);
//   k = n - 2: the last interval and everything above, class High (footprint g[n-4..n-1]);
//   else class None with the footprint starting at k - 1
// and built once at creation on the host with the same operations in the same type (IEEE divisions, no contraction: the
// library is compiled with -ffp-contract=off), so a kernel that reads a record holds the bits the seven divisions of the
// per-point setup would have produced.  t itself is sign * (x - gref) / h, taken with quotient_fast from rh.
#pragma once

#include <stddef.h>

#include <vector>

namespace interpn {

template <typename T>
struct CubicCellRecord {
  T gref, h, rh;        // t's reference coordinate, divisor and RN(1 / h)
  T r0, a0, c0, rr0;    // first central difference (CubicDimRect) and RN(1 / r0)
  T r1, a1, c1, rr1;    // second one (class None; else 1)
  T fast;               // 1: h, r0 and r1 are divisors the short form takes (interpn_device.h::fast_divisor); 0: they are not
};
static_assert(sizeof(CubicCellRecord<double>) == 96 && sizeof(CubicCellRecord<float>) == 48, "twelve values, 16-byte pieces");

// the host's fast_divisor: positive, 2^-128 <= b < 2^128 (f32: 2^-16 <= b < 2^16)
template <typename T>
inline bool cubic_record_divisor_ok(T b) {
  const double lo = sizeof(T) == 8 ? 0x1p-128 : 0x1p-16, hi = sizeof(T) == 8 ? 0x1p128 : 0x1p16;
  return (double)b >= lo && (double)b < hi;  // (NaN: false)
}

template <typename T>
inline void build_cubic_cell_records(const T* g, int n, std::vector<CubicCellRecord<T>>& out) {
  out.assign((size_t)(n - 1), CubicCellRecord<T>());
  const T one = (T)1;
  for (int k = 0; k <= n - 2; ++k) {
    CubicCellRecord<T>& r = out[(size_t)k];
    r.r1 = one; r.a1 = one; r.c1 = one;
    if (k == 0) {  // Low
      const T g0 = g[0], g1 = g[1], g2 = g[2];
      const T h01 = g1 - g0, h12 = g2 - g1;
      r.r0 = h12 / h01;
      r.a0 = one / (one + r.r0);
      r.c0 = r.r0 / (r.r0 + one);
      r.gref = g1;
      r.h = h01;
    } else if (k == n - 2) {  // High
      const T g1 = g[n - 3], g2 = g[n - 2], g3 = g[n - 1];
      const T h12 = g2 - g1, h23 = g3 - g2;
      r.r0 = h12 / h23;
      r.a0 = r.r0 / (r.r0 + one);
      r.c0 = one / (one + r.r0);
      r.gref = g2;
      r.h = h23;
    } else {  // None
      const T g0 = g[k - 1], g1 = g[k], g2 = g[k + 1], g3 = g[k + 2];
      const T h01 = g1 - g0, h12 = g2 - g1, h23 = g3 - g2;
      r.r0 = h01 / h12;
      r.a0 = r.r0 / (r.r0 + one);
      r.c0 = one / (one + r.r0);
      r.r1 = h23 / h12;
      r.a1 = one / (one + r.r1);
      r.c1 = r.r1 / (r.r1 + one);
      r.gref = g1;
      r.h = h12;
    }
    r.rh = one / r.h;
    r.rr0 = one / r.r0;
    r.rr1 = one / r.r1;
    r.fast = (cubic_record_divisor_ok(r.h) && cubic_record_divisor_ok(r.r0) && cubic_record_divisor_ok(r.r1)) ? one : (T)0;
  }
}

}  // namespace interpn
