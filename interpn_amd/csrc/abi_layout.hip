// Which re-laid copy of the grid a handle keeps: bricked tables for the multilinear kernels, tiled
// tables for the multicubic kernels, per-bucket records for 1-D rectilinear axes.  (C ABI
// internals, see abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

namespace interpn_abi {

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
    double hit = thresholds(g.cfg).table_l2_model / (double)bytes;  // share of the 4 MiB L2 the table keeps beside the non-temporal streams
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
      (bytes11 <= thresholds(g.cfg).binned_table_min || g.ndims == 2 || g.kind == kRectilinear))
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
  // 3-D grids whose dims 0, 1 give at most 4096 class pairs keep it as well when the sorted evaluation is asked for at
  // creation (INTERPN_HIP_BINNED=1; cubic3_column.h: not taken by itself, profiles/REJECTED.md round 5; 64^3 f64: 33.5 MiB)
  const bool column3 = g.ndims == 3 && (long long)(g.n[0] - 1) * (g.n[1] - 1) <= kMaxBins && g.cfg.column != 0 && g.cfg.binned == 1;
  // 3-D grids keep it for the sweep evaluation of large batches too (cubic_sweep.h; option sweep = 0 at creation: not built)
  // — where the automatic rule can take it (k_cubic_sweep.hip::cubic_sweep_applies: regular grids, the table between the L2's size
  // and 128 MiB, f32 32 MiB), or anywhere when the sweep is forced at creation (INTERPN_HIP_SWEEP=1)
  const bool sweep3 = g.ndims == 3 && (g.cfg.sweep > 0 ||
                                       (g.cfg.sweep < 0 && g.kind == kRegular && bytes11 > thresholds(g.cfg).table_l2_sized &&
                                        bytes11 <= (g.dtype == kF64 ? (size_t)128 << 20 : (size_t)32 << 20)));
  if (!forced && (g.ndims == 4 || column3 || sweep3) && best != 4 && fits11) {
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
  if (g.axis_image_bytes <= thresholds(g.cfg).axis_lds_wide && !(env && !strcmp(env, "on"))) return INTERPN_HIP_OK;
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
    const size_t l2_sized = thresholds(g.cfg).table_l2_sized, l2_share = thresholds(g.cfg).table_l2_share;  // 6 MiB, 3 MiB on MI355X
    if (g.ndims >= 4 && fits(bcell) && (bcell <= l2_sized || b22 > l2_share)) {
      cell = true;
    } else if (fits(bone)) {
      if (bone <= l2_sized || b22 > l2_sized) { si = 1; sj = 1; j4 = f32; }
      else if (b12 <= l2_sized) { si = 1; sj = 2; }
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
  // The table the sweep evaluation of large device-resident batches runs on (linear_sweep.h): the
  // one just built where the layouts agree, else a second one (64^3 f64: 10.2 MiB beside 5.2).
  g.sweep_bricks = nullptr;
  int wi = 0, wj = 0, wcell = 0;
  if (!(env && !strcmp(env, "off")) && sweep_layout(g, &wi, &wj, &wcell)) {
    unsigned nbs[3];
    size_t bsw = 0;
    if (wcell == 2) brick_j4_geometry(g, nbs, &bsw);
    else brick_geometry(g, wi, wj, nbs, &bsw);
    if (g.brick_cell == wcell && g.brick_step[0] == wi && g.brick_step[1] == wj) {
      g.sweep_bricks = g.bricks;
    } else if (bsw / esz < 0xFFFFFFFFull && bsw <= free_b / 4 && pool_alloc(h->device, &h->sweep_owned, bsw) == hipSuccess) {
      GridDesc t = g;
      t.brick_cell = wcell;
      t.brick_step[0] = wi;
      t.brick_step[1] = wj;
      for (int k = 0; k < 3; ++k) t.brick_nb[k] = nbs[k];
      t.brick_nb[3] = 0;
      // the second table is optional: a failure to build it leaves the handle on its brick kernel
      if (build_bricks(t, h->sweep_owned, nullptr) == hipSuccess && hipStreamSynchronize(nullptr) == hipSuccess) {
        g.sweep_bricks = h->sweep_owned;
      } else {
        (void)hipGetLastError();
        pool_free(h->device, h->sweep_owned);
        h->sweep_owned = nullptr;
      }
    } else {
      (void)hipGetLastError();
      h->sweep_owned = nullptr;
    }
    if (g.sweep_bricks) {
      g.sweep_step[0] = wi;
      g.sweep_step[1] = wj;
      g.sweep_cell = wcell;
      for (int k = 0; k < 3; ++k) g.sweep_nb[k] = nbs[k];
      g.sweep_table_bytes = bsw;
    }
  }
  return INTERPN_HIP_OK;
}

}  // namespace interpn_abi
