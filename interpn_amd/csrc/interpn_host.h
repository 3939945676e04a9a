// Host-side description of a device-resident interpolator and the launcher entry points.
// Internal to the library (the public surface is include/interpn_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

#include <initializer_list>

namespace interpn {

enum Method : int { kLinear = 0, kCubic = 1, kNearest = 2 };
enum Kind : int { kRegular = 0, kRectilinear = 1 };
enum DType : int { kF64 = 0, kF32 = 1 };

// Per-handle launch options.  Defaults are latched from the environment ONCE, when the handle is
// created (abi_options.hip::latch_env); interpn_hip_set_option changes them afterwards.  Nothing on
// the launch path reads the environment.
struct LaunchConfig {
  int num_cus = 256;       // MI355X: 8 XCDs x 32 CUs (queried per device at creation, like the next three)
  int num_xcds = 8;        // L2 domains of the device
  long long l2_bytes = 4ll << 20;        // L2 of one XCD
  long long lds_per_cu = 160ll << 10;    // LDS of a CU
  long long lds_per_wg = 64ll << 10;     // LDS a workgroup gets without opt-in
  int blocks_per_cu = 8;   // 256-thread workgroups resident per CU that a persistent grid is sized for
  int iters_per_block = 0; // brick kernels: 256-wide iterations per workgroup (0 = the kernel's default)
  int ppl = 0;             // multilinear brick kernels: points per lane (0 = auto, 1 = scalar streams)
  int axis_regs = -1;      // rectilinear brick kernels: -1 auto, 0 axes in LDS, 1 lanes + probe sequence, 2 lanes + lane table
  int force_generic = 0;   // route every evaluation through the runtime-N kernel (testing)
  int generic_runtime = 0; // recursive arms: keep the runtime-N form (testing)
  int generic_vec = -1;    // recursive arms: -1 auto, 0 one-tree form, 1 row-vector form where it is compiled
  int persistent = 0;      // C-order regular / nearest kernels: persistent grid instead of one pass
  int axis_lds_kb = -1;    // LDS budget for the rectilinear axis image in KiB (-1 = the kernel's default)
  long long host_chunk = 0;  // points per chunk of the host-pointer pipeline (0 = default)
  int deal = 1;            // binned evaluation: deal the sorted points out to the XCDs (cubic_brick.h `eighth`)
  int binned = -1;         // tiled multicubic, device-pointer evaluation: -1 auto, 0 never, 1 always sort the points first
  int column = -1;         // binned 4-D multicubic on regular grids: evaluate sorted points out of an LDS-resident table column (-1/1 where it applies, 0 never)
  int column_part = 0;     // column evaluation: points per workgroup (0 = automatic)
  int axis_records = 1;    // rectilinear multilinear / nearest: search with per-bucket records where the handle has them (0: coordinates + tables as before)
  int stage_timing = 0;    // binned evaluation: record HIP events between its launches (interpn_hip_stage_ms; bench.py)
  int bin_scramble = 0;    // testing: the sort misplaces every 5th point by one bin (results must not change: exercises the column kernel's out-of-cell path)
  int column_threads = 768; // column evaluation: threads of the persistent workgroup (768 = 12 waves = three per SIMD at 168 VGPRs; 384 and 256: tests)
  int column_groups = 1;    // column evaluation: independent wave groups inside the workgroup (768 threads: 1 or 2; measured on cfg4: 0.74 ms with one group, 0.79 with two — the set-up two groups hide from each other costs less than the coarser end of the launch; cubic_column.h)
  int scatter_staged = 1;   // column evaluation's sort: stage a chunk's records in LDS in bin order and copy them out linearly (1) or store every record directly (0)
  int column_coef = 1;      // column evaluation: dim 0 by per-part Hermite coefficients (cubic_column.h "Coefficient columns"; 0: every node from the table values, the round-3/4 form)
  int column_tail = 0x84;   // column evaluation: the last 1 / (v >> 4) of the bins are cut (v & 15) times finer, so that the launch ends in small pieces (0: all bins whole)
  int column_keys = 1;      // column evaluation on regular grids: the sort leaves every point's local-sort key (8 bits) in the upper bits of its index word, so that the column kernel's local sort reads 4 bytes per point instead of the 32-byte record (0: keys from the records, 10 bits)
  int hist_wgs_per_cu = 4;  // the sort's histogram kernel: persistent workgroups per CU, each flushing its counters once (one global atomic per bin and workgroup): cfg4 0.048 -> 0.041 ms (0: one workgroup per 8192-point chunk)
  int column_pad = -1;      // column evaluation: LDS tiles 16 bytes apart (1), bare (0), or bare where that saves phases (-1)
  int column_cpp = 0;       // column evaluation: classes of dim 2 per K-range phase at most (0 = as many as the LDS share holds; tests force several phases on small grids)
  long long debug_stamps = 0;  // measurement aid: device address of 8 x u64 per part of the column kernel for in-kernel time stamps (0 = off)
  long long debug_stamps_bytes = 0;  // ... and the size of that buffer: a launch whose parts need more than it holds writes no stamps
  int bin_slice_log2 = 25; // binned evaluation: log2 of the points sorted and evaluated per slice (bounds a scratch block)
  int sweep = -1;          // 3-D f64 multilinear, device-pointer evaluation: the sweep kernel (linear_sweep.h) -1 where it pays, 0 never, 1 whenever the handle has its table
  int finish_kernel = 0;   // interpn_hip_finish: 1 = the status word reaches the host by a one-lane kernel instead of an 8-byte copy (3.5 us less per call: 21.4 -> 17.8 us for a 1e3-point device call).  Off by default: the host then sees the word a few microseconds before the HIP runtime has retired the commands in front of it, and hipEventElapsedTime on an event recorded before finish() can still say "not ready" (observed once in ~1e4 calls); callers that only consume results or synchronise their events themselves can switch it on
  int gated_iters = 4;     // rows of 256 lanes per workgroup of the gated brick launch behind an automatic sweep launch (an empty workgroup costs dispatch time)
  int sweep_probe = 2;     // automatic sweep launches: sample the batch on the device first and let the one-pass kernel take coherent batches — 0: never (the sweep kernel whatever the points look like), 1: every launch, 2: every launch until three samples in a row came out unordered, then every 16th (abi_sweep.hip)
  // (sweep itself: -1 automatic, 0 never, 1 always, 2 always with the sample deciding between the two kernels)
  int sweep_period = 0;    // sweep evaluation: ticks of 10 ns per sweep of the leading index (0: what the previous launch measured; 1: no clock, rows in sorted order; tests / tuning)
};

// What the most recent launch through a handle ran: the kernel template and its arguments in
// template order after the element type (reported by interpn_hip_kernel_name in the spelling
// rocprofv3 prints).  Written by the launchers without locking: concurrent evaluations on one
// handle run the same kernel, so a torn write cannot mix two different answers in practice.
struct KernelTag {
  const char* name = nullptr;  // static string, e.g. "k_linear_brick"
  int nargs = 0;
  int args[12] = {0};
  unsigned bool_mask = 0;      // bit k set: args[k] is a bool template parameter
  void set(const char* nm, std::initializer_list<int> a, unsigned bools) {
    name = nm;
    nargs = 0;
    for (int v : a)
      if (nargs < 12) args[nargs++] = v;
    bool_mask = bools;
  }
};

struct GridDesc {
  // Per-LAUNCH settings (set on a private copy of the handle's description by abi_sweep.hip, never on the handle's own):
  // an automatic sweep launch is a sampling kernel + the sweep kernel + the one-pass kernel, the latter two gated by the
  // sample's verdict word in device memory (k_linear_sweep.hip::k_sweep_probe).
  const unsigned* launch_gate = nullptr;  // one-pass kernels: do nothing unless this word is non-zero (few, fat workgroups)
  bool sweep_gated = false;               // sweep kernels: do nothing if the scratch block's verdict word is set
  bool launch_fat = false;                // one-pass kernels: the gated launch's workgroup shape without a gate (a handle whose last samples all said "coherent")
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
  // Rectilinear axes as one device image: per axis the coordinates (8-byte aligned) followed by
  // the bucket table ((M+1) x u32), see interpn_device.h::Axis.  Staged into LDS when small.
  const void* axis_image = nullptr;
  unsigned axis_image_bytes = 0;   // coordinates + tables only: what the kernels stage when they search without records
  unsigned axis_alloc_bytes = 0;   // the whole allocation: image + records region
  unsigned axis_g_off[8] = {0};    // byte offsets inside the image
  unsigned axis_tab_off[8] = {0};
  int axis_buckets[8] = {0};       // M per axis, 0 = no table
  double axis_g0[8] = {0};
  double axis_scale[8] = {0};
  // Lane table of an axis with at most 64 coordinates (sorted, finite): 255 buckets, the count of
  // coordinates in front of each bucket packed four bytes per word (64 words: one per lane),
  // followed by one word holding the largest bucket population.  0 = none.
  unsigned axis_ltab_off[8] = {0};
  double axis_lscale[8] = {0};
  int axis_lscan[8] = {0};  // largest bucket population of the lane table (read back when it is built)
  // Per-bucket search records (multilinear / nearest; axes too long for the lane-resident search):
  // when no bucket of an axis' table holds more than one coordinate, one record per bucket —
  // {g[k-1], g[k], g[k+1], k} with k = tab[b] — answers a cell query with ONE LDS access instead of
  // the table's two words, a scan probe and the two bracketing coordinates (interpn_device.h::
  // axis_cell).  They sit behind the coordinates and tables in the axis image and are staged
  // INSTEAD of them.  axis_rec_bytes = 0: none (some axis is not eligible, or every axis fits a lane).
  unsigned axis_rec_off[8] = {0};   // byte offsets inside the allocation
  unsigned axis_rec_base = 0;       // first byte of the records region (behind the image)
  unsigned axis_rec_bytes = 0;      // its size
  // Compact form (interpn_device.h::AxisRecordC), built when the full records exceed the LDS
  // budget of the kernel that will run (20 KiB for N >= 3, 60 KiB for N <= 2) but {g[k], k} plus
  // a copy of the coordinates fit: the region then holds, per axis, the coordinates
  // (axis_recg_off) followed by the records (axis_rec_off).
  int axis_rec_compact = 0;
  unsigned axis_recg_off[8] = {0};
  // Per-cell records of a rectilinear multicubic handle (interpn_device.h::CubicCellRecord): what a point's cell fixes of
  // cubic_rect_dim_setup — spacing ratios, central-difference weights, their reciprocals — computed once at creation with the
  // divisions the kernels would do per point.  n - 1 records per axis behind the image; 0 bytes = none.
  unsigned axis_crec_off[8] = {0};
  unsigned axis_crec_bytes = 0;
  // Optional bricked copy of `vals` (multilinear, 3 <= N <= 6; see k_linear_brick.hip): the last
  // three dims in 2 x 2 x KW bricks of one 128-B line, steps (brick_step[0], brick_step[1], KW-1).
  // brick_cell = 1 (N >= 4): 2 x 2 x 2 x KW bricks over the last FOUR dims instead (a whole 4-D cell
  // per line); brick_nb[3] then counts the bricks along dimension N-4.
  const void* bricks = nullptr;
  // 1-D multilinear on a rectilinear axis: `bricks` holds one record per search bucket
  // (k_linear1_records.hip) when rec1_buckets != 0.
  int rec1_buckets = 0;
  double rec1_scale = 0.0;
  int brick_step[2] = {2, 2};
  unsigned brick_nb[4] = {0, 0, 0, 0};
  // 4-D multicubic whose `bricks` is not the fully overlapped layout: a second, fully overlapped
  // tile table that large batches are evaluated on after sorting (binned evaluation); else null.
  const void* bricks11 = nullptr;
  unsigned bricks11_nb[2] = {0, 0};
  int brick_cell = 0;
  // 3-D multilinear: the table the sweep evaluation of large device-resident batches runs on
  // (linear_sweep.h; layout sweep_step, see k_linear_sweep.hip::sweep_layout) — `bricks` itself where
  // the layouts agree, else a second table the handle owns.  nullptr: the sweep never applies.
  const void* sweep_bricks = nullptr;
  int sweep_step[2] = {0, 0};
  int sweep_cell = 0;  // 0: 2 x 2 x KW bricks stepped sweep_step; 2 (f32): 2 x 4 x 4 bricks
  unsigned sweep_nb[3] = {0, 0, 0};
  size_t sweep_table_bytes = 0;
  // check_bounds limits per dimension, in the element type's arithmetic
  // (multilinear/regular.rs:160-166: starts + steps*(dims-1), min/max; rectilinear.rs:121-123).
  double bound_lo[8] = {0};
  double bound_hi[8] = {0};
  LaunchConfig cfg;
  mutable KernelTag tag;
  mutable int last_binned = 0;  // the most recent device-pointer evaluation sorted its points first (binned evaluation)
};

// Brick kernels cover the batch with ONE pass of small workgroups (each owning `iters` consecutive
// 256-lane rows) instead of a persistent grid-stride loop: the dispatcher then balances the XCDs
// dynamically (measured on 1e8 points, 64^3: 1.36 -> 1.29 ms).  Rectilinear kernels stage their
// axes per workgroup and want a few rows each to amortise that.
// `setup`: 0 = none (regular grids), 1 = per-wave register loads of lane-resident axes (6 small
// loads per wave: 4 rows amortise them, cfg3 1.39 -> 1.30..1.36 ms, 4-D rectilinear 2.26 -> 2.14),
// 2 = per-workgroup LDS staging of the axis image (8 rows).
inline unsigned brick_iters(const GridDesc& g, size_t npts, int points_per_lane, int setup) {
  unsigned iters = g.cfg.iters_per_block > 0 ? (unsigned)g.cfg.iters_per_block : (setup == 2 ? 8u : (setup == 1 ? 4u : 1u));
  // keep the grid below 2^23 workgroups (the dispatch packet counts work-items in 32 bits)
  const size_t rows = (npts + (size_t)256 * points_per_lane - 1) / ((size_t)256 * points_per_lane);
  while ((rows + iters - 1) / iters > (1u << 23)) iters *= 2;
  return iters;
}

constexpr unsigned long long kNoBadIndexHost = ~0ull;  // value of the device first-bad-index word when no point failed

// LDS budget for rectilinear axes: keeps 8 workgroups per CU resident (160 KiB / 8).
constexpr size_t kMaxGridLdsBytes = 20 * 1024;
// ... or most of the 64 KiB a workgroup gets without opt-in, for 1-D / 2-D kernels with no other LDS use.
constexpr size_t kMaxGridLdsBytesWide = 60 * 1024;

// The tuning thresholds as functions of the device (round 4; on MI355X they evaluate to the
// constants the measurements of DESIGN.md were taken with — tests/test_gpu_parity.py asserts it):
// an XCD's L2 (4 MiB), the LDS of a CU (160 KiB) and of a workgroup without opt-in (64 KiB), the CUs.
struct Thresholds {
  size_t table_l2_sized;      // a re-laid table up to this size is "L2-sized": 1.5 x L2 (6 MiB; the L2 keeps ~3.3 MiB of it beside the streams, the rest hits often enough)
  size_t table_l2_share;      // what an L2 holds of a table beside the streams: 0.75 x L2 (3 MiB)
  double table_l2_model;      // the same in the multicubic tile-layout model: 0.875 x L2 (3.5 MiB)
  size_t binned_table_min;    // sorting pays only for tables beyond 2 x L2 (8 MiB)
  size_t binned_points_min;   // ... and batches of at least 2048 points per CU (2^19)
  size_t bin_table_share;     // the sort's bins are sized to L2 / 8 of the table (512 KiB)
  size_t axis_lds;            // rectilinear axes in LDS: LDS per CU / 8 (20 KiB: 8 workgroups stay resident)
  size_t axis_lds_wide;       // ... 1-D / 2-D kernels: workgroup LDS without opt-in - 4 KiB (60 KiB)
  size_t column_lds;          // the column kernel's workgroup: the whole LDS of a CU (160 KiB)
};
inline Thresholds thresholds(const LaunchConfig& c) {
  Thresholds t;
  const size_t l2 = (size_t)c.l2_bytes;
  t.table_l2_sized = l2 + l2 / 2;
  t.table_l2_share = l2 - l2 / 4;
  t.table_l2_model = 0.875 * (double)l2;
  t.binned_table_min = 2 * l2;
  t.binned_points_min = (size_t)2048 * (size_t)c.num_cus;
  t.bin_table_share = l2 / 8;
  t.axis_lds = (size_t)c.lds_per_cu / 8;
  t.axis_lds_wide = (size_t)c.lds_per_wg - 4096;
  t.column_lds = (size_t)c.lds_per_cu;
  return t;
}

// Sweep evaluation (k_linear_sweep.hip)
bool sweep_layout(const GridDesc& g, int* si, int* sj, int* cell);
int sweep_applies(const GridDesc& g, size_t npts);
size_t sweep_work_bytes();
int nearest_sweep_applies(const GridDesc& g, size_t npts);  // k_nearest.hip (2-D / 3-D nearest neighbour, regular grids)
hipError_t launch_nearest_sweep(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad,
                                void* work, hipStream_t stream);
int linear2_sweep_applies(const GridDesc& g, size_t npts);  // k_linear2_brick.hip (2-D multilinear, regular grids)
hipError_t launch_linear2_sweep(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad,
                                void* work, hipStream_t stream);
int cubic_sweep_applies(const GridDesc& g, size_t npts);   // k_cubic_sweep.hip (3-D multicubic; sweep_applies / launch_linear_sweep hand cubic handles over)
hipError_t launch_cubic_sweep(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad,
                              void* work, hipStream_t stream);
hipError_t launch_linear_sweep(const GridDesc& g, const void* const* obs, void* out, size_t npts, unsigned long long* first_bad,
                               void* work, hipStream_t stream);
// Automatic sweep launches of the handles sweep_probe_applies() names: the sampling kernel whose verdict (a word of the
// scratch block, at this offset) gates the sweep launch (GridDesc::sweep_gated) and the one-pass launch
// (GridDesc::launch_gate) enqueued behind it
bool sweep_probe_applies(const GridDesc& g);
hipError_t launch_sweep_probe(const GridDesc& g, const void* const* obs, size_t npts, void* work, hipStream_t stream,
                              unsigned* host_word = nullptr, unsigned seq = 0);
size_t sweep_probe_word_offset();

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
// nearest-neighbour (src/nearest/*.rs), N = 1..6
template <typename T>
hipError_t launch_nearest(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                          unsigned long long* first_bad, hipStream_t stream);
template <typename T>
hipError_t launch_generic(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                          unsigned long long* first_bad, hipStream_t stream);

// Bricked multilinear path (k_linear_brick.hip).
void brick_geometry(const GridDesc& g, int si, int sj, unsigned nb[3], size_t* bytes);
void brick_cell_geometry(const GridDesc& g, unsigned nb[4], size_t* bytes);
void brick_j4_geometry(const GridDesc& g, unsigned nb[3], size_t* bytes);  // f32 2 x 4 x 4 bricks (brick_cell == 2)
hipError_t build_bricks(const GridDesc& g, void* bricks, hipStream_t stream);
template <typename T>
hipError_t launch_linear_brick(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                               unsigned long long* first_bad, hipStream_t stream);

// 1-D multilinear-rectilinear from per-bucket records (k_linear1_records.hip).
size_t records1_bytes(const GridDesc& g, int M);
hipError_t build_records1(const GridDesc& g, int M, double scale, void* recs, unsigned* maxpop_dev, hipStream_t stream);
template <typename T>
hipError_t launch_linear1_records(const GridDesc& g, const T* const* obs, T* out, size_t npts, hipStream_t stream);

// Bricked 2-D multilinear path (k_linear2_brick.hip): 2 x KW2 bricks, steps (1, KW2-1).
void brick2_geometry(const GridDesc& g, unsigned nb[2], size_t* bytes);
hipError_t build_bricks2(const GridDesc& g, void* bricks, hipStream_t stream);
template <typename T>
hipError_t launch_linear2_brick(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                                unsigned long long* first_bad, hipStream_t stream);

// Tiled multicubic path (k_cubic_brick.hip): dims 0,1 in 4 x 4 tiles stepped brick_step[0..1].
void cubic_tile_geometry(const GridDesc& g, int si, int sj, unsigned nb[2], size_t* bytes);
hipError_t build_cubic_tiles(const GridDesc& g, void* tiles, hipStream_t stream);
// `scatter` / `index_base`: binned evaluation (below); nullptr / 0 = points evaluated in place.
template <typename T>
hipError_t launch_cubic_brick(const GridDesc& g, const T* const* obs, T* out, size_t npts,
                              unsigned long long* first_bad, hipStream_t stream, const unsigned* scatter = nullptr,
                              size_t index_base = 0);

// Binned evaluation of the tiled multicubic kernels (k_bin_points.hip).  A 4-D cubic point reads
// 16 table lines; with unordered points and a table far beyond the 4 MiB L2 every one of them is
// an L2 miss and the kernel runs at the fabric's line rate (cfg4: 2.8 ms per 1e7 points).  The
// points of a batch are therefore first counting-sorted by the tile position of their footprint
// in dims 0,1 — a copy of the coordinates in table order plus the original index of every point —
// so that the workgroups in flight at any moment share a few hundred KiB of the table; the cubic
// kernel then reads the sorted copy and scatters its results to out[original index].  The key is
// only a locality hint (any assignment of points to bins gives the same results).
struct BinPlan {
  int nbins = 0;      // <= kMaxBins
  int nb1 = 0;        // bins along dim 1
  int mult = 1;       // bins are visited in the order key * mult mod nbins (k_bin_points.hip::bin_key)
  int inv_mult = 1;   // its inverse mod nbins
  // 1: a bin is a pair of saturation CLASSES of dims 0, 1 on a regular grid (class 0 = floc <= 0,
  // c = floc for interior cells, n - 2 = floc >= n - 2; ncell = n - 1 classes per dim, no shift):
  // what the column kernel sorts by, so that a whole bin shares its (i, j) footprint AND the form
  // of its dim-0 / dim-1 nodes.  0: footprint cells, coarsened by `shift`.
  int classes = 0;
  int ncell[2] = {0, 0};   // footprint origins per dim: n - 3
  int shift[2] = {0, 0};   // bin = cell >> shift
  // classes on a RECTILINEAR grid are exact: the sort searches dims 0, 1 like the kernels do
  // (interpn_device.h::axis_partition_point on the handle's coordinates + bucket tables)
  int rect = 0;
  const void* axis_g[2] = {nullptr, nullptr};
  const unsigned* axis_tab[2] = {nullptr, nullptr};
  int axis_n[2] = {0, 0}, axis_M[2] = {0, 0};
  double axis_g0[2] = {0, 0}, axis_scale[2] = {0, 0};
  double start[2] = {0, 0};
  double scale[2] = {0, 0};  // cell ~ floor((x - start) * scale) - 1
};
constexpr int kMaxBins = 4096;       // class-pair bins of the column evaluation (16-bit keys; 64^4 has 63^2 = 3969)
constexpr int kMaxTiledBins = 1024;  // bins of the tiled kernels' sort: one bin per thread of its scatter kernel's scan
constexpr size_t kBinSlicePoints = (size_t)1 << 25;  // points sorted and evaluated per slice (bounds the scratch)
// `classes`: one bin per pair of saturation classes of dims 0, 1 (what the column evaluation
// needs; regular grids); false when they do not fit kMaxBins.
bool make_bin_plan(const GridDesc& g, size_t table_bytes, BinPlan* plan, bool classes = false);
size_t bin_scratch_bytes(const GridDesc& g, size_t slice_points);
// What the column evaluation (cubic_column.h) gets from the sort.
struct BinExtras {
  // in: the column kernel's local sort key of a point, class of dim 2 * key_q3 + (class of dim 3 >> key_sh3)
  // (ColumnPlan::q3 / sh3, < 256), rides in the upper eight bits of its index word (slices of at most
  // 2^24 points); key_q3 = 0: plain indices
  int key_q3 = 0, key_sh3 = 0;
  const void* records = nullptr;          // the slice's points in bin order, one N-element record each
  const unsigned* bin_end = nullptr;      // end of every bin in sorted order
  const unsigned* part_prefix = nullptr;  // work list: parts in front of every bin, [nbins] = total
  unsigned* work = nullptr;               // the column kernel's part counter (zeroed by the scan)
  const unsigned* bin_flags = nullptr;    // one bit per bin: some point of it lies outside the grid along dim 0 (written only under linearised extrapolation)
};
// Sort `npts` points (one slice) into `scratch`; returns the sorted coordinate arrays and the
// original indices (within the slice).  `stage` (optional, 4 events): recorded in front of the
// histogram, behind it, behind the scan and behind the scatter.  `totals_clean`: the scratch block's
// bin counters are known to be zero (the scan of the previous complete sort left them so), no reset launch.  `extras` (4-D only): the sorted points are written as
// records instead, and the bins are cut into parts of at most `part_points` points.
hipError_t bin_points(const GridDesc& g, const BinPlan& plan, const void* const* obs, size_t npts, void* scratch,
                      const void** binned_obs, const unsigned** index, hipStream_t stream, BinExtras* extras = nullptr,
                      unsigned part_points = 0, hipEvent_t* stage = nullptr, bool totals_clean = false);

// Column evaluation of sorted 4-D multicubic points on a regular grid (cubic_column.h): does it
// apply to this grid (LDS capacity, classes fit the bins), and the launch.
bool cubic_column_applies(const GridDesc& g);
constexpr unsigned kColumnMaxPart = 12288;  // points per workgroup at most (16-bit local order: 24 KiB of LDS beside the column)
// The K-range phases of the column evaluation for this grid and these options (k_cubic_column.hip).
struct ColumnPlan {
  int threads = 768;         // workgroup size
  int groups = 2;            // wave groups of a workgroup, each working on a part of its own
  unsigned part_points = 0;  // points of a part at most (32 per thread of a group)
  int cpp = 0;               // classes of dim 2 per phase
  int nphase = 0;
  int q3 = 0, sh3 = 0;       // local sort key = class2 * q3 + (class3 >> sh3)
  unsigned pitch = 0;        // LDS bytes from tile to tile: the tile + 16 bytes of padding, or the bare tile where that saves phases
  unsigned perm_pad = 0;     // entries of a part's local order that live in the tiles' padding
  unsigned sub_bytes = 0;    // LDS bytes of a phase's sub-column
  unsigned group_bytes = 0;  // dynamic LDS of a group: sub-column + local order
  size_t axes_bytes = 0;     // rectilinear: the axis image behind the groups' regions
  size_t lds_bytes = 0;      // dynamic LDS of a workgroup
};
bool cubic_column_plan(const GridDesc& g, ColumnPlan* plan);
// the sort writes the local-sort keys into the index words (regular grids; option column_keys)
inline bool column_keys_in_index(const GridDesc& g) { return g.cfg.column_keys != 0 && g.kind == kRegular; }
constexpr size_t kColumnKeySlicePoints = (size_t)1 << 24;  // then an index has 24 bits
template <typename T>
hipError_t launch_cubic_column(const GridDesc& g, const BinPlan& plan, const BinExtras& extras, const unsigned* index,
                               T* out, size_t npts, size_t max_parts, unsigned long long* first_bad, size_t index_base,
                               hipStream_t stream);
// The same for sorted 3-D points (cubic3_column.h): the cell's column of n2 tiles in LDS, 256-thread workgroups.
bool cubic3_column_applies(const GridDesc& g);
unsigned cubic3_column_part_points();
template <typename T>
hipError_t launch_cubic3_column(const GridDesc& g, const BinPlan& plan, const BinExtras& extras, const unsigned* index,
                                T* out, size_t npts, size_t max_parts, unsigned long long* first_bad, size_t index_base,
                                hipStream_t stream);

// Bucket table of one axis (device): tab[0..M] from the coordinates g[0..n).
template <typename T>
hipError_t build_buckets(const T* g, int n, int M, T g0, T scale, unsigned* tab, hipStream_t stream);
template <typename T>
hipError_t build_lane_table(const T* g, int n, T g0, T scale, unsigned* words65, hipStream_t stream);

// check_bounds: OR into flag[0] whether any of x[0..n) violates [lo, hi] by atol or more
// (src/multilinear/regular.rs:168-171).
template <typename T>
hipError_t launch_check_bounds(const T* x, size_t n, T lo, T hi, T atol, unsigned* flag, hipStream_t stream);

// True when the templated (flattened-arm) kernels apply: N within the flattened range of the
// reference's dispatch and the grid indexable with 32 bits.
inline bool fast_path(const GridDesc& g) {
  if (g.method == kNearest) return true;
  const int maxn = g.method == kLinear ? 6 : 4;
  return g.ndims >= 1 && g.ndims <= maxn && g.nvals < 0xFFFFFFFFull;
}

}  // namespace interpn
