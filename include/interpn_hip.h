/*
 * interpn_hip.h — C ABI of libinterpn_hip.so: MI355X (gfx950) implementation of the batched
 * per-observation-point hot path of jlogan03/interpn v0.8.2.
 *
 * What it replaces (paths relative to the reference repository):
 *   multilinear::regular::interpn       src/multilinear/regular.rs:51-117
 *   multilinear::rectilinear::interpn   src/multilinear/rectilinear.rs:49-83
 *   multicubic::regular::interpn        src/multicubic/regular.rs:52-136
 *   multicubic::rectilinear::interpn    src/multicubic/rectilinear.rs:54-104
 * and who would call it: the bodies of the PyO3 functions interpn_{linear,cubic}_
 * {regular,rectilinear}_{f64,f32} (src/python.rs:55-85, 119-147, 228-292) through a Rust
 * `hip_sys` extern block (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - Every Rust slice `&[T]` crosses the boundary as (pointer, length); a slice of slices
 *     `&[&[T]]` as (array of pointers, array of lengths, count).  All buffers are borrowed for
 *     the duration of the call only; nothing is retained after a function returns, except the
 *     device copies a handle owns (and a device `vals` buffer lent with INTERPN_HIP_MEM_DEVICE,
 *     which must outlive the handle).
 *   - `vals` is C-ordered; `obs` is struct-of-arrays (one array per dimension); `out` has one
 *     element per observation point and is written only in [0, nout).
 *   - Return value: 0 on success, otherwise an interpn_hip_status.  interpn_hip_strerror()
 *     returns, for the statuses that mirror a reference error, the reference's exact
 *     `&'static str` (what src/python.rs turns into AssertionError(msg)).
 *   - INTERPN_HIP_ERR_UNREPRESENTABLE ("Unrepresentable coordinate value"): the reference
 *     aborts the batch at the first failing point i with out[0..i) written and out[i..]
 *     untouched.  The host-pointer entry points reproduce exactly that.  The device-pointer
 *     entry point reports i through interpn_hip_finish(); device `out[i..]` is unspecified.
 *   - Conditions on which the reference *panics* (mismatched slice lengths in
 *     multicubic::*::interpn, usize overflow) return INTERPN_HIP_ERR_REFERENCE_PANIC instead
 *     of aborting the process.
 *   - Numerics follow the reference's `fma` cargo feature ON (what every published wheel is
 *     built with, pyproject.toml:72); interpn_hip_set_fma(0) selects the non-fused flavour
 *     (plain `cargo test`).  Results are bit-identical to the Rust code of the same flavour.
 *   - Memory: besides the C-ordered `vals`, a handle may keep a second, re-laid copy of the grid
 *     (cache-line bricks for multilinear N = 2..6, 4 x 4 tiles for multicubic N = 2..4; up to 16x
 *     the grid, bounded by a quarter of the free device memory).  Tuning / testing knobs.  The
 *     environment is read ONCE PER HANDLE, when it is created (never on the launch path); the
 *     per-handle options can be changed afterwards with interpn_hip_set_option(h, "<name>", v),
 *     <name> = the variable's suffix in lower case:
 *       INTERPN_HIP_BRICKS=off|11|12|22|c4|j4 (linear) |44|24|22|14|11 (cubic)   force / disable a layout (creation only;
 *                                       c4 = 4-D cell bricks, j4 = the f32-only 2 x 4 x 4 bricks)
 *       INTERPN_HIP_BLOCKS_PER_CU=n     workgroups per CU a persistent launch grid is sized for (default 8)
 *       INTERPN_HIP_ITERS_PER_BLOCK=n   256-lane rows per workgroup of the one-pass brick kernels (0 = default)
 *       INTERPN_HIP_PPL=1               one point per lane in the multilinear brick kernels (0 = auto)
 *       INTERPN_HIP_FORCE_GENERIC=1     route every evaluation through the runtime-N kernel
 *       INTERPN_HIP_GENERIC_RUNTIME=1   recursive arms: runtime-N form
 *       INTERPN_HIP_GENERIC_VEC=0|1     recursive arms: one-tree / row-vector form (-1 = auto; the
 *                                       row-vector form exists only where it compiles without
 *                                       AGPR or scratch spills, see k_generic.hip)
 *       INTERPN_HIP_AXIS_REGS=0|1|2     rectilinear axes <= 64 coordinates: 0 = search in LDS,
 *                                       1 = across lanes without the lane table, 2 = with it (-1 = auto)
 *       INTERPN_HIP_AXIS_LDS_KB=n       LDS budget of the rectilinear axis image (-1 = default)
 *       INTERPN_HIP_PERSISTENT=1        C-order regular / nearest kernels: persistent grid
 *       INTERPN_HIP_HOST_CHUNK=n        points per chunk of the host-pointer pipeline (0 = default 2 Mi)
 *       INTERPN_HIP_BINNED=-1|0|1       tiled multicubic, device-pointer evaluation: sort the points by table position
 *                                       first (auto: large 4-D batches; 1: always, N = 2..4; 0: never)
 *       INTERPN_HIP_COLUMN=-1|0|1       sorted 4-D multicubic on a regular grid: evaluate out of an LDS-resident table
 *                                       column (auto: from ~3000 points per bin; 1: wherever it applies; 0: never);
 *                                       INTERPN_HIP_COLUMN_THREADS=256|384|768, INTERPN_HIP_COLUMN_PART=n (points per part at most),
 *                                       INTERPN_HIP_COLUMN_GROUPS=1|2, INTERPN_HIP_COLUMN_CPP=n (dim-2 classes per K-range phase),
 *                                       INTERPN_HIP_COLUMN_COEF=0 (every node from the table values; default 1: dim 0 from
 *                                       per-part Hermite coefficients, cubic_column.h), INTERPN_HIP_COLUMN_PAD=-1|0|1 (LDS tiles
 *                                       bare / 16 bytes apart), INTERPN_HIP_COLUMN_TAIL=0xDV (the last 1/D of the bins cut V
 *                                       times finer; default 0x84), INTERPN_HIP_COLUMN_KEYS=0 (regular grids: the local sort's keys
 *                                       from the records instead of from the upper eight bits of the index words, where the sort
 *                                       leaves them by default — slices are then 2^24 points at most),
 *                                       INTERPN_HIP_SCATTER_STAGED=0 (the sort stores records directly)
 *       INTERPN_HIP_SWEEP=-1|0|1        2-D / 3-D multilinear, nearest-neighbour and multicubic (f64, f32), device-pointer evaluation: the sweep kernel (every wave orders 1024 / 1536
 *                                       points by leading cell index on chip, all waves walk the one-line brick table in step
 *                                       with a clock; linear_sweep.h; multicubic: 512 / 1280 points by their cell along dim 2,
 *                                       rows on the fully overlapped tile table, cubic_sweep.h): auto (multilinear: batches of >= 4
 *                                       rounds per wave ~ 1.26e7 points in f64, 8 where the L2 holds the table, f32 3 / 6; multicubic:
 *                                       regular grids with that table beyond the L2, from 2 rounds per wave ~ 4e6 points; 2-D
 *                                       multicubic: f64 regular grids from 8 rounds per wave ~ 2.2e7 points), never,
 *                                       or whenever the handle has the table (creation: 0 also skips building it);
 *                                       INTERPN_HIP_SWEEP_PERIOD=n ticks of 10 ns per sweep (0 = what the previous launch
 *                                       measured, 1 = no clock); INTERPN_HIP_SWEEP_LAYOUT=11|12 (creation only, 3-D f64
 *                                       multilinear: the sweep kernel's table with one line per cell / 1.5 lines per cell at half
 *                                       the size; default: by the table's size against the points an XCD holds per sweep)
 *       INTERPN_HIP_BIN_SLICE_LOG2=n    log2 of the points sorted per slice (16..27, default 25): bounds a scratch block
 *       INTERPN_HIP_AXIS_RECORDS=0      rectilinear multilinear: search with coordinates + tables, not per-bucket records
 *       INTERPN_HIP_CUBIC_RECORDS=n     rectilinear multicubic (creation only): per-cell records of the axes (cubic_cell_record.h: a
 *                                       dimension's setup without divisions) while they take at most n KiB (default 40; 0: never —
 *                                       the kernels then do the setup's divisions per point)
 *       INTERPN_HIP_BIN_SCRAMBLE=1      testing: the sort misplaces every 5th point by one bin (results must not change)
 *       options without an environment variable: "fma" (the handle's flavour), "stage_timing" (interpn_hip_stage_ms),
 *                                       "debug_stamps" + "debug_stamps_bytes" (address and size of a device buffer for the column
 *                                       kernel's time stamps, 64 bytes per part; a launch that needs more writes none)
 *       INTERPN_HIP_POOL_MB=n           device bytes of destroyed handles kept for reuse, per device
 *                                       (process-wide, read once; default 1024; 0 = release
 *                                       everything at destroy)
 *   - Thread safety: all functions are re-entrant.  Device-pointer evaluations on one handle may
 *     run concurrently (the grid is read-only; the sticky first-bad-index word of the handle is
 *     shared by them); host-pointer evaluations on one handle share its staging buffers and are
 *     serialised internally — use one handle per thread to overlap them.
 */
#ifndef INTERPN_HIP_H
#define INTERPN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum interpn_hip_status {
  INTERPN_HIP_OK = 0,
  /* mirrors of the reference's error strings */
  INTERPN_HIP_ERR_DIM_MISMATCH = 1,      /* "Dimension mismatch" */
  INTERPN_HIP_ERR_MIN_TWO_ENTRIES = 2,   /* "All grids must have at least two entries"  multilinear/regular.rs:245 */
  INTERPN_HIP_ERR_MIN_2_ENTRIES = 3,     /* "All grids must have at least 2 entries"    multilinear/rectilinear.rs:192 */
  INTERPN_HIP_ERR_MIN_FOUR_ENTRIES = 4,  /* "All grids must have at least four entries" multicubic/regular.rs:261 */
  INTERPN_HIP_ERR_MIN_4_ENTRIES = 5,     /* "All grids must have at least 4 entries"    multicubic/rectilinear.rs:214 */
  INTERPN_HIP_ERR_NOT_MONOTONIC = 6,     /* "All grids must be monotonically increasing" */
  INTERPN_HIP_ERR_UNREPRESENTABLE = 7,   /* "Unrepresentable coordinate value"          multilinear/regular.rs:418 */
  INTERPN_HIP_ERR_TOO_MANY_DIMS = 8,     /* "Dimension exceeds maximum (8). Use interpolator struct directly for higher dimensions." */
  INTERPN_HIP_ERR_REFERENCE_PANIC = 9,   /* the reference would panic here */
  INTERPN_HIP_ERR_TOO_MANY_DIMS_6 = 10,  /* "Dimension exceeds maximum (6)."           nearest/regular.rs:97 */
  /* statuses of this implementation */
  INTERPN_HIP_ERR_INVALID_ARGUMENT = 32, /* null pointer, unknown enum value, dtype mismatch */
  INTERPN_HIP_ERR_UNSUPPORTED = 33,      /* axis longer than 2^31-257 points (f32: 2^24) */
  INTERPN_HIP_ERR_NO_DEVICE = 34,        /* no usable HIP device */
  INTERPN_HIP_ERR_OUT_OF_MEMORY = 35,    /* device or pinned-host allocation failed */
  INTERPN_HIP_ERR_HIP = 36               /* any other HIP runtime failure (see interpn_hip_last_hip_error) */
} interpn_hip_status;

enum { INTERPN_HIP_LINEAR = 0, INTERPN_HIP_CUBIC = 1, INTERPN_HIP_NEAREST = 2 }; /* method */
/* The reference's `fma` cargo feature (Cargo.toml:34-38) is a PER-INTERPOLATOR property here: OR one
 * of these into the `method` argument of interpn_hip_create_* (neither = the process default, see
 * interpn_hip_set_fma).  Handles of both flavours can be evaluated concurrently from different
 * threads.  Also readable / writable as the per-handle option "fma" (not while evaluations of the
 * same handle are being enqueued by another thread). */
enum { INTERPN_HIP_FLAVOUR_FMA = 0x100, INTERPN_HIP_FLAVOUR_NO_FMA = 0x200 };
enum { INTERPN_HIP_MEM_HOST = 0, INTERPN_HIP_MEM_DEVICE = 1 }; /* where a buffer lives */

const char* interpn_hip_strerror(int status);
/* Text of the last HIP runtime error seen by the calling thread ("" if none). */
const char* interpn_hip_last_hip_error(void);
/* Library version "major.minor.patch". */
const char* interpn_hip_version(void);
/* DEPRECATED default-setter, kept for the one-shot entry points (whose signatures are the Rust
 * functions' and have no place for a flavour — there it plays the role of the compile-time cargo
 * feature): flavour of interpolators created afterwards WITHOUT an INTERPN_HIP_FLAVOUR_* flag
 * (process-wide; default 1).  Returns the previous value.  Never changes an existing handle. */
int interpn_hip_set_fma(int enabled);
/* Number of visible HIP devices (0 when none; never fails). */
int interpn_hip_device_count(void);
/* Give back to the driver the device memory of destroyed handles that the library keeps for reuse
 * on `device` (-1: the current device): cached and parked blocks alike (INTERPN_HIP_POOL_MB).
 * Waits for the device like hipFree.  *freed_bytes (may be NULL) receives what was released. */
int interpn_hip_trim(int device, size_t* freed_bytes);

/* ------------------------------------------------------------------------------------------
 * One-shot entry points, host pointers — drop-in for the calls made inside the PyO3 bodies.
 * They build the interpolator, move the batch through the current HIP device in chunks and
 * return when `out` is complete.
 *
 * interpn_hip_linear_regular_*      <- multilinear::regular::interpn(dims, starts, steps, vals, obs, out)
 *                                      src/python.rs:69-76
 * interpn_hip_linear_rectilinear_*  <- multilinear::rectilinear::interpn(grids, vals, obs, out)
 *                                      src/python.rs:133-138
 * interpn_hip_cubic_regular_*       <- multicubic::regular::interpn(dims, starts, steps, vals,
 *                                      linearize_extrapolation, obs, out)  src/python.rs:243-251
 * interpn_hip_cubic_rectilinear_*   <- multicubic::rectilinear::interpn(grids, vals,
 *                                      linearize_extrapolation, obs, out)  src/python.rs:277-283
 * interpn_hip_nearest_regular_*     <- nearest::regular::interpn(dims, starts, steps, vals, obs, out)
 *                                      src/python.rs:162-169   (N <= 6)
 * interpn_hip_nearest_rectilinear_* <- nearest::rectilinear::interpn(grids, vals, obs, out)
 *                                      src/python.rs:192
 * ---------------------------------------------------------------------------------------- */
#define INTERPN_HIP_DECLARE_ONESHOT(T, SUFFIX)                                                            \
  int interpn_hip_linear_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts,             \
                                          size_t nstarts, const T* steps, size_t nsteps, const T* vals,  \
                                          size_t nvals, const T* const* obs, const size_t* obs_lens,     \
                                          size_t nobs, T* out, size_t nout);                             \
  int interpn_hip_linear_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens,            \
                                              size_t ngrids, const T* vals, size_t nvals,                \
                                              const T* const* obs, const size_t* obs_lens, size_t nobs,  \
                                              T* out, size_t nout);                                      \
  int interpn_hip_cubic_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts,              \
                                         size_t nstarts, const T* steps, size_t nsteps, const T* vals,   \
                                         size_t nvals, int linearize_extrapolation, const T* const* obs, \
                                         const size_t* obs_lens, size_t nobs, T* out, size_t nout);      \
  int interpn_hip_cubic_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens,             \
                                             size_t ngrids, const T* vals, size_t nvals,                 \
                                             int linearize_extrapolation, const T* const* obs,           \
                                             const size_t* obs_lens, size_t nobs, T* out, size_t nout);      \
  int interpn_hip_nearest_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts,            \
                                           size_t nstarts, const T* steps, size_t nsteps, const T* vals, \
                                           size_t nvals, const T* const* obs, const size_t* obs_lens,    \
                                           size_t nobs, T* out, size_t nout);                            \
  int interpn_hip_nearest_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens,           \
                                               size_t ngrids, const T* vals, size_t nvals,               \
                                               const T* const* obs, const size_t* obs_lens, size_t nobs, \
                                               T* out, size_t nout);

INTERPN_HIP_DECLARE_ONESHOT(double, f64)
INTERPN_HIP_DECLARE_ONESHOT(float, f32)

/* ------------------------------------------------------------------------------------------
 * Persistent interpolators — the counterpart of MultilinearRegular::new / MulticubicRegular::new
 * ... (src/multilinear/regular.rs:225-259, rectilinear.rs:175-201, multicubic/regular.rs:239-288,
 * rectilinear.rs:193-228) whose grid stays resident in HBM between evaluations, and of
 * `.interp(obs, out)` (regular.rs:268-283 etc.).  This is the surface the Python classes'
 * `.eval()` sits on (src/interpn/multilinear_regular.py:101-168).
 *
 * `method`  INTERPN_HIP_LINEAR | INTERPN_HIP_CUBIC | INTERPN_HIP_NEAREST
 * `vals_mem` INTERPN_HIP_MEM_HOST: `vals` is copied to the device;
 *            INTERPN_HIP_MEM_DEVICE: `vals` is a device pointer on `device`, borrowed (e.g. the
 *            buffer an RCCL broadcast just filled on this rank).
 * `device`  HIP device ordinal, or -1 for the calling thread's current device.
 * Validation (order and messages) is the reference's `new`.  Axis coordinates, starts and steps
 * are always host pointers (they are a few KiB).
 * ---------------------------------------------------------------------------------------- */
typedef struct interpn_hip_interp interpn_hip_interp;

#define INTERPN_HIP_DECLARE_CREATE(T, SUFFIX)                                                             \
  int interpn_hip_create_regular_##SUFFIX(int method, const size_t* dims, size_t ndims, const T* starts, \
                                          size_t nstarts, const T* steps, size_t nsteps, const T* vals,  \
                                          size_t nvals, int vals_mem, int linearize_extrapolation,       \
                                          int device, interpn_hip_interp** handle);                      \
  int interpn_hip_create_rectilinear_##SUFFIX(int method, const T* const* grids,                         \
                                              const size_t* grid_lens, size_t ngrids, const T* vals,     \
                                              size_t nvals, int vals_mem, int linearize_extrapolation,   \
                                              int device, interpn_hip_interp** handle);

INTERPN_HIP_DECLARE_CREATE(double, f64)
INTERPN_HIP_DECLARE_CREATE(float, f32)

/* Clone `src` onto `device` (-1 = current) of the same process: the grid (and the axes of a
 * rectilinear grid) travels device to device (hipMemcpyPeer, i.e. xGMI between the GPUs of one
 * node; no host staging, no second upload), the re-laid table is rebuilt on the target.  The
 * single-process way to "replicate the read-only grid on every GPU"  in
 * front of interpn_hip_eval_host_sharded; multi-process callers broadcast `vals` with RCCL and
 * create from the device pointer (INTERPN_HIP_MEM_DEVICE).  The clone owns its copy. */
int interpn_hip_replicate(const interpn_hip_interp* src, int device, interpn_hip_interp** out);

/* sizeof of the element type of the interpolator (8 or 4), number of dimensions, device. */
int interpn_hip_elem_size(const interpn_hip_interp* h);
int interpn_hip_ndims(const interpn_hip_interp* h);
int interpn_hip_device(const interpn_hip_interp* h);

/* Evaluate on host arrays (synchronous).  `obs`/`out` element type must be the handle's.
 * Mirrors `.interp(obs, out)`: "Dimension mismatch" unless nobs == ndims and every
 * obs_lens[d] == nout; abort-at-first-bad-point semantics as described above. */
int interpn_hip_eval_host(interpn_hip_interp* h, const void* const* obs, const size_t* obs_lens,
                          size_t nobs, void* out, size_t nout);

/* Single-process multi-GPU form of `interpn_hip_eval_host`: cuts the observation index into
 * `nhandles` contiguous ranges (the first nout % nhandles ranges one point longer; SURVEY.md
 * section 8(e)) and evaluates range r with handles[r] on that handle's device, one host thread per
 * handle, no device-to-device traffic.  The handles must be distinct and describe the same
 * interpolator (create one per device from the same grid).  Status and checks as
 * `interpn_hip_eval_host`.  On INTERPN_HIP_ERR_UNREPRESENTABLE `*first_bad_index` (optional)
 * receives the global index i of the first failing point; out[0..i) holds the reference's
 * results, out[i..] is unspecified here (ranges behind the failing one ran concurrently),
 * where the one-handle form leaves it untouched. */
int interpn_hip_eval_host_sharded(interpn_hip_interp* const* handles, size_t nhandles,
                                  const void* const* obs, const size_t* obs_lens, size_t nobs,
                                  void* out, size_t nout, uint64_t* first_bad_index);

/* Device-resident single-process multi-GPU form (round 4): a caller that is ONE process — the
 * reference's PyO3 module is (src/python.rs:58-80) — with the observation points already sharded
 * over the devices.  Shard r is obs[r][0..nobs) (a HOST array of `nobs` DEVICE pointers on the
 * device of handles[r], each to npoints[r] elements) -> out[r] (a device pointer there), enqueued
 * on streams[r] (hipStream_t; `streams` or an entry may be NULL = default stream).  All shards
 * are enqueued before the first one is waited for; the call returns when every shard has finished.
 * No PCIe traffic besides the 8-byte status words, no collective.  The handles must be distinct
 * and describe the same interpolator (interpn_hip_replicate).  On
 * INTERPN_HIP_ERR_UNREPRESENTABLE `*first_bad_index` (optional) is the global index of the
 * first failing point, counting the shards' points in shard order; results in front of it are
 * the reference's, results behind it unspecified. */
int interpn_hip_eval_device_sharded(interpn_hip_interp* const* handles, size_t nhandles,
                                    const void* const* const* obs, size_t nobs, void* const* out,
                                    const size_t* npoints, void* const* streams,
                                    uint64_t* first_bad_index);

/* Evaluate on device arrays (asynchronous on `stream`, a hipStream_t; NULL = default stream).
 * `obs` is a HOST array of `nobs` DEVICE pointers, each to `npoints` elements; `out` is a device
 * pointer to `npoints` elements.  Returns as soon as the work is enqueued: one kernel, no copy and
 * no synchronisation, so the call can be captured into a hipGraph.  One exception outside capture:
 * large batches on 4-D multicubic grids whose re-laid table is far beyond the L2 are first
 * counting-sorted by the table position of their footprint (three more launches on `stream`
 * into a scratch block the handle keeps; allocated on the first such call), which makes the
 * 16 table lines a point reads L2 hits instead of misses (option "binned": -1 auto, 0 never,
 * 1 always for N = 2..4).  Results and the first-failing-index contract are unchanged. */
int interpn_hip_eval_device(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out,
                            size_t npoints, void* stream);

/* The same evaluation, telling the caller which path it took.
 *   flags        INTERPN_HIP_EVAL_NO_ALLOC: never allocate (scratch that interpn_hip_reserve has
 *                not provided is then a reason to evaluate in place, reported below).
 *   *path_taken  INTERPN_HIP_PATH_IN_PLACE (one kernel on the points as given),
 *                INTERPN_HIP_PATH_BINNED (points counting-sorted first: 4 launches per slice) or
 *                INTERPN_HIP_PATH_SWEEP (3-D multilinear, large batches: one persistent kernel
 *                whose waves order their points on chip and walk the table in step; 1.25 KiB of
 *                scratch per stream).
 *   *why         for IN_PLACE on a handle that could bin: the reason (INTERPN_HIP_WHY_*); else 0.
 * Either pointer may be NULL.  The sorted path needs scratch (36 B per point of a slice for 4-D
 * f64, slices of at most 2^25 points): one block per stream that uses the handle concurrently, at
 * most 4; streams beyond that take turns through events.  After interpn_hip_reserve(h, n, k) no
 * evaluation of at most n points on at most k streams allocates anything. */
enum { INTERPN_HIP_PATH_IN_PLACE = 0, INTERPN_HIP_PATH_BINNED = 1, INTERPN_HIP_PATH_SWEEP = 2 };
enum { INTERPN_HIP_EVAL_NO_ALLOC = 1 };
enum {
  INTERPN_HIP_WHY_NONE = 0,          /* binned, or binning never applies to this handle */
  INTERPN_HIP_WHY_SMALL_OR_OFF = 1,  /* batch below the break-even size, or option "binned" = 0 */
  INTERPN_HIP_WHY_CAPTURE = 2,       /* `stream` is being captured into a graph */
  INTERPN_HIP_WHY_NO_SCRATCH = 3,    /* no reserved scratch block is free / large enough and allocation was not allowed */
  INTERPN_HIP_WHY_ALLOC_FAILED = 4,  /* the scratch allocation failed */
  INTERPN_HIP_WHY_MISALIGNED = 5     /* sweep evaluation only: `out` or a coordinate array is not 16-byte aligned (its streams are 16-byte accesses) */
};
int interpn_hip_eval_device_ex(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out,
                               size_t npoints, void* stream, unsigned flags, int* path_taken, int* why);

/* Pre-allocate what device-pointer evaluations of up to `npoints` points on up to `nstreams`
 * concurrent streams need (the sorted path's scratch blocks; nothing for handles that never
 * sort).  Synchronous; may be called again with larger numbers.  Per-handle counters readable
 * with interpn_hip_get_option: "evals_binned", "evals_in_place", "scratch_allocs",
 * "scratch_bytes". */
int interpn_hip_reserve(interpn_hip_interp* h, size_t npoints, int nstreams);

/* Measurement aid (bench.py): with option "stage_timing" = 1 a sorted (binned) single-slice
 * evaluation records HIP events between its launches; this waits for the most recent one and
 * returns its stage durations in milliseconds: ms[0] histogram (+ counter reset), ms[1] scan,
 * ms[2] scatter, ms[3] evaluation kernel.  n >= 4.  INVALID_ARGUMENT when no such evaluation ran. */
int interpn_hip_stage_ms(interpn_hip_interp* h, double* ms, size_t n);

/* Wait for `stream` and report the sticky status of the device evaluations enqueued since the
 * last finish: 0, or INTERPN_HIP_ERR_UNREPRESENTABLE with the smallest failing point index
 * (relative to the evaluation it occurred in) in *first_bad_index.  Clears the sticky word. */
int interpn_hip_finish(interpn_hip_interp* h, void* stream, uint64_t* first_bad_index);

/* Tuning knob: workgroups per CU the launch grid is sized for (default 8). */
int interpn_hip_set_blocks_per_cu(interpn_hip_interp* h, int blocks_per_cu);

/* Per-handle tuning / testing options by name (the list is in the header comment above:
 * "blocks_per_cu", "iters_per_block", "ppl", "axis_regs", "force_generic", "generic_runtime",
 * "generic_vec", "persistent", "axis_lds_kb", "host_chunk", "binned", "deal", "bin_slice_log2" (points
 * per sorted slice, 16..27, default 25: bounds a scratch block), "fma"; read-only: "last_binned" = 1
 * if the most recent device-pointer evaluation sorted its points first, and the counters listed
 * at interpn_hip_reserve).  Their defaults are latched from the
 * INTERPN_HIP_* environment variables when the handle is created.  INTERPN_HIP_ERR_INVALID_ARGUMENT
 * for an unknown name or a value out of range.  Not synchronised against evaluations running
 * concurrently on the same handle. */
int interpn_hip_set_option(interpn_hip_interp* h, const char* name, long long value);
int interpn_hip_get_option(const interpn_hip_interp* h, const char* name, long long* value);

/* Name of the kernel instantiation the most recent evaluation through this handle launched, in
 * rocprofv3's spelling without return type and argument list, e.g.
 * "interpn::k_linear_brick<double, 3, false, true, 1, 2, 2, 0>"; "" before the first launch.
 * (bench.py reports this instead of a hard-coded string.) */
int interpn_hip_kernel_name(const interpn_hip_interp* h, char* buf, size_t buflen);

/* Bytes of the re-laid grid copy the handle keeps (0 = the kernels read the C-ordered `vals`);
 * *step_i / *step_j (optional) receive the layout's brick / tile steps. */
size_t interpn_hip_table_bytes(const interpn_hip_interp* h, int* step_i, int* step_j);

/* Waits only for work enqueued through THIS handle (an event behind its last launch on every
 * caller stream, plus its own staging streams) before its device memory is recycled; other
 * streams of the device keep running.  Launches captured into a graph cannot be tracked: the
 * caller must not replay such a graph after destroying the handle. */
void interpn_hip_destroy(interpn_hip_interp* h);

/* ------------------------------------------------------------------------------------------
 * check_bounds (src/multilinear/regular.rs:145-182, rectilinear.rs:109-134; exported to Python
 * as check_bounds_{regular,rectilinear}_{f64,f32}, src/python.rs:87-117, 203-226).
 * out[d] = 1 if any observation violates the bounds of dimension d by atol or more.
 * Host pointers; streamed through the current device.
 * ---------------------------------------------------------------------------------------- */
/* The same check on DEVICE-resident coordinates, limits taken from the handle's grid (the
 * pre-pass of the reference's Python `interpn(..., check_bounds=True)`, src/interpn/__init__.py:
 * 115-132, without moving the points over PCIe).  `obs`: host array of `nobs` device pointers
 * to `npoints` elements of the handle's type; `out`: host, `nout` = ndims bytes.  Synchronous
 * on `stream`. */
int interpn_hip_check_bounds_device(interpn_hip_interp* h, const void* const* obs, size_t nobs,
                                    size_t npoints, double atol, uint8_t* out, size_t nout,
                                    void* stream);

#define INTERPN_HIP_DECLARE_BOUNDS(T, SUFFIX)                                                            \
  int interpn_hip_check_bounds_regular_##SUFFIX(const size_t* dims, size_t ndims, const T* starts,      \
                                                size_t nstarts, const T* steps, size_t nsteps,          \
                                                const T* const* obs, const size_t* obs_lens,            \
                                                size_t nobs, T atol, uint8_t* out, size_t nout);        \
  int interpn_hip_check_bounds_rectilinear_##SUFFIX(const T* const* grids, const size_t* grid_lens,     \
                                                    size_t ngrids, const T* const* obs,                 \
                                                    const size_t* obs_lens, size_t nobs, T atol,        \
                                                    uint8_t* out, size_t nout);

INTERPN_HIP_DECLARE_BOUNDS(double, f64)
INTERPN_HIP_DECLARE_BOUNDS(float, f32)

#ifdef __cplusplus
}
#endif

#endif /* INTERPN_HIP_H */
