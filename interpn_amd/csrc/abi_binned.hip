// Sorted evaluation of device-resident batches: the counting sort by tile position / saturation
// class in front of the tiled or the LDS-column multicubic kernel, its scratch blocks per stream,
// and the device-pointer entry points.  (C ABI internals, see abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

namespace interpn_abi {


// Binned evaluation of the tiled multicubic kernels on device-resident points (interpn_host.h).
// Returns -1 when the path does not apply or cannot be taken right now (the caller then launches
// the kernel on the points as they are), otherwise a status.  Chosen automatically for 4-D grids
// whose tile table is far beyond the L2 and batches large enough to pay for the three sorting
// launches (cfg4: 2.8 -> 1.85 ms per 1e7 points; from about 5e5 points on; 3-D grids lose: 4 lines
// per point are cheaper than sorting them); `binned` = 1 forces it for N = 2..4 (tests),
// 0 turns it off.  Not taken while the stream is being captured into a graph (it may have to
// allocate) or while another thread is inside it with the same handle.
// Slice length of the sorted path (bounds one scratch block): option "bin_slice_log2".
size_t bin_slice_points(const GridDesc& g) {
  const int lg = g.cfg.bin_slice_log2 >= 16 && g.cfg.bin_slice_log2 <= 27 ? g.cfg.bin_slice_log2 : 25;
  const size_t s = (size_t)1 << lg;
  return s < kBinSlicePoints ? s : kBinSlicePoints;
}

// Does the sorted path apply to this handle at all / to a batch of `npoints`?  (No HIP calls.)
// Returns 0 = never for this handle, 1 = not for this batch (too small / switched off), 2 = yes.
int binned_applies(const GridDesc& g, size_t npoints) {
  if (g.method != kCubic || !g.bricks || g.ndims < 2 || g.ndims > 4) return 0;
  if (g.cfg.binned == 0 || g.cfg.force_generic) return 1;
  const bool main11 = g.brick_step[0] == 1 && g.brick_step[1] == 1;
  const bool second = !main11 && g.bricks11 != nullptr;
  if (g.cfg.binned < 0) {
    if (g.ndims != 4 && g.ndims != 3) return 0;
    if (!main11 && !second) return 0;
    if (g.ndims == 3) return 0;  // 3-D: never by itself — measured (round 5, 64^3 f64, 1e7 points): sort 0.35 + column kernel 0.55 ms
                                 // against 0.65 in place (four miss lines per point); option "binned" = 1 takes it (cubic3_column.h)
    if (main11) {
      unsigned nb[2];
      size_t table = 0;
      cubic_tile_geometry(g, 1, 1, nb, &table);
      if (table <= thresholds(g.cfg).binned_table_min) return 0;  // an L2-sized table (<= 2 x L2) is gathered at the hit rate anyway
    }
    if (npoints < thresholds(g.cfg).binned_points_min) return 1;  // 2048 points per CU: 2^19 on MI355X
  }
  return 2;
}

// Take a scratch block of at least `need` bytes for an evaluation on `stream` (see BinSlot).
// On success the block is marked busy and, where another stream used it last, `stream` has been
// made to wait for that use.  `*why` says why not otherwise.
interpn_hip_interp::BinSlot* take_bin_slot(interpn_hip_interp* h, size_t need, hipStream_t stream, bool may_alloc, int* why) {
  using Slot = interpn_hip_interp::BinSlot;
  std::unique_lock<std::mutex> lk(h->bin_mu);  // (step 5 drops it around a wait)
  Slot* pick = nullptr;
  bool wait = false;
  for (auto& sl : h->bin_slots)  // 1. the block this stream used last: stream order is enough
    if (!sl.busy && sl.bytes >= need && sl.recorded && sl.last_stream == stream) { pick = &sl; break; }
  if (!pick)
    for (auto& sl : h->bin_slots) {  // 2. an idle block
      if (sl.busy || sl.bytes < need) continue;
      if (!sl.recorded) { pick = &sl; break; }
      const hipError_t q = hipEventQuery(sl.event);
      if (q == hipSuccess) { pick = &sl; break; }
      if (q != hipErrorNotReady) (void)hipGetLastError();
    }
  if (!pick && may_alloc && h->bin_slots.size() < interpn_hip_interp::kMaxBinSlots) {  // 3. a new block
    Slot sl;
    if (hipEventCreateWithFlags(&sl.event, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); *why = INTERPN_HIP_WHY_ALLOC_FAILED; return nullptr; }
    if (pool_alloc(h->device, &sl.scratch, need) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipEventDestroy(sl.event);
      *why = INTERPN_HIP_WHY_ALLOC_FAILED;
      return nullptr;
    }
    sl.bytes = need;
    h->scratch_allocs.fetch_add(1);
    h->bin_slots.push_back(sl);
    pick = &h->bin_slots.back();
  }
  if (!pick) {  // 4. the least recently used block that is large enough: wait for it on the device
    for (auto& sl : h->bin_slots)
      if (!sl.busy && sl.bytes >= need && (!pick || sl.stamp < pick->stamp)) pick = &sl;
    wait = pick != nullptr;
  }
  if (!pick && may_alloc) {  // 5. grow the least recently used block that nobody is enqueueing into
    for (auto& sl : h->bin_slots)
      if (!sl.busy && (!pick || sl.stamp < pick->stamp)) pick = &sl;
    if (pick) {
      // The block's last use must complete before it is freed.  That wait and the reallocation
      // run WITHOUT the lock (the slot is marked busy, so nobody else takes it): other threads
      // keep enqueueing through this handle meanwhile.  Growth is synchronous for the caller —
      // interpn_hip_reserve avoids it.
      pick->busy = true;
      const bool recorded = pick->recorded;
      const hipEvent_t ev = pick->event;
      void* const old = pick->scratch;
      lk.unlock();
      bool freed = false;
      void* fresh = nullptr;
      bool ok = !recorded || hipEventSynchronize(ev) == hipSuccess;
      if (ok) {
        pool_free(h->device, old);
        h->last_probe_word = nullptr;
        freed = true;
        ok = pool_alloc(h->device, &fresh, need) == hipSuccess;
      }
      if (!ok) (void)hipGetLastError();
      lk.lock();
      pick->busy = false;
      if (!ok) {
        if (freed) { pick->scratch = nullptr; pick->bytes = 0; pick->recorded = false; }  // else the old block stays as it was
        pick->totals_clean = false;
        pick->sweep_clean = false;
        *why = INTERPN_HIP_WHY_ALLOC_FAILED;
        return nullptr;
      }
      pick->scratch = fresh;
      pick->bytes = need;
      pick->totals_clean = false;
        pick->sweep_clean = false;
      pick->recorded = false;
      h->scratch_allocs.fetch_add(1);
    }
  }
  if (!pick) { *why = INTERPN_HIP_WHY_NO_SCRATCH; return nullptr; }
  if (wait && pick->recorded && pick->last_stream != stream &&
      hipStreamWaitEvent(stream, pick->event, 0) != hipSuccess) { (void)hipGetLastError(); *why = INTERPN_HIP_WHY_NO_SCRATCH; return nullptr; }
  if (!wait && pick->recorded && pick->last_stream != stream) (void)hipStreamWaitEvent(stream, pick->event, 0);  // complete already: free
  pick->busy = true;
  pick->stamp = ++h->bin_uses;
  return pick;
}

// Binned evaluation of the tiled multicubic kernels on device-resident points (interpn_host.h).
// Returns -1 when the path does not apply or cannot be taken right now (`*why` says which; the
// caller then launches the kernel on the points as they are), otherwise a status.  Chosen
// automatically for 4-D grids whose tile table is far beyond the L2 and batches large enough to
// pay for the sorting launches (cfg4: 2.8 -> 1.4 ms per 1e7 points; from about 5e5 points on;
// 3-D grids lose: 4 lines per point are cheaper than sorting them); `binned` = 1 forces it for
// N = 2..4 (tests), 0 turns it off.  Not taken while the stream is being captured into a graph.
int eval_device_binned(interpn_hip_interp* h, const void* const* obs, void* out, size_t npoints, hipStream_t stream,
                       unsigned flags, int* why) {
  const GridDesc& g = h->desc;
  *why = INTERPN_HIP_WHY_NONE;
  const int applies = binned_applies(g, npoints);
  if (applies < 2) { *why = applies ? INTERPN_HIP_WHY_SMALL_OR_OFF : INTERPN_HIP_WHY_NONE; return -1; }
  // The table the sorted points are evaluated on: the handle's own when it is the fully
  // overlapped one (or the evaluation is forced), else the second, fully overlapped table 4-D
  // handles keep for this purpose (maybe_build_cubic_tiles).
  const bool main11 = g.brick_step[0] == 1 && g.brick_step[1] == 1;
  const bool second = !main11 && g.bricks11 != nullptr;
  GridDesc second_desc;
  const GridDesc* use = &g;
  if (second) {
    second_desc = g;
    second_desc.bricks = g.bricks11;
    second_desc.brick_step[0] = second_desc.brick_step[1] = 1;
    second_desc.brick_nb[0] = g.bricks11_nb[0];
    second_desc.brick_nb[1] = g.bricks11_nb[1];
    use = &second_desc;
  }
  BinPlan plan;
  bool column3_go = false;
  // Column evaluation (cubic_column.h): 4-D regular grids whose (k, l) column of tiles fits the LDS
  // and whose (i, j) cells fit one bin each — cfg4.  The sorted points of a cell are then
  // evaluated out of LDS instead of 16 L2 lines per point.
  bool column = g.cfg.column != 0 && (second || main11) && cubic_column_applies(*use);
  const bool column3 = g.ndims == 3 && g.cfg.column != 0 && (second || main11) && cubic3_column_applies(*use);
  if (g.ndims == 3 && g.cfg.binned < 0 && !column3) { *why = INTERPN_HIP_WHY_NONE; return -1; }
  {
    // automatic mode: a part's rows must be mostly full and its column fills amortised — from about
    // three rows of the workgroup's lanes per (class pair) bin on (768 threads: 2304 points; 32^4,
    // round 4: 2e6 points 0.30 against 0.30 ms for the tiled kernel on sorted points, 3e6 0.39-0.42
    // against 0.43-0.47, 4e6 0.46-0.51 against 0.59-0.64, 2e7 1.96-2.02 against 2.66-2.69)
    const size_t slice_max0 = bin_slice_points(g);
    const size_t per_slice = npoints < slice_max0 ? npoints : slice_max0;
    ColumnPlan cp0;
    // ... and every K-range phase of a part must still hand each wave a row: one workgroup's lanes per bin
    // and phase (64^4, 1e7 points: 2520 points per bin in six phases, 2.62 ms against 2.17 for the tiled kernel
    // on the sorted points; 48^4: 4527 points in three phases, 1.14 against 1.65)
    // With coefficient columns (second session) a one-phase column pays from one row of the workgroup's
    // lanes per bin on (32^4: 5e5 points 0.143 against 0.164 ms in place, 1e6 0.163 against 0.167 sorted + tiled,
    // 2e6 0.218 against 0.286; 24^4: 1e6 0.145 against 0.160; profiles/r04_column_threshold.txt).
    size_t per_bin_min = 2304;
    if (column && cubic_column_plan(*use, &cp0)) {
      const size_t rows = cp0.nphase == 1 ? 1 : (cp0.nphase > 3 ? (size_t)cp0.nphase : 3);
      per_bin_min = rows * (size_t)cp0.threads;
    }
    if (g.cfg.column < 0 && per_slice < per_bin_min * (size_t)(g.n[0] - 1) * (size_t)(g.n[1] - 1)) column = false;
  }
  {
    unsigned nbt[2];
    size_t tbytes = 0;  // of the table the sorted points will be evaluated on
    if (second || main11) cubic_tile_geometry(g, 1, 1, nbt, &tbytes);
    else cubic_tile_geometry(g, g.brick_step[0], g.brick_step[1], nbt, &tbytes);
    bool have_plan = false;
    if (column3) have_plan = make_bin_plan(g, tbytes, &plan, /*classes=*/true);
    if (column3 && !have_plan && g.cfg.binned < 0) { *why = INTERPN_HIP_WHY_NONE; return -1; }
    if (!have_plan) {
      if (column && !make_bin_plan(g, tbytes, &plan, /*classes=*/true)) column = false;
      if (!column && !make_bin_plan(g, tbytes, &plan)) { *why = INTERPN_HIP_WHY_SMALL_OR_OFF; return -1; }
    }
    column3_go = column3 && have_plan;
  }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); *why = INTERPN_HIP_WHY_CAPTURE; return -1; }
  if (cs != hipStreamCaptureStatusNone) { *why = INTERPN_HIP_WHY_CAPTURE; return -1; }
  size_t slice_max = bin_slice_points(g);
  if (column && column_keys_in_index(*use) && slice_max > kColumnKeySlicePoints) slice_max = kColumnKeySlicePoints;  // 24-bit indices beside the keys
  const size_t slice = npoints < slice_max ? npoints : slice_max;
  interpn_hip_interp::BinSlot* slot =
      take_bin_slot(h, bin_scratch_bytes(g, slice), stream, !(flags & INTERPN_HIP_EVAL_NO_ALLOC), why);
  if (!slot) return -1;
  const size_t elem = g.dtype == kF64 ? 8 : 4;
  hipError_t err = hipSuccess;
  // per-stage timing on request (single-slice evaluations only)
  hipEvent_t* stage = nullptr;
  slot->staged = false;
  if (g.cfg.stage_timing && npoints <= slice) {
    bool okev = true;
    for (hipEvent_t& e : slot->stage)
      if (!e && hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); e = nullptr; okev = false; }
    if (okev) stage = slot->stage;
  }
  for (size_t begin = 0; begin < npoints && err == hipSuccess; begin += slice) {
    const size_t count = npoints - begin < slice ? npoints - begin : slice;
    const void* src[8];
    const void* sorted[8];
    for (int d = 0; d < g.ndims; ++d) src[d] = static_cast<const char*>(obs[d]) + begin * elem;
    const unsigned* index = nullptr;
    char* dst = static_cast<char*>(out) + begin * elem;
    if (column3_go) {  // 3-D: the cell's column of n2 tiles (cubic3_column.h)
      const size_t q3 = cubic3_column_part_points();
      const size_t max_parts3 = 4 * (count / q3) + (size_t)plan.nbins + 1;
      BinExtras extras3;
      err = bin_points(g, plan, src, count, slot->scratch, sorted, &index, stream, &extras3, (unsigned)q3, stage, slot->totals_clean);
      slot->totals_clean = err == hipSuccess;
      slot->sweep_clean = false;
      if (err != hipSuccess) break;
      if (g.dtype == kF64)
        err = launch_cubic3_column<double>(*use, plan, extras3, index, reinterpret_cast<double*>(dst), count, max_parts3, h->first_bad, begin, stream);
      else
        err = launch_cubic3_column<float>(*use, plan, extras3, index, reinterpret_cast<float*>(dst), count, max_parts3, h->first_bad, begin, stream);
      continue;
    }
    if (column) {
      // a bin is cut into equal parts of at most 16 points per thread of the column workgroup (the
      // registers of its local sort); two such workgroups share a CU, the dispatcher hands parts
      // to whichever frees up
      ColumnPlan cplan;
      if (!cubic_column_plan(*use, &cplan)) { err = hipErrorInvalidValue; break; }
      size_t q = cplan.part_points;
      if (g.cfg.column_part > 0 && (size_t)g.cfg.column_part < q) q = (size_t)g.cfg.column_part;
      const size_t max_parts = 4 * (count / q) + (size_t)plan.nbins + 1;  // upper bound (the scan cuts the last bins finer)
      BinExtras extras;
      if (column_keys_in_index(*use) && count <= kColumnKeySlicePoints && cplan.q3 * (g.n[2] - 1) <= 256) {
        extras.key_q3 = cplan.q3;
        extras.key_sh3 = cplan.sh3;
      }
      err = bin_points(g, plan, src, count, slot->scratch, sorted, &index, stream, &extras, (unsigned)q, stage, slot->totals_clean);
      slot->totals_clean = err == hipSuccess;
      slot->sweep_clean = false;
      if (err != hipSuccess) break;
      if (g.dtype == kF64)
        err = launch_cubic_column<double>(*use, plan, extras, index, reinterpret_cast<double*>(dst), count, max_parts, h->first_bad, begin, stream);
      else
        err = launch_cubic_column<float>(*use, plan, extras, index, reinterpret_cast<float*>(dst), count, max_parts, h->first_bad, begin, stream);
      continue;
    }
    err = bin_points(g, plan, src, count, slot->scratch, sorted, &index, stream, nullptr, 0, stage, slot->totals_clean);
    slot->totals_clean = err == hipSuccess;
      slot->sweep_clean = false;
    if (err != hipSuccess) break;
    if (g.dtype == kF64)
      err = launch_cubic_brick<double>(*use, reinterpret_cast<const double* const*>(sorted), reinterpret_cast<double*>(dst), count,
                                       h->first_bad, stream, index, begin);
    else
      err = launch_cubic_brick<float>(*use, reinterpret_cast<const float* const*>(sorted), reinterpret_cast<float*>(dst), count,
                                      h->first_bad, stream, index, begin);
  }
  if (stage && err == hipSuccess && hipEventRecord(stage[4], stream) == hipSuccess) slot->staged = true;
  // Whatever was enqueued — also a sequence cut short by a failure — is followed by the block's
  // event, so that the next user of the block on another stream waits for it.
  {
    std::lock_guard<std::mutex> lk(h->bin_mu);
    if (hipEventRecord(slot->event, stream) == hipSuccess) {
      slot->recorded = true;
    } else {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(stream);  // no event behind the work: make it complete before anyone reuses the block
      slot->recorded = false;
    }
    slot->last_stream = stream;
    slot->busy = false;
    if (err == hipSuccess) {
      h->desc.tag = use->tag;
      h->desc.last_binned = 1;
    }
  }
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return hip_fail(err);
  }
  return INTERPN_HIP_OK;
}

}  // namespace interpn_abi

extern "C" {

int interpn_hip_eval_device_ex(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out, size_t npoints,
                               void* stream, unsigned flags, int* path_taken, int* why_out) {
  if (path_taken) *path_taken = INTERPN_HIP_PATH_IN_PLACE;
  if (why_out) *why_out = INTERPN_HIP_WHY_NONE;
  if (!h || (!obs && nobs)) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if (flags & ~(unsigned)INTERPN_HIP_EVAL_NO_ALLOC) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  int st = validate_obs(h->desc, nullptr, nobs, npoints);
  if (st) return st;
  if (npoints == 0) return INTERPN_HIP_OK;
  if (!out) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  for (size_t i = 0; i < nobs; ++i)
    if (!obs[i]) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  int why = INTERPN_HIP_WHY_NONE;
  st = eval_device_binned(h, obs, out, npoints, static_cast<hipStream_t>(stream), flags, &why);
  if (why_out) *why_out = why;
  if (st > 0) {
    // part of the sequence may be in flight on `stream` without a mark behind it
    std::lock_guard<std::mutex> lk(h->marks_mu);
    h->sync_device_at_destroy = true;
    return st;
  }
  bool swept = false;
  if (st < 0 && sweep_applies(h->desc, npoints)) {  // 3-D f64 multilinear: the sweep kernel for large batches (linear_sweep.h)
    int why_sweep = INTERPN_HIP_WHY_NONE;
    const int ss = eval_device_sweep(h, obs, out, npoints, static_cast<hipStream_t>(stream), flags, &why_sweep);
    // the sweep's reason replaces the sorted path's only where that path had none, or where the sweep's is one of the
    // specific ones (capture, no scratch, allocation failed, misaligned arrays)
    if (why_out && (why == INTERPN_HIP_WHY_NONE || why_sweep > INTERPN_HIP_WHY_SMALL_OR_OFF)) *why_out = why_sweep;
    if (ss > 0) {
      std::lock_guard<std::mutex> lk(h->marks_mu);
      h->sync_device_at_destroy = true;
      return ss;
    }
    swept = ss == 0;
  }
  if (swept) {
    if (path_taken) *path_taken = INTERPN_HIP_PATH_SWEEP;
    h->desc.last_binned = 0;
    h->evals_sweep.fetch_add(1);
  } else if (st < 0) {
    HIP_TRY(launch_any(h->desc, obs, out, npoints, h->first_bad, static_cast<hipStream_t>(stream)));
    if (binned_applies(h->desc, npoints)) {  // handles that can sort: keep the report field honest
      std::lock_guard<std::mutex> lk(h->bin_mu);
      h->desc.last_binned = 0;
    } else {
      h->desc.last_binned = 0;
    }
    h->evals_in_place.fetch_add(1);
  } else {
    if (path_taken) *path_taken = INTERPN_HIP_PATH_BINNED;
    h->evals_binned.fetch_add(1);
  }
  mark_stream(h, static_cast<hipStream_t>(stream));
  return INTERPN_HIP_OK;
}

int interpn_hip_eval_device(interpn_hip_interp* h, const void* const* obs, size_t nobs, void* out, size_t npoints,
                            void* stream) {
  return interpn_hip_eval_device_ex(h, obs, nobs, out, npoints, stream, 0u, nullptr, nullptr);
}

int interpn_hip_stage_ms(interpn_hip_interp* h, double* ms, size_t n) {
  if (!h || !ms || n < 4) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  std::lock_guard<std::mutex> lk(h->bin_mu);
  const interpn_hip_interp::BinSlot* best = nullptr;
  for (const auto& sl : h->bin_slots)
    if (sl.staged && !sl.busy && (!best || sl.stamp > best->stamp)) best = &sl;
  if (!best) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipEventSynchronize(best->stage[4]));
  for (int k = 0; k < 4; ++k) {
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, best->stage[k], best->stage[k + 1]));
    ms[k] = (double)f;
  }
  return INTERPN_HIP_OK;
}

int interpn_hip_reserve(interpn_hip_interp* h, size_t npoints, int nstreams) {
  if (!h || nstreams < 0) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if ((size_t)nstreams > interpn_hip_interp::kMaxBinSlots) nstreams = (int)interpn_hip_interp::kMaxBinSlots;
  const GridDesc& g = h->desc;
  const bool sorts = binned_applies(g, npoints) >= 2, sweeps = sweep_applies(g, npoints) >= 2;
  if (npoints == 0 || nstreams == 0 || (!sorts && !sweeps)) return INTERPN_HIP_OK;  // nothing to provide
  DeviceGuard guard(h->device);
  if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
  const size_t slice_max = bin_slice_points(g);
  // the sorted path's slice, or the sweep kernel's work words (linear_sweep.h: 1.25 KiB per stream)
  const size_t need = sorts ? bin_scratch_bytes(g, npoints < slice_max ? npoints : slice_max) : sweep_work_bytes();
  std::lock_guard<std::mutex> lk(h->bin_mu);
  int have = 0;
  for (auto& sl : h->bin_slots)
    if (sl.bytes >= need) ++have;
  // grow blocks that are too small first (idle ones only), then add new ones
  for (auto& sl : h->bin_slots) {
    if (have >= nstreams) break;
    if (sl.bytes >= need || sl.busy) continue;
    if (sl.recorded) HIP_TRY(hipEventSynchronize(sl.event));
    pool_free(h->device, sl.scratch);
    h->last_probe_word = nullptr;
    sl.scratch = nullptr;
    sl.bytes = 0;
    sl.totals_clean = false;
    sl.sweep_clean = false;
    sl.recorded = false;
    hipError_t e = pool_alloc(h->device, &sl.scratch, need);
    if (e != hipSuccess) { (void)hipGetLastError(); sl.scratch = nullptr; return INTERPN_HIP_ERR_OUT_OF_MEMORY; }
    sl.bytes = need;
    h->scratch_allocs.fetch_add(1);
    ++have;
  }
  while (have < nstreams && h->bin_slots.size() < interpn_hip_interp::kMaxBinSlots) {
    interpn_hip_interp::BinSlot sl;
    HIP_TRY(hipEventCreateWithFlags(&sl.event, hipEventDisableTiming));
    hipError_t e = pool_alloc(h->device, &sl.scratch, need);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(sl.event); return INTERPN_HIP_ERR_OUT_OF_MEMORY; }
    sl.bytes = need;
    h->scratch_allocs.fetch_add(1);
    h->bin_slots.push_back(sl);
    ++have;
  }
  return have >= nstreams ? INTERPN_HIP_OK : INTERPN_HIP_ERR_OUT_OF_MEMORY;
}

}  // extern "C"
