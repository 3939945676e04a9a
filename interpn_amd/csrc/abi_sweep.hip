// Sweep evaluation of large device-resident 3-D multilinear batches (linear_sweep.h): the scratch
// block with the launch's work words, per stream, and the decision to take the path.  (C ABI
// internals, see abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

namespace interpn_abi {

// Returns -1 when the path does not apply or cannot be taken right now (`*why` says which; the
// caller then launches the brick kernel on the points as they are), otherwise a status.  The work
// words (1.25 KiB: round counters, the period measurement) live in a scratch block of the handle,
// taken per stream exactly like the sorted path's blocks (take_bin_slot): two streams never share
// one in flight.  Not taken while the stream is being captured into a graph (the block's event
// cannot be recorded there), for batches that give a wave fewer than four rounds, or for streams
// that are not 16-byte aligned.
int eval_device_sweep(interpn_hip_interp* h, const void* const* obs, void* out, size_t npoints, hipStream_t stream,
                      unsigned flags, int* why) {
  const GridDesc& g = h->desc;
  *why = INTERPN_HIP_WHY_NONE;
  const int applies = sweep_applies(g, npoints);
  if (applies < 2) { *why = applies ? INTERPN_HIP_WHY_SMALL_OR_OFF : INTERPN_HIP_WHY_NONE; return -1; }
  if (reinterpret_cast<uintptr_t>(out) % 16) { *why = INTERPN_HIP_WHY_MISALIGNED; return -1; }
  for (int d = 0; d < g.ndims; ++d)
    if (reinterpret_cast<uintptr_t>(obs[d]) % 16) { *why = INTERPN_HIP_WHY_MISALIGNED; return -1; }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); *why = INTERPN_HIP_WHY_CAPTURE; return -1; }
  if (cs != hipStreamCaptureStatusNone) { *why = INTERPN_HIP_WHY_CAPTURE; return -1; }
  interpn_hip_interp::BinSlot* slot = take_bin_slot(h, sweep_work_bytes(), stream, !(flags & INTERPN_HIP_EVAL_NO_ALLOC), why);
  if (!slot) return -1;
  hipError_t err = hipSuccess;
  // first use of the block by this path, or the sorted path has used it since: reset the work words (a complete launch leaves
  // its counters zero and its measured period in place; that period word is one of the sort's bin counters)
  if (!slot->sweep_clean) err = hipMemsetAsync(slot->scratch, 0, sweep_work_bytes(), stream);
  // automatic mode: the device decides between the sweep kernel and the one-pass kernel (k_linear_sweep.hip::k_sweep_probe:
  // a sampling kernel in front, verdict 1 / 0 in the scratch block); both launches work from private copies of the
  // description that carry the gate (the handle's own is shared by threads)
  bool probe = sweep_probe_applies(g) && npoints >= 256u * 64u;
  unsigned* host_word = nullptr;
  unsigned seq = 0;
  bool one_pass_only = false;
  if (probe && g.cfg.sweep_probe == 2) {
    // Thinned-out sampling: a handle whose last three samples all said "unordered" is sampled on every 16th automatic
    // launch only (the sample and the gated launch behind the sweep kernel cost ~1.5 % of a 1e8-point launch); one
    // "coherent" verdict brings every launch's sample back.  Whichever kernel runs, the results are the same bits.
    std::lock_guard<std::mutex> lk(h->bin_mu);
    if (!h->probe_host) {
      void* dp = nullptr;
      if (pool_take_pinned_word(h->device, &h->probe_host) == hipSuccess && h->probe_host &&
          hipHostGetDevicePointer(&dp, h->probe_host, 0) == hipSuccess && dp) {
        *(volatile unsigned long long*)h->probe_host = 0;
        h->probe_host_dev = static_cast<unsigned*>(dp);
      } else {
        (void)hipGetLastError();
        h->probe_host_dev = nullptr;  // no host view of the verdicts: every launch is sampled
      }
    }
    if (h->probe_host_dev) {
      const unsigned w = (unsigned)*(volatile unsigned long long*)h->probe_host;
      if ((w >> 1) != h->probe_seen && (w >> 1) != 0) {
        h->probe_seen = w >> 1;
        h->probe_streak = (w & 1u) ? 0 : h->probe_streak + 1;
        h->probe_streak_coherent = (w & 1u) ? h->probe_streak_coherent + 1 : 0;
      }
      if (h->probe_streak >= 3 && h->probe_skipped < 15) {
        ++h->probe_skipped;
        probe = false;
      } else if (h->probe_streak_coherent >= 3 && h->probe_skipped < 15) {
        ++h->probe_skipped;  // ... and the mirror image: the last three samples all said "coherent" — the one-pass kernel alone
        probe = false;
        one_pass_only = true;
      } else {
        h->probe_skipped = 0;
        h->probe_seq = h->probe_seq >= 0x7FFFFFFEu ? 1u : h->probe_seq + 1u;
        host_word = h->probe_host_dev;
        seq = h->probe_seq;
      }
    }
  }
  if (probe && err == hipSuccess) err = launch_sweep_probe(g, obs, npoints, slot->scratch, stream, host_word, seq);
  if (one_pass_only) {
    if (err == hipSuccess) {
      GridDesc gb = g;
      gb.launch_fat = true;  // (the workgroup shape of the gated launch: what coherent batches run best in)
      err = launch_any(gb, obs, out, npoints, h->first_bad, stream);
      g.tag = gb.tag;
    }
  } else if (err == hipSuccess) {
    if (probe) {
      GridDesc gs = g;
      gs.sweep_gated = true;
      err = launch_linear_sweep(gs, obs, out, npoints, h->first_bad, slot->scratch, stream);
      g.tag = gs.tag;  // the handle reports the sweep kernel (which of the pair ran is known on the device only: option sweep_probe_took_brick)
    } else {
      err = launch_linear_sweep(g, obs, out, npoints, h->first_bad, slot->scratch, stream);
    }
  }
  if (probe && err == hipSuccess) {
    GridDesc gb = g;
    gb.launch_gate = reinterpret_cast<const unsigned*>(static_cast<const unsigned char*>(slot->scratch) + sweep_probe_word_offset());
    err = launch_any(gb, obs, out, npoints, h->first_bad, stream);
  }
  h->last_probe_word = probe ? static_cast<const unsigned char*>(slot->scratch) + sweep_probe_word_offset() : nullptr;
  slot->sweep_clean = err == hipSuccess;
  slot->totals_clean = false;
  {
    std::lock_guard<std::mutex> lk(h->bin_mu);
    if (hipEventRecord(slot->event, stream) == hipSuccess) {
      slot->recorded = true;
    } else {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(stream);
      slot->recorded = false;
    }
    slot->last_stream = stream;
    slot->busy = false;
    slot->staged = false;
  }
  if (err != hipSuccess) {
    (void)hipGetLastError();
    return hip_fail(err);
  }
  return INTERPN_HIP_OK;
}

}  // namespace interpn_abi
