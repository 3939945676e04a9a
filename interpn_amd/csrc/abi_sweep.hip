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
  // 3-D multilinear in automatic mode: the device decides between this kernel and the brick kernel (k_linear_sweep.hip::k_sweep_probe)
  const bool probe = g.method == kLinear && g.ndims == 3 && g.cfg.sweep < 0 && g.cfg.sweep_probe != 0 && g.bricks != nullptr;
  if (probe && err == hipSuccess) err = launch_sweep_probe(g, obs, npoints, slot->scratch, stream);
  if (err == hipSuccess) err = launch_linear_sweep(g, obs, out, npoints, h->first_bad, slot->scratch, stream, probe);
  if (probe && err == hipSuccess) {
    const unsigned* gate = reinterpret_cast<const unsigned*>(static_cast<const unsigned char*>(slot->scratch) + sweep_probe_word_offset());
    const KernelTag primary = g.tag;  // the handle reports the sweep kernel (which of the pair ran is known on the device only: option sweep_probe_took_brick)
    err = g.dtype == kF64 ? launch_linear_brick<double>(g, reinterpret_cast<const double* const*>(obs), static_cast<double*>(out), npoints, h->first_bad, stream, gate)
                          : launch_linear_brick<float>(g, reinterpret_cast<const float* const*>(obs), static_cast<float*>(out), npoints, h->first_bad, stream, gate);
    g.tag = primary;
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
