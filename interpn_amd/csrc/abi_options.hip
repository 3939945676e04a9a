// Per-handle options (latched from INTERPN_HIP_* once, at creation; nothing on the launch path
// reads the environment), and what a handle reports about itself.  (C ABI internals, see abi_internal.h.)
#include "abi_internal.h"

using namespace interpn;
using namespace interpn_abi;

namespace interpn_abi {

// Options by name (interpn_hip_set_option / interpn_hip_get_option, and the INTERPN_HIP_<NAME>
// environment variables latched at creation).  Returns false for an unknown name or a value out
// of range; `set == false` reads.
bool option_access(LaunchConfig& c, const char* name, long long* value, bool set) {
  struct Opt { const char* name; int* field; long long lo, hi; };
  const Opt opts[] = {
      {"blocks_per_cu", &c.blocks_per_cu, 1, 65536},
      {"iters_per_block", &c.iters_per_block, 0, 65536},
      {"ppl", &c.ppl, 0, 2},
      {"axis_regs", &c.axis_regs, -1, 2},
      {"force_generic", &c.force_generic, 0, 1},
      {"generic_runtime", &c.generic_runtime, 0, 1},
      {"generic_vec", &c.generic_vec, -1, 1},
      {"persistent", &c.persistent, 0, 1},
      {"axis_lds_kb", &c.axis_lds_kb, -1, 60},
      {"binned", &c.binned, -1, 1},
      {"deal", &c.deal, 0, 1},
      {"bin_slice_log2", &c.bin_slice_log2, 16, 27},
      {"column", &c.column, -1, 1},
      {"column_part", &c.column_part, 0, 1 << 20},
      {"column_threads", &c.column_threads, 256, 1024},
      {"column_groups", &c.column_groups, 1, 2},
      {"column_cpp", &c.column_cpp, 0, 1 << 20},
      {"column_coef", &c.column_coef, 0, 1},
      {"column_pad", &c.column_pad, -1, 1},
      {"hist_wgs_per_cu", &c.hist_wgs_per_cu, 0, 64},
      {"column_keys", &c.column_keys, 0, 1},
      {"column_tail", &c.column_tail, 0, 255},
      {"scatter_staged", &c.scatter_staged, 0, 1},
      {"bin_scramble", &c.bin_scramble, 0, 1},
      {"stage_timing", &c.stage_timing, 0, 1},
      {"axis_records", &c.axis_records, 0, 1},
      {"sweep", &c.sweep, -1, 2},
      {"sweep_period", &c.sweep_period, 0, 1000000},
      {"sweep_probe", &c.sweep_probe, 0, 2},
      {"gated_iters", &c.gated_iters, 0, 4096},
      {"finish_kernel", &c.finish_kernel, 0, 1},
  };
  if (!name || !value) return false;
  if (!strcmp(name, "host_chunk")) {
    if (set) {
      if (*value < 0) return false;
      c.host_chunk = *value;
    } else {
      *value = c.host_chunk;
    }
    return true;
  }
  if (!strcmp(name, "debug_stamps")) {  // a device address (measurement aid, cubic_column.h)
    if (set) c.debug_stamps = *value;
    else *value = c.debug_stamps;
    return true;
  }
  if (!strcmp(name, "debug_stamps_bytes")) {  // capacity of the stamp buffer in bytes (checked at launch)
    if (set) {
      if (*value < 0) return false;
      c.debug_stamps_bytes = *value;
    } else {
      *value = c.debug_stamps_bytes;
    }
    return true;
  }
  for (const Opt& o : opts) {
    if (strcmp(name, o.name)) continue;
    if (set) {
      if (*value < o.lo || *value > o.hi) return false;
      *o.field = (int)*value;
    } else {
      *value = *o.field;
    }
    return true;
  }
  return false;
}

// The environment is read here, once per handle, and nowhere on the launch path.
void latch_env(LaunchConfig& c) {
  static const char* const names[] = {"blocks_per_cu", "iters_per_block", "ppl", "axis_regs", "force_generic",
                                      "generic_runtime", "generic_vec", "persistent", "axis_lds_kb", "host_chunk", "binned", "deal",
                                      "bin_slice_log2", "column", "column_part", "column_threads", "column_groups", "column_cpp", "column_coef", "column_pad", "hist_wgs_per_cu", "column_keys", "column_tail", "scatter_staged", "axis_records", "bin_scramble", "sweep", "sweep_period", "sweep_probe", "gated_iters", "finish_kernel"};
  for (const char* nm : names) {
    char var[64] = "INTERPN_HIP_";
    size_t k = strlen(var);
    for (const char* q = nm; *q && k + 1 < sizeof(var); ++q) var[k++] = (char)toupper((unsigned char)*q);
    var[k] = 0;
    const char* env = getenv(var);
    if (!env || !*env) continue;
    char* end = nullptr;
    long long v = strtoll(env, &end, 0);  // decimal, 0x.. or 0.. (INTERPN_HIP_COLUMN_TAIL is documented in hex)
    if (end == env || *end != 0) continue;  // not a number: ignored
    (void)option_access(c, nm, &v, true);  // out-of-range values are ignored, as before
  }
}

}  // namespace interpn_abi

extern "C" {

int interpn_hip_set_blocks_per_cu(interpn_hip_interp* h, int blocks_per_cu) {
  if (!h || blocks_per_cu < 1 || blocks_per_cu > 65536) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  h->desc.cfg.blocks_per_cu = blocks_per_cu;
  return INTERPN_HIP_OK;
}

int interpn_hip_set_option(interpn_hip_interp* h, const char* name, long long value) {
  if (!h || !name) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if (!strcmp(name, "fma")) {  // the flavour is read at every launch; tables do not depend on it
    if (value != 0 && value != 1) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
    h->desc.fma = (int)value;
    return INTERPN_HIP_OK;
  }
  return option_access(h->desc.cfg, name, &value, true) ? INTERPN_HIP_OK : INTERPN_HIP_ERR_INVALID_ARGUMENT;
}

int interpn_hip_get_option(const interpn_hip_interp* h, const char* name, long long* value) {
  if (!h || !name || !value) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  if (!strcmp(name, "last_binned")) {  // read-only: did the most recent device-pointer evaluation sort its points first?
    *value = h->desc.last_binned;
    return INTERPN_HIP_OK;
  }
  if (!strcmp(name, "fma")) { *value = h->desc.fma; return INTERPN_HIP_OK; }
  // read-only: how the handle's rectilinear axes will be searched (0 no records, 1 full, 2 compact) and what is staged
  if (!strcmp(name, "axis_rec_mode")) { *value = h->desc.axis_rec_bytes ? (h->desc.axis_rec_compact ? 2 : 1) : 0; return INTERPN_HIP_OK; }
  if (!strcmp(name, "axis_rec_bytes")) { *value = h->desc.axis_rec_bytes; return INTERPN_HIP_OK; }
  if (!strcmp(name, "axis_image_bytes")) { *value = h->desc.axis_image_bytes; return INTERPN_HIP_OK; }
  {  // read-only: the device the handle lives on and the thresholds derived from it (interpn_host.h::Thresholds)
    const LaunchConfig& c = h->desc.cfg;
    const Thresholds t = thresholds(c);
    const struct { const char* nm; long long v; } ro[] = {
        {"dev_num_cus", c.num_cus}, {"dev_num_xcds", c.num_xcds}, {"dev_l2_bytes", c.l2_bytes}, {"dev_lds_per_cu", c.lds_per_cu},
        {"dev_lds_per_wg", c.lds_per_wg}, {"thr_table_l2_sized", (long long)t.table_l2_sized}, {"thr_table_l2_share", (long long)t.table_l2_share},
        {"thr_binned_table_min", (long long)t.binned_table_min}, {"thr_binned_points_min", (long long)t.binned_points_min},
        {"thr_bin_table_share", (long long)t.bin_table_share}, {"thr_axis_lds", (long long)t.axis_lds},
        {"thr_axis_lds_wide", (long long)t.axis_lds_wide}, {"thr_column_lds", (long long)t.column_lds}};
    for (const auto& r : ro)
      if (!strcmp(name, r.nm)) { *value = r.v; return INTERPN_HIP_OK; }
  }
  if (!strncmp(name, "col_", 4)) {  // read-only: how the column evaluation would run on this grid with the present options (k_cubic_column.hip)
    ColumnPlan cp;
    const bool ok = h->desc.method == kCubic && h->desc.ndims == 4 && cubic_column_plan(h->desc, &cp);
    const struct { const char* nm; long long v; } ro[] = {
        {"col_applies", ok ? 1 : 0}, {"col_nphase", ok ? cp.nphase : 0}, {"col_cpp", ok ? cp.cpp : 0}, {"col_pitch", ok ? (long long)cp.pitch : 0},
        {"col_lds_bytes", ok ? (long long)cp.lds_bytes : 0}, {"col_part_points", ok ? (long long)cp.part_points : 0},
        {"col_perm_pad", ok ? (long long)cp.perm_pad : 0}, {"col_q3", ok ? cp.q3 : 0}, {"col_sh3", ok ? cp.sh3 : 0},
        {"col_threads", ok ? cp.threads : 0}, {"col_groups", ok ? cp.groups : 0}};
    for (const auto& r : ro)
      if (!strcmp(name, r.nm)) { *value = r.v; return INTERPN_HIP_OK; }
    return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  }
  if (!strcmp(name, "evals_binned")) { *value = h->evals_binned.load(); return INTERPN_HIP_OK; }
  if (!strcmp(name, "evals_in_place")) { *value = h->evals_in_place.load(); return INTERPN_HIP_OK; }
  if (!strcmp(name, "evals_sweep")) { *value = h->evals_sweep.load(); return INTERPN_HIP_OK; }
  if (!strcmp(name, "sweep_table_bytes")) { *value = h->desc.sweep_bricks ? (long long)h->desc.sweep_table_bytes : 0; return INTERPN_HIP_OK; }
  if (!strcmp(name, "sweep_probe_took_brick")) {  // read-only, synchronises the device: 1 / 0 = what the last gated launch's sample decided, -1 = the last sweep launch was not gated
    *value = -1;
    if (h->last_probe_word) {
      DeviceGuard guard(h->device);
      if (!guard.ok()) return INTERPN_HIP_ERR_NO_DEVICE;
      unsigned w = 0;
      if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&w, h->last_probe_word, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return INTERPN_HIP_ERR_INVALID_ARGUMENT; }
      *value = w ? 1 : 0;
    }
    return INTERPN_HIP_OK;
  }
  if (!strcmp(name, "sweep_probe_streak")) {  // read-only: samples in a row whose verdict was "unordered", as far as the host knows (thinned policy)
    interpn_hip_interp* hm = const_cast<interpn_hip_interp*>(h);
    std::lock_guard<std::mutex> lk(hm->bin_mu);
    *value = hm->probe_streak;
    return INTERPN_HIP_OK;
  }
  if (!strcmp(name, "sweep_cell")) { *value = h->desc.sweep_bricks ? h->desc.sweep_cell : 0; return INTERPN_HIP_OK; }  // 0: 2 x 2 x KW bricks, 2: 2 x 4 x 4 (f32)
  if (!strcmp(name, "dev_pci")) {  // read-only: (domain << 16) | (bus << 8) | device of the GPU the handle lives on; -1 if unknown
    int dom = 0, bus = 0, dv = 0;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, h->device) != hipSuccess ||
        hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, h->device) != hipSuccess ||
        hipDeviceGetAttribute(&dv, hipDeviceAttributePciDeviceId, h->device) != hipSuccess) {
      (void)hipGetLastError();
      *value = -1;
    } else {
      *value = ((long long)dom << 16) | ((long long)(bus & 0xFF) << 8) | (long long)(dv & 0xFF);
    }
    return INTERPN_HIP_OK;
  }
  if (!strcmp(name, "sweep_layout")) { *value = h->desc.sweep_bricks ? h->desc.sweep_step[0] * 10 + h->desc.sweep_step[1] : 0; return INTERPN_HIP_OK; }
  if (!strcmp(name, "scratch_allocs")) { *value = h->scratch_allocs.load(); return INTERPN_HIP_OK; }
  if (!strcmp(name, "scratch_bytes")) {
    interpn_hip_interp* hm = const_cast<interpn_hip_interp*>(h);
    std::lock_guard<std::mutex> lk(hm->bin_mu);
    long long tot = 0;
    for (const auto& sl : hm->bin_slots) tot += (long long)sl.bytes;
    *value = tot;
    return INTERPN_HIP_OK;
  }
  LaunchConfig c = h->desc.cfg;
  return option_access(c, name, value, false) ? INTERPN_HIP_OK : INTERPN_HIP_ERR_INVALID_ARGUMENT;
}

// "interpn::k_linear_brick<double, 3, false, true, 1, 2, 2, 0>" — rocprofv3's spelling of the
// instantiation, without the return type and the argument list.
int interpn_hip_kernel_name(const interpn_hip_interp* h, char* buf, size_t buflen) {
  if (!h || !buf || buflen == 0) return INTERPN_HIP_ERR_INVALID_ARGUMENT;
  const KernelTag t = h->desc.tag;
  if (!t.name) {
    buf[0] = 0;
    return INTERPN_HIP_OK;
  }
  std::string out = std::string("interpn::") + t.name + "<" + (h->desc.dtype == kF64 ? "double" : "float");
  for (int k = 0; k < t.nargs; ++k) {
    out += ", ";
    if (t.bool_mask & (1u << k)) out += t.args[k] ? "true" : "false";
    else out += std::to_string(t.args[k]);
  }
  out += ">";
  snprintf(buf, buflen, "%s", out.c_str());
  return INTERPN_HIP_OK;
}

// Bytes of the re-laid grid copy the handle keeps (0 = kernels read the C-ordered `vals`), and
// its layout steps; for reports.
size_t interpn_hip_table_bytes(const interpn_hip_interp* h, int* step_i, int* step_j) {
  if (!h || !h->desc.bricks) return 0;
  const GridDesc& g = h->desc;
  if (step_i) *step_i = g.brick_step[0];
  if (step_j) *step_j = g.brick_step[1];
  size_t bytes = 0;
  unsigned nb[3];
  unsigned nb4[4];
  if (g.method == kCubic) cubic_tile_geometry(g, g.brick_step[0], g.brick_step[1], nb, &bytes);
  else if (g.ndims == 1) bytes = records1_bytes(g, g.rec1_buckets);
  else if (g.ndims == 2) brick2_geometry(g, nb, &bytes);
  else if (g.brick_cell == 2) brick_j4_geometry(g, nb, &bytes);
  else if (g.brick_cell) brick_cell_geometry(g, nb4, &bytes);
  else brick_geometry(g, g.brick_step[0], g.brick_step[1], nb, &bytes);
  return bytes;
}

}  // extern "C"
