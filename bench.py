#!/usr/bin/env python3
"""Headline benchmark: Mpoints/s (f64) of 3-D multilinear-regular interpolation on random
observation points, plus the achieved fraction of the HBM roofline for the kernel and the CPU
oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

* The headline (`value`, `roofline`) is BASELINE.json configs[1] at every N: 3-D multilinear-regular,
  64^3 f64 grid, 1e8 random obs PER GPU (weak scaling over a fixed grid, as north_star asks), so
  value(N) / value(1) compares identical per-GPU work.
* `--gpus N` (N > 1): observation points sharded contiguously, the grid generated on rank 0 and
  replicated by ONE RCCL broadcast, no collective in the timed loop.  The record then carries a
  second block `cfg5` = BASELINE.json configs[4] measured by the same protocol: 128^3 grid, 1e8 obs
  per rank (8e8 over 8 GPUs), with per-rank kernel times and the same workload on one GPU alone.
  When no launcher has set WORLD_SIZE this process only SPAWNS the N ranks (fresh child processes,
  created before anything touches a GPU here) and relays rank 0's line; under
  `python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.
* `--workload cfg5` makes the 128^3 grid the headline instead (e.g. its shard on one GPU).

A step is one pass of the hot path over one batch: one `interpn_hip_eval_device` launch over the
rank's device-resident points followed by the status check (`interpn_hip_finish`).  Inputs are
synthetic: axes linspace(-1,1,n), vals U(-1,1), obs i.i.d. uniform over the
grid extent, unordered.  EXACTLY K steps are timed between barrier + synchronize pairs; rank 0
prints ONE JSON line.  At N = 1 the line also carries
  roofline.sustained   the same launch repeated for >= 0.5 s (K may be as small as 20 = 26 ms),
  roofline.ablation    stream-only / gather-only variants of the same kernel source (the L2-bound
                       ceiling argument of DESIGN.md section 4.1, reproducible by the driver),
  configs[]            every single-GPU BASELINE configuration (cfg2, cfg3, cfg4 with both
                       linearize flags, the cfg5 shard): kernel time, roofline fraction, the kernel
                       name queried from the handle, and a bitwise check of sampled points against
                       the CPU oracle (outside every timed region),
  cpu_baseline         the oracle (a port: the Rust reference cannot be built here) on host cores.
"""

from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
METRIC = "Mpoints/sec (f64) per GPU + achieved HBM GB/s vs roofline, 3D linear-regular"
WORKLOADS = {
    # name: (grid points per axis, BASELINE.json entry)
    "cfg2": (64, "BASELINE configs[1]"),
    "cfg5": (128, "BASELINE configs[4] shard"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="auto", choices=["auto", "cfg2", "cfg5"],
                    help="headline grid: auto = cfg2 (64^3) at every N; cfg5 = 128^3")
    ap.add_argument("--points", type=int, default=100_000_000, help="observation points per GPU")
    ap.add_argument("--grid", type=int, default=0, help="grid points per axis (0 = the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="points for the CPU baseline (0 = auto, ~10-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration table (N = 1)")
    ap.add_argument("--no-ablate", action="store_true", help="skip the stream-only / gather-only ablation (N = 1)")
    ap.add_argument("--no-live-traffic", action="store_true", help="N = 1: skip the two rocprofv3 --pmc child passes that measure roofline.traffic in this run")
    ap.add_argument("--no-cfg5", action="store_true", help="N > 1: skip the second block on the 128^3 grid (BASELINE configs[4])")
    ap.add_argument("--sustain-seconds", type=float, default=0.5)
    ap.add_argument("--spinup-seconds", type=float, default=0.15,
                    help="untimed launches in front of the W warm-up steps, until the GPU clocks have ramped (0 = none)")
    # Test aids for a 1-GPU box: run the multi-rank control flow with every rank on cuda:0 over
    # gloo (RCCL refuses two ranks on one device).  The driver never passes these.
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-device", action="store_true")
    # 1-GPU check of the RCCL calls themselves: initialise the process group (backend nccl = RCCL)
    # even with one rank, so that broadcast / all_gather / all_reduce / barrier run on device tensors.
    ap.add_argument("--force-dist", action="store_true")
    ap.add_argument("--spawn-timeout", type=float, default=1500.0)
    # CPU test of the spawn / rendezvous path only (tests/test_bench_spawn.py): no GPU work, every
    # rank joins a gloo group and rank 0 prints who took part.  Not a benchmark.
    ap.add_argument("--dry-run", action="store_true")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# N > 1 without a launcher: spawn the ranks.  Nothing in this function (or before it) touches HIP.
def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv) -> int:
    """Start `args.gpus` child processes of this script, one per GPU, as a launcher would
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); rank 0's stdout (the JSON line) is ours.
    Returns the largest exit code.  Children are fresh processes made by fork+exec from a parent
    that has not initialised the GPU."""
    world = args.gpus
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0",
                    "INTERPN_BENCH_SPAWNED": "1"})
        out = None if r == 0 else subprocess.DEVNULL  # only rank 0 prints the record
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    deadline = time.time() + args.spawn_timeout
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is not None:
                pending.remove(p)
                rc = max(rc, abs(code))
                if code != 0:  # one rank failed: the others would wait in a collective for ever
                    deadline = min(deadline, time.time() + 20.0)
        if pending and time.time() > deadline:
            for p in pending:
                p.kill()  # exact PIDs we started
            rc = max(rc, 124)
            break
        time.sleep(0.05)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass
    return rc


# ---------------------------------------------------------------------------------------------
def make_axes(n, ndims, rectilinear, rng):
    g = np.linspace(-1.0, 1.0, n)
    step = g[1] - g[0]
    grids = []
    for _ in range(ndims):
        gg = g.copy()
        if rectilinear:  # interior nodes jittered by up to a quarter step
            j = (rng.random(n) - 0.5) * 0.5 * step
            j[0] = j[-1] = 0.0
            gg = gg + j
            assert np.all(np.diff(gg) > 0)
        grids.append(gg)
    return grids, step


def time_launches(torch, it, obs, out, launches=0, seconds=0.0, finish_each=False):
    """Run `launches` evaluations (or as many as fill `seconds`), each bracketed by HIP events on
    the launch stream (torch's current stream is what the ABI is given).  Returns per-launch ms."""
    ms = []
    t_end = time.perf_counter() + seconds
    pending = []
    n = 0
    while (launches and n < launches) or (not launches and (time.perf_counter() < t_end or n < 8)):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        it.eval_tensors(obs, out)
        b.record()
        if finish_each:
            it.finish()
        pending.append((a, b))
        n += 1
        if len(pending) >= 64 or finish_each:
            if not finish_each:
                it.finish()
            ms += _elapsed(pending)
            pending = []
    it.finish()
    ms += _elapsed(pending)
    return ms


def _elapsed(pairs):
    """Event times of completed launches.  `finish()` has seen the status word, so the GPU is done with everything in front
    of it — but the runtime's own record of an event may lag that by microseconds (seen once with the status word
    delivered by a kernel): waiting on the (already passed) end event costs nothing and makes the read safe."""
    out = []
    for x, y in pairs:
        y.synchronize()
        out.append(x.elapsed_time(y))
    return out


def oracle_eval(spec, sub):
    """CPU oracle on a small sample (the checker; never the thing measured on the GPU side)."""
    from oracle import pyoracle

    want = np.zeros(sub[0].size)
    if spec["method"] == "linear" and spec["kind"] == "regular":
        pyoracle.linear_regular(spec["dims"], spec["starts"], spec["steps"], spec["vals"], sub, want)
    elif spec["method"] == "linear":
        pyoracle.linear_rectilinear(spec["grids"], spec["vals"], sub, want)
    elif spec["kind"] == "regular":
        pyoracle.cubic_regular(spec["dims"], spec["starts"], spec["steps"], spec["vals"], spec["linearize"], sub, want)
    else:
        pyoracle.cubic_rectilinear(spec["grids"], spec["vals"], spec["linearize"], sub, want)
    return want


def build_spec(method, kind, n, ndims, linearize, seed):
    rng = np.random.default_rng(seed)
    grids, step = make_axes(n, ndims, kind == "rectilinear", rng)
    return {"method": method, "kind": kind, "n": n, "ndims": ndims, "linearize": bool(linearize),
            "dims": [n] * ndims, "starts": np.full(ndims, -1.0), "steps": np.full(ndims, step), "grids": grids,
            "vals": rng.uniform(-1.0, 1.0, n**ndims)}


def make_interp(interpn_amd, spec, device, vals=None):
    vals = spec["vals"] if vals is None else vals
    if spec["kind"] == "regular":
        return interpn_amd.Interpolator.regular(spec["method"], spec["dims"], spec["starts"], spec["steps"], vals,
                                                spec["linearize"], device, np.float64)
    return interpn_amd.Interpolator.rectilinear(spec["method"], spec["grids"], vals, spec["linearize"], device,
                                                np.float64)


def config_row(torch, interpn_amd, name, spec, obs, out, device, seconds, check_points=100_000):
    """One row of the per-configuration table: kernel time over >= `seconds` of launches, roofline
    fraction, kernel name from the handle, sampled bitwise check against the oracle."""
    it = make_interp(interpn_amd, spec, device)
    P = obs[0].numel()
    for _ in range(3):
        it.eval_tensors(obs, out)
    it.finish()
    ms = time_launches(torch, it, obs, out, seconds=seconds)
    kernel_ms = float(np.mean(ms))
    bpp = 8 * (spec["ndims"] + 1)
    achieved = P * bpp / (kernel_ms * 1e-3) / 1e9
    tbytes, si, sj = it.table_layout()
    gen = torch.Generator(device=obs[0].device)
    gen.manual_seed(99)
    idx = torch.randint(0, P, (check_points,), device=obs[0].device, generator=gen)
    sub = [o[idx].cpu().numpy() for o in obs]
    same = bool(np.array_equal(out[idx].cpu().numpy(), oracle_eval(spec, sub)))
    row = {"config": name, "points": P, "grid": spec["dims"], "kernel": it.kernel_name(),
           "table_MiB": round(tbytes / 2**20, 2), "layout_steps": [si, sj],
           "launches": len(ms), "kernel_ms": round(kernel_ms, 4), "kernel_ms_min": round(float(np.min(ms)), 4),
           "Mpoints_per_s": round(P / kernel_ms / 1e3, 1), "algorithmic_bytes_per_point": bpp,
           "achieved_GBps": round(achieved, 1), "frac": round(achieved / HBM_PEAK_GBPS, 4),
           "oracle_check": {"points": check_points, "bitwise_equal": same}}
    if it.get_option("last_binned"):
        # 4-D multicubic: the batch is counting-sorted by table position first (3 more launches on
        # the same stream); kernel_ms is the whole evaluation.  Its stages from HIP events recorded
        # between the launches (option stage_timing), and the same handle with the points evaluated
        # in place, for comparison:
        it.set_option("stage_timing", 1)
        stages = []
        for _ in range(24):
            it.eval_tensors(obs, out)
            it.finish()
            stages.append(it.stage_ms())
        it.set_option("stage_timing", 0)
        st = {k: round(float(np.median([s[k] for s in stages])), 4) for k in ("hist", "scan", "scatter", "kernel")}
        it.set_option("binned", 0)
        ms0 = time_launches(torch, it, obs, out, seconds=seconds / 2)
        row["binned"] = {"launches_per_evaluation": 4, "stage_ms": st, "unsorted_kernel_ms": round(float(np.mean(ms0)), 4),
                         "stage_note": "histogram (+ counter reset) | scan | scatter of the points into bin order | evaluation "
                                       "kernel on the sorted points; HIP events on the launch stream, median of 24 evaluations"}
        if spec["method"] == "cubic" and spec["ndims"] == 4:
            row["bound"] = cubic4_bound(row, P, st["kernel"], float(np.mean(ms0)), spec)
    row["traffic"] = committed_config_traffic(row, spec)
    it.close()
    return row


# FP64 vector issue peak measured on this part (tools/fp64_rate.hip -> profiles/r03_fp64_vector_rate.txt; the
# microarchitecture guide gives none): v_add/v_mul/v_fma_f64 all issue one wave64 instruction per ~4.1 cycles per SIMD.
FP64_WAVE_INSTR_PER_US_PER_SIMD = 566.0
FP64_FMA_PEAK_TFLOPS = 74.1  # = 566e6 x 1024 SIMDs x 64 lanes x 2 flop
L2_HIT_LINES_PER_S = 2.7e11  # 128 channels x 1 line per clock
L2_MISS_LINES_PER_S = 5.5e10  # random 128-B lines from the Infinity Cache / HBM (tools/tune_sector)


def coherent_batch_row(torch, it, out, dev, P, bpp):
    """The headline handle on a batch that is coherent as it stands — re-gridding onto a 464^3 lattice, last dimension
    fastest — through the automatic path: the device-side sample in front of the launch sends it to the one-pass kernel
    (DESIGN.md section 4.4).  Kernel ms of the automatic launch (sample + the kernel that ran + the one that returned at
    once), of the sweep kernel forced onto the same batch, and whether both leave the same bits."""
    m = 464
    ax = torch.linspace(-1.0, 1.0, m, dtype=torch.float64, device=dev)
    lat = []
    for t in torch.meshgrid(ax, ax, ax, indexing="ij"):
        f = t.reshape(-1)
        lat.append(torch.cat([f, f[:P - m ** 3]]).contiguous())
        del f
    try:
        it.set_option("sweep", 1)
        forced = float(np.mean(time_launches(torch, it, lat, out, seconds=0.1)))
        keep = out.clone()
        it.set_option("sweep", -1)
        it.set_option("sweep_probe", 1)  # (a sample in front of every launch: this handle's history is of unordered batches)
        auto = float(np.mean(time_launches(torch, it, lat, out, seconds=0.15)))
        took = it.get_option("sweep_probe_took_brick")
        same = bool(torch.equal(out, keep))
        del keep
    finally:
        it.set_option("sweep", -1)
        it.set_option("sweep_probe", 2)
    return {"batch": f"{m}^3 lattice over the grid's extent, last dimension fastest, {P:.0e} points",
            "automatic_ms": round(auto, 4), "sweep_kernel_forced_ms": round(forced, 4),
            "device_sample_chose_one_pass_kernel": took == 1, "bitwise_equal_to_sweep_kernel": same,
            "automatic_frac": round(P * bpp / (auto * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}


def configs_summary(rows):
    """Per configuration: kernel ms, roofline fraction, bitwise check against the oracle, fabric traffic over algorithmic."""
    out = []
    for r in rows:
        if "error" in r:
            out.append({"error": r["error"]})
            continue
        name = str(r.get("config", r.get("name", "")))
        e = {"cfg": name.split(" ")[0] + (" lin" if "linearize_extrapolation=true" in name else ""),
             "kernel_ms": r.get("kernel_ms"), "frac": r.get("frac"),
             "bitwise_equal": (r.get("oracle_check") or {}).get("bitwise_equal")}
        t = r.get("traffic") or {}
        if t.get("ratio_to_algorithmic") is not None:
            e["traffic_ratio"] = t.get("ratio_to_algorithmic")
        if t.get("fabric_requests_per_point") is not None:
            e["fabric_requests_per_point"] = t.get("fabric_requests_per_point")
        st = (r.get("binned") or {}).get("stage_ms")
        if st:
            e["stage_ms"] = st
        out.append(e)
    return out


def cubic4_bound(row, P, kernel_stage_ms, unsorted_ms, spec):
    """What bounds cfg4 ('cfg4 is not HBM-bound by construction ... report additionally
    L2/MALL gather GB/s and FP64-VALU utilization so the number is interpretable')."""
    gather_bytes = 8 * 4**4  # a 4^4 footprint of f64 = 16 table lines of 128 B
    nodes = (4**4 - 1) // 3  # 85 one-dimensional reductions per point (multicubic/regular.rs:368-421)
    node_instr = 9 + 2 + 3   # interior node: 9 add/sub, 2 mul by 0.5, 3 fma (multicubic/regular.rs:495-505, mod.rs:72-91)
    node_flop = 9 + 2 + 2 * 3
    t = kernel_stage_ms * 1e-3
    issue_peak = FP64_WAVE_INSTR_PER_US_PER_SIMD * 1e6 * 1024  # wave-instructions per second, whole chip
    column = row["kernel"].startswith("interpn::k_cubic_column")
    return {
        "hbm": "not the bound: 40 B of streams against 2 KiB of table values and ~1.2k f64 operations per point",
        "gather_bytes_per_point": gather_bytes,
        "gather_served_from": "LDS: the (k, l) column of the bin's (i, j) cell, filled once per workgroup (cubic_column.h)" if column
                              else "L2 (sorted points share table lines; tiled kernel)",
        "gather_GBps": round(P * gather_bytes / t / 1e9, 1),
        "unsorted_gather_GBps": round(P * gather_bytes / (unsorted_ms * 1e-3) / 1e9, 1),
        "l2_line_floor_ms": round(P * 16 / L2_HIT_LINES_PER_S * 1e3, 4),
        "l2_miss_line_floor_ms": round(P * 16 / L2_MISS_LINES_PER_S * 1e3, 4),
        "lds_floor_ms": round(P * gather_bytes / (256 * 256 * 2.35e9) * 1e3, 4),
        # round 4: the column kernel evaluates the 64 dim-0 nodes of a point from per-part Hermite coefficients
        # (3 fused steps each; the coefficients are computed once per part and tile line: cubic_column.h), so it
        # issues fewer instructions than the reference's per-point arithmetic, which the next lines still count
        "valu_f64_instr_per_point_as_evaluated": (64 * 3 + 21 * node_instr) if column else nodes * node_instr,
        # what the column kernel is bound by instead (profiles/REJECTED.md, ablation row; r04_traffic.json): fabric
        # requests — per point one 128-B line read for its 32-B record (gathered in the local sort's order), one
        # partial write for its result, 0.11 line of column fill: 2.04 measured
        # (TCC_EA0_RDREQ 1.00e7 + TCC_EA0_WRREQ 1.03e7 per 1e7 points, profiles/r04_traffic.json)
        "fabric_requests_per_point": 2.04 if column else None,
        "fabric_request_floor_ms": round(P * 2.04 / L2_MISS_LINES_PER_S * 1e3, 4) if column else None,
        "valu_f64_instr_per_point": nodes * node_instr,
        "valu_flop_per_point": nodes * node_flop,
        "valu_floor_ms": round(P / 64 * nodes * node_instr / issue_peak * 1e3, 4),
        "valu_frac_of_fp64_vector_issue_peak": round(P / 64 * nodes * node_instr / t / issue_peak, 4),
        "valu_TFLOPs": round(P * nodes * node_flop / t / 1e12, 2),
        "valu_frac_of_fp64_vector_peak": round(P * nodes * node_flop / t / 1e12 / FP64_FMA_PEAK_TFLOPS, 4),
        "peak_source": "measured: tools/fp64_rate.hip, profiles/r03_fp64_vector_rate.txt (566 wave-instr/us/SIMD for add, mul and fma "
                       "alike = 74.1 TFLOP/s counting fma as 2 flop; the microarchitecture guide gives no FP64 vector peak)",
        "counts": "node arithmetic only (85 nodes x 14 f64 instructions); cell location adds 8 IEEE divisions and ~300 further "
                  "instructions per point; floors: 16 lines per point at the L2 hit rate / miss rate, 2 KiB per point at 256 B/clk/CU of LDS",
    }


def cfg1_row(torch, interpn_amd, device):
    """BASELINE configs[0]: 2-D multilinear::regular, 4x4 f64 grid, 1e3 obs — the reference's own
    CPU-runnable plumbing case (src/multilinear/regular.rs:51-117, N = 2 arm).  Nothing here is
    bandwidth: the row reports call latency of the host entry points (numpy in, numpy out), of the
    device entry point, the single-thread CPU port on the same inputs, and a bitwise check of all
    1e3 points through every one of them."""
    from oracle import pyoracle

    rng = np.random.default_rng(11)
    n, P = 4, 1000
    g = np.linspace(-1.0, 1.0, n)
    dims, starts, steps = [n, n], np.full(2, -1.0), np.full(2, g[1] - g[0])
    vals = rng.uniform(-1.0, 1.0, n * n)
    obs = [rng.uniform(-1.05, 1.05, P) for _ in range(2)]  # ~5 % extrapolated
    want = np.zeros(P)
    pyoracle.linear_regular(dims, starts, steps, vals, obs, want)
    t_cpu = float("inf")
    tmp = np.zeros(P)
    for _ in range(200):
        t0 = time.perf_counter()
        pyoracle.linear_regular(dims, starts, steps, vals, obs, tmp)
        t_cpu = min(t_cpu, time.perf_counter() - t0)

    def best_of(fn, reps=300):
        best = float("inf")
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        return best

    got_oneshot = np.zeros(P)
    one = lambda: interpn_amd.raw.interpn_linear_regular_f64(dims, starts, steps, vals, obs, got_oneshot)
    one()
    t_oneshot = best_of(one)
    it = interpn_amd.Interpolator.regular("linear", dims, starts, steps, vals, False, device, np.float64)
    got_handle = np.zeros(P)
    t_handle = best_of(lambda: it.eval_host(obs, got_handle))
    dev = torch.device("cuda", device)
    obs_dev = [torch.from_numpy(o).to(dev) for o in obs]
    out_dev = torch.empty(P, dtype=torch.float64, device=dev)

    def dev_call():
        it.eval_tensors(obs_dev, out_dev)
        it.finish()

    dev_call()
    t_dev = best_of(dev_call)
    ms = time_launches(torch, it, obs_dev, out_dev, launches=200)
    row = {"config": "cfg1 2D multilinear::regular 4x4, 1e3 obs (BASELINE configs[0]: plumbing case, latency not bandwidth)",
           "points": P, "grid": dims, "kernel": it.kernel_name(),
           "call_us": {"one_shot_host_arrays": round(t_oneshot * 1e6, 1), "resident_handle_host_arrays": round(t_handle * 1e6, 1),
                       "resident_handle_device_tensors_incl_status": round(t_dev * 1e6, 1),
                       "kernel_only": round(float(np.median(ms)) * 1e3, 2)},
           "cpu_port_us": round(t_cpu * 1e6, 2), "cpu_port_Mpoints_per_s": round(P / t_cpu / 1e6, 1),
           "oracle_check": {"points": P, "bitwise_equal": bool(np.array_equal(got_oneshot, want) and np.array_equal(got_handle, want)
                                                                 and np.array_equal(out_dev.cpu().numpy(), want))}}
    it.close()
    return row


def cpu_baseline(spec, obs_dev, sample_points):
    """Time the CPU oracle (a port of the reference's algorithm; the Rust reference cannot be
    built here) single-threaded on a bounded sample of the same workload."""
    from oracle import pyoracle

    dims, starts, steps, vals = spec["dims"], spec["starts"], spec["steps"], spec["vals"]
    cal = 1_000_000
    sub = [o[:cal].cpu().numpy() for o in obs_dev]
    out = np.zeros(cal)
    t0 = time.perf_counter()
    pyoracle.linear_regular(dims, starts, steps, vals, sub, out)
    rate = cal / (time.perf_counter() - t0)
    n = sample_points or int(min(obs_dev[0].numel(), max(cal, rate * 4.0)))  # ~4 s per pass, 3 passes
    sub = [o[:n].cpu().numpy() for o in obs_dev]
    out = np.zeros(n)
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        pyoracle.linear_regular(dims, starts, steps, vals, sub, out)
        best = min(best, time.perf_counter() - t0)
    rec = {
        "value": round(n / best / 1e6, 3),
        "unit": "Mpoints/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n} points of rank 0's batch, best of 3 passes, single thread, "
                  f"-O3 -march=x86-64-v3 -ffp-contract=off, fma flavour",
    }
    # Not reference behaviour (the reference is single-threaded): the same port on every host
    # core, one contiguous slice of the same sample per thread (ctypes releases the GIL).
    try:
        from concurrent.futures import ThreadPoolExecutor

        cores = len(os.sched_getaffinity(0))
        if cores > 1:
            out_mt = np.zeros(n)
            cuts = [n * k // cores for k in range(cores + 1)]

            def work(k):
                lo, hi = cuts[k], cuts[k + 1]
                if hi > lo:
                    pyoracle.linear_regular(dims, starts, steps, vals, [o[lo:hi] for o in sub], out_mt[lo:hi])

            with ThreadPoolExecutor(cores) as ex:
                t0 = time.perf_counter()
                list(ex.map(work, range(cores)))
                dt = time.perf_counter() - t0
            rec["all_cores"] = {"value": round(n / dt / 1e6, 3), "unit": "Mpoints/s", "cores": cores,
                                "note": "not reference behaviour: same port, one slice per host thread",
                                "matches_single_thread": bool(np.array_equal(out_mt, out))}
    except Exception as e:  # the single-thread figure is the baseline; this one is optional
        rec["all_cores"] = {"error": str(e)}
    # Anchor to the reference's published data (inference, not a measurement of the reference):
    # SciPy RegularGridInterpolator on the 3-D 20^3 grid / 1e4 points case times the published
    # 11.2x speed-up of linear-regular over SciPy (reference docs).
    try:
        from scipy.interpolate import RegularGridInterpolator

        ga = np.linspace(-1.0, 1.0, 20)
        rng = np.random.default_rng(7)
        gv = rng.uniform(-1, 1, (20, 20, 20))
        pts = rng.uniform(-1, 1, (10_000, 3))
        rgi = RegularGridInterpolator((ga, ga, ga), gv, method="linear", bounds_error=False, fill_value=None)
        rgi(pts)
        tb = float("inf")
        for _ in range(5):
            t0 = time.perf_counter()
            rgi(pts)
            tb = min(tb, time.perf_counter() - t0)
        rec["scipy_anchor"] = {"scipy_Mpoints_per_s": round(1e4 / tb / 1e6, 3),
                               "inferred_reference_Mpoints_per_s": round(11.2 * 1e4 / tb / 1e6, 3),
                               "note": "inference: SciPy 20^3/1e4 points here x published 11.2x; not a run of the reference"}
    except Exception as e:
        rec["scipy_anchor"] = {"error": str(e)}
    return rec, out, n


def _traffic_profile():
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            return json.load(f)
    except Exception:
        return None


def committed_traffic(kernel, points, grid, table_bytes):
    """HBM/fabric bytes per launch of the headline kernel from the committed rocprofv3 --pmc passes
    (separate runs, as the guide prescribes) — reported only if that profile was taken on the SAME
    kernel instantiation, table layout, grid and batch as this run; otherwise null.  Never measured
    inside this run: PMC collection needs its own rocprofv3 passes."""
    tj = _traffic_profile()
    if tj is None:
        return None, "no committed traffic profile"
    same = (tj.get("points") == points and tj.get("grid") == grid and tj.get("kernel") == kernel and
            tj.get("table_bytes") == table_bytes)
    if not same:
        return None, (f"committed profile is for kernel={tj.get('kernel')!r} grid={tj.get('grid')} "
                      f"table_bytes={tj.get('table_bytes')}: does not match this run, not reported")
    return tj.get("hbm_bytes_per_launch"), tj.get("source", "profiles/traffic_latest.json")


def live_traffic(kernel, n, points, timeout_s=150):
    """Fabric-side bytes per launch of the headline kernel measured NOW: two child runs of the same workload under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: one counter per pass, kernel trace only, as MI355X_MICROARCH.md's HBM
    section prescribes; FETCH_SIZE x 2 on gfx950 — the factor profiles/r06_calibration_pmc_*.csv re-measured at 1.99998 on
    a stream kernel of known byte count).  After the timed region; any failure leaves the committed figure in place."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if any(k.startswith("ROCP") or k.startswith("ROCPROF") for k in os.environ):
        return None, "this process itself runs under a profiler: no nested PMC passes"
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    short = kernel.split("<")[0].split("::")[-1]
    got = {}
    tmp = tempfile.mkdtemp(prefix="interpn_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            env = dict(os.environ, TMPDIR="/tmp", INTERPN_BENCH_ROOT=ROOT)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.join(ROOT, "tools", "traffic_child.py"), str(n), str(points), "12"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == counter and short + "<" in row.get("Kernel_Name", ""):
                        vals.append(float(row["Counter_Value"]))
            if r.returncode != 0 or len(vals) < 4:
                return None, f"PMC pass {counter} failed (rc {r.returncode}, {len(vals)} rows): {r.stderr[-200:]!r}"
            got[counter] = (float(np.mean(vals[2:])), len(vals) - 2)  # (the first launches: period calibration, the sample)
    except Exception as e:  # timeout, missing tool, unreadable output ...
        return None, "PMC passes failed: " + repr(e)[:200]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    rd = got["FETCH_SIZE"][0] * 1024.0 * 2.0
    wr = got["WRITE_SIZE"][0] * 1024.0
    return {"bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "launches_averaged": got["FETCH_SIZE"][1],
            "corrections": {"FETCH_SIZE": 2.0, "WRITE_SIZE": 1.0, "unit": "KiB"}}, \
        "measured in this run: child processes of the same workload under rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes)"


def committed_config_traffic(row, spec):
    """The same for a row of `configs[]`: fabric-side bytes per evaluation (all of its kernels) and
    their ratio to the algorithmic bytes, the TCC hit / miss counts of the evaluation kernel — from the
    committed per-configuration --pmc passes (tools/profile_r04.sh), if kernel, table and batch match."""
    tj = _traffic_profile()
    if not tj or "configs" not in tj:
        return {"fabric_bytes_per_evaluation": None, "measured_in_this_run": False, "source": "no committed per-configuration profile"}
    n = spec["dims"][0]
    for key, e in tj["configs"].items():
        if e.get("kernel") == row["kernel"] and e.get("points") == row["points"] and e.get("grid") == n and \
                e.get("ndims") == spec["ndims"] and abs(e.get("table_bytes", 0) - row["table_MiB"] * 2**20) < 2**16:
            main = {}
            for k, d in e.get("kernels", {}).items():
                if row["kernel"].split("::")[-1].split("<")[0] == k:
                    main = {c: d[c] for c in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum",
                                              "TCC_EA0_WRREQ_64B_sum") if c in d}
            # fabric REQUESTS (128-byte reads, 64-/32-byte writes) of all the evaluation's kernels, and the rate this run's time
            # makes of them (the one-pass brick kernels on unordered points sit at 5.2-5.8e10 requests/s, the fabric's
            # random-line rate; the sweep kernel makes fewer requests — DESIGN.md section 4; cfg4, whose writes are partial lines: 4e10)
            reqs = sum(d.get("TCC_EA0_RDREQ_sum", 0.0) + d.get("TCC_EA0_WRREQ_sum", 0.0) for d in e.get("kernels", {}).values())
            return {"fabric_requests_per_evaluation": round(reqs) if reqs else None,
                    "fabric_requests_per_point": round(reqs / row["points"], 3) if reqs else None,
                    "fabric_requests_per_s_at_this_runs_time": round(reqs / (row["kernel_ms"] * 1e-3), -8) if reqs and row.get("kernel_ms") else None,
                    "fabric_bytes_per_evaluation": e["fabric_bytes_per_evaluation"],
                    "fabric_read_bytes": e["fabric_read_bytes_per_evaluation"], "fabric_write_bytes": e["fabric_write_bytes_per_evaluation"],
                    "algorithmic_bytes": e["algorithmic_bytes"], "ratio_to_algorithmic": e["ratio_to_algorithmic"],
                    "evaluation_kernel_tcc": main, "measured_in_this_run": False,
                    "source": f"profiles/traffic_latest.json [{key}]: separate rocprofv3 --pmc passes (FETCH_SIZE x {tj.get('fetch_correction', 2.0):.3f}, "
                              f"WRITE_SIZE x {tj.get('write_correction', 1.0):.3f}), same kernel instantiation, table and batch as this row"}
    return {"fabric_bytes_per_evaluation": None, "measured_in_this_run": False,
            "source": "committed per-configuration profile has no entry for this kernel / table / batch"}


def host_path_block(interpn_amd, device, points=10_000_000):
    """End to end on HOST arrays (numpy in, numpy out: upload, kernel, download) for cfg2's shapes at
    1e7 points — SURVEY.md section 8(d) "GPU timing" asks for it as a separate line; PCIe-bound by
    construction and never the `value`."""
    rng = np.random.default_rng(11)
    n = 64
    spec = build_spec("linear", "regular", n, 3, False, 1)
    obs = [rng.uniform(-1.0, 1.0, points) for _ in range(3)]
    out = np.zeros(points)
    it = make_interp(interpn_amd, spec, device)
    it.eval_host(obs, out)  # warm-up: lanes allocated, pages touched
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        it.eval_host(obs, out)
        best = min(best, time.perf_counter() - t0)
    it.close()
    t0 = time.perf_counter()
    interpn_amd.raw.interpn_linear_regular_f64(spec["dims"], spec["starts"], spec["steps"], spec["vals"], obs, out)
    oneshot = time.perf_counter() - t0
    return {"workload": f"3D multilinear::regular {n}^3 f64, {points:.0e} points, numpy arrays in pageable host memory -> numpy array",
            "resident_handle_Gpoints_per_s": round(points / best / 1e9, 3), "resident_handle_GBps_over_pcie": round(points * 32 / best / 1e9, 1),
            "one_shot_raw_call_Gpoints_per_s": round(points / oneshot / 1e9, 3),
            "note": "upload of 24 B/point + download of 8 B/point over one PCIe link, chunked double-lane pipeline "
                    "(abi_host.hip); bound by the link, not by the kernel"}


def run_ablation(torch, spec, obs, out, it, seconds):
    """Stream-only and gather-only variants of the product kernel (same source, a template flag;
    tools/ablate_linear3d.hip), on the same table layout the handle chose.  Outputs of these
    variants are meaningless by construction; only their durations are reported."""
    import ctypes

    path = os.path.join(ROOT, "tools", "libinterpn_ablate.so")
    if not os.path.exists(path):
        return {"error": "tools/libinterpn_ablate.so not built (python -c 'import __graft_entry__ as g; g.build()')"}
    lib = ctypes.CDLL(path)
    lib.ablate_create.restype = ctypes.c_void_p
    lib.ablate_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.ablate_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.ablate_destroy.argtypes = [ctypes.c_void_p]
    _, si, sj = it.table_layout()
    if not si:
        return {"error": "the handle reads the C-ordered grid: no brick kernel to ablate"}
    dev = obs[0].device
    vals_dev = torch.from_numpy(spec["vals"]).to(dev)
    h = lib.ablate_create(ctypes.c_void_p(vals_dev.data_ptr()), spec["n"], si, sj, float(spec["steps"][0]))
    if not h:
        return {"error": "ablate_create failed"}
    stream = torch.cuda.current_stream(dev).cuda_stream
    P = obs[0].numel()
    res = {}
    try:
        for mode, key in ((0, "full_ms"), (1, "stream_only_ms"), (2, "gather_only_ms")):
            ms = []
            t_end = time.perf_counter() + seconds
            k = 0
            while time.perf_counter() < t_end or k < 8:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = lib.ablate_launch(h, mode, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(),
                                       P, ctypes.c_void_p(stream))
                b.record()
                if rc != 0:
                    return {"error": f"ablate_launch mode {mode} failed with HIP error {rc}"}
                torch.cuda.synchronize()
                if k >= 2:
                    ms.append(a.elapsed_time(b))
                k += 1
            res[key] = round(float(np.mean(ms)), 4)
    finally:
        lib.ablate_destroy(h)
    res["note"] = ("same kernel source as the product path (interpn_amd/csrc/linear_brick.h, template flag ABL): "
                   "stream_only = coordinates read + result written, cell values synthesised (no table access); "
                   "gather_only = table gathers + arithmetic, coordinates synthesised, nothing stored; "
                   "full = the unmodified kernel through the harness (cross-check of kernel_ms)")
    return res


def run_sweep_ablation(torch, spec, obs, out, kernel_name, seconds):
    """Measurement builds of the SWEEP kernel's own source (interpn_amd/csrc/linear_sweep.h, template flag ABL;
    tools/ablate_linear3d.hip) in the product's shape — full / no table access / no streams — and the cost of
    the line visits alone: as many visits of random 128-byte lines as the batch has points, made the way the
    kernel makes them (quad-cooperative, 16 lines per wave instruction), from an L2-resident table and from one of
    this run's table size.  The last two are the floor under any kernel that reads one line per point
    (DESIGN.md section 4.1).  Outputs of the measurement builds are meaningless by construction."""
    import ctypes

    path = os.path.join(ROOT, "tools", "libinterpn_ablate.so")
    if not os.path.exists(path):
        return {"error": "tools/libinterpn_ablate.so not built"}
    if "k_linear_sweep<double, false, true, 1, 1, 12, 768, 0, false, 0, 4" not in kernel_name:
        return {"error": f"the product kernel of this run is {kernel_name!r}: no measurement build of that shape"}
    lib = ctypes.CDLL(path)
    lib.ablate_create.restype = ctypes.c_void_p
    lib.ablate_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.ablate_launch_sweep.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.ablate_set_sweep_parked.argtypes = [ctypes.c_int]
    lib.ablate_set_sweep_abl.argtypes = [ctypes.c_int]
    lib.ablate_set_sweep_clock.argtypes = [ctypes.c_uint]
    lib.ablate_line_visits.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    lib.ablate_destroy.argtypes = [ctypes.c_void_p]
    dev = obs[0].device
    vals_dev = torch.from_numpy(spec["vals"]).to(dev)
    h = lib.ablate_create(ctypes.c_void_p(vals_dev.data_ptr()), spec["n"], 1, 1, float(spec["steps"][0]))
    if not h:
        return {"error": "ablate_create failed"}
    stream = torch.cuda.current_stream(dev).cuda_stream
    P = obs[0].numel()
    P16 = P // 16 * 16  # (the harness takes 16-byte aligned streams; the last few points do not matter here)
    res = {}

    def timed(fn):
        ms, k = [], 0
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end or k < 8:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn()
            b.record()
            if rc != 0:
                raise RuntimeError(f"HIP error {rc}")
            torch.cuda.synchronize()
            if k >= 3:  # (the sweep kernel measures its period on its first launches)
                ms.append(a.elapsed_time(b))
            k += 1
        return round(float(np.mean(ms)), 4)

    try:
        lib.ablate_set_sweep_parked(4)
        lib.ablate_set_sweep_clock(0)
        for abl, key in ((0, "full_ms"), (1, "no_table_access_ms"), (2, "no_streams_ms")):
            lib.ablate_set_sweep_abl(abl)
            res[key] = timed(lambda: lib.ablate_launch_sweep(h, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), P16, 12, 768, 0,
                                                             ctypes.c_void_p(stream)))
        lib.ablate_set_sweep_abl(0)
        table = torch.rand(16 * 2**20 // 8, dtype=torch.float64, device=dev)
        sink = torch.zeros(8, dtype=torch.float64, device=dev)
        for nbytes, key in ((2 * 2**20, "line_visits_L2_resident_table_ms"), (10 * 2**20, "line_visits_10MiB_table_ms")):
            res[key] = timed(lambda: lib.ablate_line_visits(ctypes.c_void_p(table.data_ptr()), nbytes, P16, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(stream)))
    except RuntimeError as e:
        return {"error": str(e)}
    finally:
        lib.ablate_destroy(h)
    res["note"] = ("measurement builds of the product kernel's source in its shape (12 rows in registers + 4 parked, 768 threads): no_table_access = "
                   "streams, sort and arithmetic on made-up cell values; no_streams = sort, gathers and arithmetic on made-up coordinates, nothing stored; "
                   "line_visits = as many quad-cooperative reads of random 128-byte lines as the batch has points and nothing else (tools/tune_sector.hip) — "
                   "the floor of one line per point; the sweep keeps its 10 MiB table L2-resident in time")
    return res


def sharding_text(world, bcast):
    """The record's description of the multi-GPU data path, generated from what this run did."""
    head = ("obs sharded contiguously, one shard per rank" if world > 1 else
            "single rank: the whole batch on one GPU")
    if bcast is None:
        return head + "; no process group, grid uploaded by the rank itself; no collective anywhere"
    lib = "RCCL (nccl backend), device tensors" if bcast["backend"] == "nccl" else \
          f"{bcast['backend']} backend, staged through the host"
    return (f"{head}; grid replicated from rank 0 by one broadcast over {lib}, {bcast['ranks']} rank(s), "
            f"{bcast['bytes']} bytes; no collective in the timed loop")


# ---------------------------------------------------------------------------------------------
def device_identity(torch):
    """(current device index, PCI domain, bus, device) of this rank's GPU; -1 where the runtime does not say."""
    idx = torch.cuda.current_device()
    props = torch.cuda.get_device_properties(idx)
    return [idx, int(getattr(props, "pci_domain_id", -1)), int(getattr(props, "pci_bus_id", -1)),
            int(getattr(props, "pci_device_id", -1))]


def gather_devices(torch, dist, rank, world, coll_dev, ident=None):
    """One all-gather of every rank's (rank, device index, PCI domain, bus, device); `dist` None: a single process."""
    mine = [rank] + (ident if ident is not None else device_identity(torch))
    if dist is None or world == 1:
        rows = [mine]
    else:
        got = [torch.zeros(5, dtype=torch.int64, device=coll_dev) for _ in range(world)]
        dist.all_gather(got, torch.tensor(mine, dtype=torch.int64, device=coll_dev))
        rows = [[int(v) for v in g.tolist()] for g in got]
    return [{"rank": r[0], "cuda_device": r[1], "pci": "%04x:%02x:%02x" % (r[2], r[3], r[4]) if min(r[2:]) >= 0 else None}
            for r in rows]


def check_distinct_devices(devices, world):
    """A world of N ranks must sit on N distinct GPUs: by PCI address where every rank has one, else by device index."""
    keys = [d["pci"] for d in devices] if all(d["pci"] for d in devices) else [d["cuda_device"] for d in devices]
    if len(set(keys)) != world:
        raise SystemExit(f"bench.py: {world} ranks on {len(set(keys))} distinct device(s) {sorted(set(map(str, keys)))}: "
                         "not a multi-GPU measurement (pass --same-device to share one GPU on purpose)")


def dry_run(args):
    """Spawn / rendezvous check without a GPU: what the launcher contract gives every rank."""
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("INTERPN_BENCH_DRY_FAIL_RANK") == str(rank):
        raise SystemExit(3)
    # no GPU here: the identity a rank would report is made up from LOCAL_RANK (INTERPN_BENCH_DRY_SAME_PCI: every rank
    # claims the same GPU, the case the real run must refuse)
    lr = int(os.environ.get("LOCAL_RANK", "0"))
    ident = [0, 0, 7, 0] if os.environ.get("INTERPN_BENCH_DRY_SAME_PCI") else [lr, 0, 0x10 + lr, 0]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        world = dist.get_world_size()
        got = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, torch.tensor([rank, int(os.environ.get("LOCAL_RANK", "-1"))], dtype=torch.int64))
        devices = gather_devices(torch, dist, rank, world, torch.device("cpu"), ident=ident)
        dist.barrier()
        ranks = [[int(x[0]), int(x[1])] for x in got]
        dist.destroy_process_group()
    else:
        ranks = [[0, lr]]
        devices = gather_devices(torch, None, rank, world, torch.device("cpu"), ident=ident)
    if rank == 0:
        if not args.same_device:
            check_distinct_devices(devices, world)
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": ranks, "steps": args.steps,
                          "config": {"devices": devices},
                          "spawned": bool(os.environ.get("INTERPN_BENCH_SPAWNED"))}), flush=True)


def worker(args):
    if args.dry_run:
        return dry_run(args)
    import torch
    import torch.distributed as dist

    import interpn_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: interpn_amd has no CPU path")
    if args.same_device:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        world = dist.get_world_size()  # what the process group actually is
    coll_dev = dev if (use_dist and dist.get_backend() == "nccl") else torch.device("cpu")
    # Which physical device did every rank land on?  (rank, torch's current device, PCI domain / bus / device as the
    # HIP runtime reports them for that device) from every rank; rank 0 refuses to report a multi-GPU figure measured on
    # fewer distinct devices than ranks (unless --same-device asked for exactly that).
    devices = gather_devices(torch, dist if use_dist else None, rank, world, coll_dev)
    if rank == 0 and not args.same_device:
        check_distinct_devices(devices, world)

    P = args.points
    NDIMS = 3
    bpp = 8 * (NDIMS + 1)  # read 3 f64 coordinates + write 1 f64 result

    def barrier():
        if use_dist:
            dist.barrier()

    # Observation shard of this rank: i.i.d. uniform over the grid extent, device resident.
    gen = torch.Generator(device=dev)
    gen.manual_seed(3 + 1000 * rank)
    obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(NDIMS)]
    out = torch.empty(P, dtype=torch.float64, device=dev)

    def measure(n):
        """One workload (n^3 grid, this rank's P points) by the bench contract: grid generated on
        rank 0 and replicated by ONE broadcast, W warm-up steps, EXACTLY K timed steps between
        barrier + synchronize pairs, max over ranks.  Returns (record pieces, handle, spec)."""
        spec = build_spec("linear", "regular", n, NDIMS, False, seed=1)
        vals_dev = torch.empty(n**NDIMS, dtype=torch.float64, device=dev)
        if rank == 0:
            vals_dev.copy_(torch.from_numpy(spec["vals"]))
        bcast = None  # what really replicated the grid (the record reports this, not an intention)
        if use_dist:
            torch.cuda.synchronize()
            tb0 = time.perf_counter()
            if dist.get_backend() == "nccl":
                dist.broadcast(vals_dev, src=0)  # RCCL over xGMI, device to device
                torch.cuda.synchronize()
                where = "device"
            else:
                stage = vals_dev.cpu()
                dist.broadcast(stage, src=0)
                vals_dev.copy_(stage)
                where = "host"
            bcast = {"collective": "broadcast", "backend": dist.get_backend(), "on": where, "src": 0,
                     "ranks": dist.get_world_size(), "bytes": int(vals_dev.numel() * 8),
                     "ms": round((time.perf_counter() - tb0) * 1e3, 3)}
        it = make_interp(interpn_amd, spec, local_rank, vals=vals_dev)
        # A first barrier BEFORE the untimed launches: the first collective of a process group pays
        # RCCL's one-off set-up (tens of ms with the GPU idle, after which the clocks have dropped:
        # profiles/r03_nccl_barrier_clock_ramp.txt shows the 30 launches behind such a barrier
        # running 1.37 -> 1.25 ms); the barrier that brackets the timed region is then a warm one
        # that every rank reaches within a step of the others.
        barrier()
        # Clock spin-up (untimed, every rank): the first ~25 launches after an idle period run 2-6 %
        # slow while the GPU's clocks ramp (rocprofv3 kernel trace: 1.305, 1.292, 1.274 ... 1.226 ms),
        # longer than the driver's W warm-up steps cover; the W steps and the K timed steps follow.
        if args.spinup_seconds > 0:
            time_launches(torch, it, obs, out, seconds=args.spinup_seconds)
        for _ in range(args.warmup):
            it.eval_tensors(obs, out)
            it.finish()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        torch.cuda.synchronize()
        tb0 = time.perf_counter()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        barrier_ms = (t0 - tb0) * 1e3  # GPU idle time in front of the first timed launch
        for k in range(args.steps):
            ev[k][0].record()
            it.eval_tensors(obs, out)
            ev[k][1].record()
            it.finish()
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        step_ms = [a.elapsed_time(b) for a, b in ev]
        kernel_ms_local = float(np.mean(step_ms))
        # Same workload on ONE GPU with the others idle (rank 0 alone), so that the N-rank figure can
        # be read against an identical single-GPU one taken in the same run.  AFTER the timed region:
        # the ranks that idle at this barrier come back with cold clocks.
        solo_ms = None
        if world > 1:
            if rank == 0:
                solo_ms = float(np.mean(time_launches(torch, it, obs, out, launches=max(20, min(args.steps, 200)),
                                                      finish_each=True)))
            barrier()
        t = torch.tensor([elapsed, kernel_ms_local], dtype=torch.float64, device=coll_dev)
        per_rank_ms = [kernel_ms_local]
        if use_dist:
            gathered = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
            dist.all_gather(gathered, torch.tensor([kernel_ms_local], dtype=torch.float64, device=coll_dev))
            per_rank_ms = [float(x[0]) for x in gathered]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return {"elapsed": float(t[0]), "kernel_ms": float(t[1]), "per_rank_ms": per_rank_ms, "solo_ms": solo_ms,
                "vals_dev": vals_dev, "broadcast": bcast,
                "step_ms": step_ms, "barrier_ms": barrier_ms}, it, spec

    # The headline workload keeps ONE grid for every N (north_star: "throughput on synthetic random
    # obs over a fixed grid reported at 1/2/4/8 GPUs"): BASELINE configs[1]'s 64^3 grid, 1e8 obs per
    # rank (weak scaling), so value(N) / value(1) is a scaling figure on identical per-GPU work.
    # BASELINE configs[4] (128^3 grid, 1e8 obs per rank = 8e8 over 8 GPUs) is measured by the same
    # protocol as a second block ("cfg5") whenever N > 1; at N = 1 its shard is a row of configs[].
    workload = args.workload if args.workload != "auto" else "cfg2"
    n = args.grid or WORKLOADS[workload][0]
    m, it, spec = measure(n)
    elapsed, kernel_ms, per_rank_ms, solo_ms = m["elapsed"], m["kernel_ms"], m["per_rank_ms"], m["solo_ms"]
    second = None
    if world > 1 and workload == "cfg2" and not args.grid and not args.no_cfg5:
        m5, it5, _ = measure(WORKLOADS["cfg5"][0])
        if rank == 0:
            tb5, si5, sj5 = it5.table_layout()
            v5 = P * world * args.steps / m5["elapsed"] / 1e6
            a5 = P * bpp / (m5["kernel_ms"] * 1e-3) / 1e9
            second = {
                "workload": f"3D multilinear::regular, 128^3 f64 grid, {P:.0e} random obs per GPU = {P * world:.0e} obs over "
                            f"{world} GPUs (BASELINE configs[4]); " + sharding_text(world, m5["broadcast"]),
                "value": round(v5, 1), "unit": "Mpoints/s", "value_per_gpu": round(v5 / world, 1),
                "ms_per_step": round(m5["elapsed"] / args.steps * 1e3, 4), "kernel": it5.kernel_name(),
                "table_MiB": round(tb5 / 2**20, 2), "layout_steps": [si5, sj5],
                "kernel_ms": round(m5["kernel_ms"], 4), "kernel_ms_per_rank": [round(x, 4) for x in m5["per_rank_ms"]],
                "achieved_GBps_per_gpu": round(a5, 1), "frac": round(a5 / HBM_PEAK_GBPS, 4),
                "single_gpu_same_workload": {"kernel_ms": round(m5["solo_ms"], 4),
                                             "Mpoints_per_s": round(P / m5["solo_ms"] / 1e3, 1)} if m5["solo_ms"] else None,
            }
        it5.close()
        del m5

    if rank == 0:
        kernel = it.kernel_name()
        tbytes, si, sj = it.table_layout()
        total_points = P * world * args.steps
        value = total_points / elapsed / 1e6
        achieved = P * bpp / (kernel_ms * 1e-3) / 1e9

        # Measured device-copy bandwidth (1 read + 1 write of 0.8 GB), reported beside the nominal peak.
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        scratch = torch.empty_like(out)
        scratch.copy_(obs[0])
        best = float("inf")
        for _ in range(5):
            c0.record()
            scratch.copy_(obs[0])
            c1.record()
            torch.cuda.synchronize()
            best = min(best, c0.elapsed_time(c1))
        copy_gbps = 2 * 8 * P / (best * 1e-3) / 1e9
        del scratch

        traffic, traffic_src = committed_traffic(kernel, P, n, tbytes)
        rec = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"3D multilinear::regular, {n}^3 f64 grid, {P:.0e} random obs per GPU "
                            f"({WORKLOADS[workload][1]}" + (f", {P * world:.0e} obs over {world} GPUs)" if world > 1 else ")"),
                "points_per_gpu": P,
                "grid": [n] * NDIMS,
                "spinup_seconds": args.spinup_seconds,
                "sharding": sharding_text(world, m["broadcast"]),
                "value_per_gpu": round(value / world, 1),
                "backend": dist.get_backend() if use_dist else None,
                "process_group": {"initialised": True, "backend": dist.get_backend(), "world_size": dist.get_world_size()}
                                 if use_dist else {"initialised": False},
                "grid_broadcast": m["broadcast"],
                "devices": devices,
                "barrier_ms_before_timed_region": round(m["barrier_ms"], 3),
                "launched_by": "bench.py spawn" if os.environ.get("INTERPN_BENCH_SPAWNED") else
                               ("external launcher" if world > 1 else "single process"),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_measured_in_this_run": False,
                "kernel": kernel,
                "table_bytes": tbytes,
                "table_MiB": round(tbytes / 2**20, 2),
                "layout_steps": [si, sj],
                "kernel_ms": round(kernel_ms, 4),
                # how the automatic path chose its kernel for this batch: a device-side sample in front
                # of the sweep launch sends coherent batches to the one-pass kernel; the host thins the samples out once they
                # keep coming out unordered (policy 2), so most timed steps are ONE kernel, some are sample + sweep + gated launch
                "auto_path_sample": {"policy": it.get_option("sweep_probe"),
                                     "unordered_verdicts_in_a_row": it.get_option("sweep_probe_streak"),
                                     "note": "policy 2: a device-side sample in front of every automatic launch until three in a row "
                                             "said 'unordered', then in front of every 16th; coherent batches go to the one-pass kernel"},
                "kernel_ms_per_rank": [round(x, 4) for x in per_rank_ms],
                "kernel_ms_min": round(float(np.min(m["step_ms"])), 4),
                "kernel_ms_max": round(float(np.max(m["step_ms"])), 4),
                "kernel_ms_first_steps": [round(x, 3) for x in m["step_ms"][:32]],
                "algorithmic_bytes_per_point": bpp,
                "measured_copy_GBps": round(copy_gbps, 1),
                "frac_of_measured_copy": round(achieved / copy_gbps, 4),
            },
        }
        if second is not None:
            rec["cfg5"] = second
        if world > 1 and solo_ms:
            rec["config"]["single_gpu_same_workload"] = {
                "kernel_ms": round(solo_ms, 4), "Mpoints_per_s": round(P / solo_ms / 1e3, 1),
                "note": "rank 0 alone on its GPU, the other ranks idle at a barrier, same grid and batch: "
                        "value / (n_gpus x this) is the scaling efficiency on an identical workload"}
        if world == 1:
            # >= 0.5 s of back-to-back launches (the K-step region above may be only 20 launches)
            ms = time_launches(torch, it, obs, out, seconds=args.sustain_seconds)
            sk = float(np.mean(ms))
            rec["roofline"]["sustained"] = {
                "launches": len(ms), "seconds": round(float(np.sum(ms)) / 1e3, 3), "kernel_ms": round(sk, 4),
                "kernel_ms_min": round(float(np.min(ms)), 4), "kernel_ms_max": round(float(np.max(ms)), 4),
                "achieved": round(P * bpp / (sk * 1e-3) / 1e9, 1), "frac": round(P * bpp / (sk * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
            # the same handle with the sweep kernel switched off: the brick kernel on the handle's own table
            # (what every round before round 5 measured), bit-compared over the whole batch
            if kernel.startswith("interpn::k_linear_sweep<"):
                try:
                    keep = out.clone()
                    it.set_option("sweep", 0)
                    msb = time_launches(torch, it, obs, out, seconds=min(0.25, args.sustain_seconds))
                    rec["roofline"]["brick_kernel"] = {
                        "kernel": it.kernel_name(), "kernel_ms": round(float(np.mean(msb)), 4), "launches": len(msb),
                        "frac": round(P * bpp / (float(np.mean(msb)) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                        "table_MiB": round(it.table_layout()[0] / 2**20, 2), "bitwise_equal_to_sweep": bool(torch.equal(out, keep)),
                        "note": "same handle, option sweep = 0: one pass of 512-point workgroups over unordered points"}
                    del keep
                finally:
                    it.set_option("sweep", -1)
            if not args.no_ablate:
                try:
                    rec["roofline"]["ablation"] = run_ablation(torch, spec, obs, out, it, 0.25)
                    rec["roofline"]["sweep_ablation"] = run_sweep_ablation(torch, spec, obs, out, rec["roofline"].get("kernel", ""), 0.2)
                    so = rec["roofline"]["ablation"].get("stream_only_ms")
                    if so:
                        # the part's achievable rate for this 3-read / 1-write stream pattern (no table access)
                        rec["roofline"]["stream_only_GBps"] = round(P * bpp / (so * 1e-3) / 1e9, 1)
                        rec["roofline"]["frac_of_stream_only"] = round(so / sk, 4)
                except Exception as e:  # a measurement aid must never take the record down
                    rec["roofline"]["ablation"] = {"error": repr(e)}
            # restore `out` (the ablation kernels scribble on it) before it is compared below
            it.eval_tensors(obs, out)
            it.finish()
            if not args.no_cpu_baseline:
                cb, cpu_out, ncpu = cpu_baseline(spec, obs, args.cpu_sample)
                # the oracle here is the checker, never the thing measured on the GPU side
                cb["gpu_matches_bitwise"] = bool(np.array_equal(out[:ncpu].cpu().numpy(), cpu_out))
                rec["cpu_baseline"] = cb
            if not args.no_configs and workload == "cfg2" and not args.grid and P == 100_000_000:
                try:
                    rec["roofline"]["coherent_batch"] = coherent_batch_row(torch, it, out, dev, P, bpp)
                except Exception as e:  # a side measurement must never take the record down
                    rec["roofline"]["coherent_batch"] = {"error": repr(e)}
            if not args.no_live_traffic and workload == "cfg2" and not args.grid and P == 100_000_000:
                lt, lsrc = live_traffic(kernel, n, P)
                rec["roofline"]["traffic_live"] = {"result": lt, "how": lsrc}
                if lt:
                    rec["roofline"]["traffic_committed_profile"] = {"bytes_per_launch": rec["roofline"]["traffic"], "source": rec["roofline"]["traffic_source"]}
                    rec["roofline"]["traffic"] = lt["bytes_per_launch"]
                    rec["roofline"]["traffic_source"] = lsrc
                    rec["roofline"]["traffic_measured_in_this_run"] = True
                    rec["roofline"]["traffic_over_algorithmic"] = round(lt["bytes_per_launch"] / (P * bpp), 3)
            if not args.no_configs and P == 100_000_000:
                rows = []
                secs = args.sustain_seconds
                try:
                    rows.append(cfg1_row(torch, interpn_amd, local_rank))
                    rows.append(config_row(torch, interpn_amd, "cfg2 3D multilinear::regular 64^3, 1e8 obs",
                                           build_spec("linear", "regular", 64, 3, False, 1), obs, out, local_rank, secs))
                    rows.append(config_row(torch, interpn_amd, "cfg3 3D multilinear::rectilinear non-uniform 64^3, 1e8 obs",
                                           build_spec("linear", "rectilinear", 64, 3, False, 2), obs, out, local_rank, secs))
                    rows.append(config_row(torch, interpn_amd, "cfg5-shard 3D multilinear::regular 128^3, 1e8 obs (one of 8 shards)",
                                           build_spec("linear", "regular", 128, 3, False, 1), obs, out, local_rank, secs))
                    P4 = 10_000_000
                    obs4 = [o[:P4] for o in obs] + [torch.rand(P4, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0]
                    for lin in (False, True):
                        rows.append(config_row(torch, interpn_amd,
                                               f"cfg4 4D multicubic::regular 32^4, 1e7 obs, linearize_extrapolation={str(lin).lower()}",
                                               build_spec("cubic", "regular", 32, 4, lin, 4), obs4, out[:P4], local_rank, secs))
                except Exception as e:
                    rows.append({"error": repr(e)})
                rec["configs"] = rows
                # the same rows in short inside a key every parser of this line keeps
                rec["roofline"]["configs_summary"] = configs_summary(rows)
                try:
                    rec["host_path"] = host_path_block(interpn_amd, local_rank)
                except Exception as e:
                    rec["host_path"] = {"error": repr(e)}
        print(json.dumps(rec), flush=True)

    it.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))
    worker(args)


if __name__ == "__main__":
    main()
