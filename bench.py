#!/usr/bin/env python3
"""Headline benchmark: Mpoints/s (f64) of 3-D multilinear-regular interpolation, 64^3 grid,
1e8 random observation points per GPU (BASELINE.json configs[1]), plus the achieved fraction of
the HBM roofline for the kernel and the CPU oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path over one batch: one `interpn_hip_eval_device` launch over the
rank's 1e8 device-resident points followed by the status check (`interpn_hip_finish`).  Inputs
are synthetic: axes linspace(-1,1,64), vals U(-1,1), obs i.i.d. uniform over
the grid extent, unordered.  With N > 1 the observation batch is sharded (weak scaling: 1e8
points per rank), the grid is broadcast once from rank 0 over RCCL, and there is no collective
in the timed loop.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
NDIMS = 3
GRID_N = 64
BYTES_PER_POINT = 8 * (NDIMS + 1)  # read 3 f64 coordinates + write 1 f64 result


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=100_000_000, help="observation points per GPU")
    ap.add_argument("--grid", type=int, default=GRID_N, help="grid points per axis")
    ap.add_argument("--cpu-sample", type=int, default=0, help="points for the CPU baseline (0 = auto, ~10-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    # Test aids for a 1-GPU box: run the multi-rank control flow with every rank on cuda:0 over
    # gloo (RCCL refuses two ranks on one device).  The driver never passes these.
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-device", action="store_true")
    return ap.parse_args()


def cpu_baseline(dims, starts, steps, vals, obs_dev, sample_points):
    """Time the CPU oracle (a port of the reference's algorithm; the Rust reference cannot be
    built here) single-threaded on a bounded sample of the same workload."""
    from oracle import pyoracle

    # calibrate on 1e6 points, then size the sample for ~12 s unless told otherwise
    cal = 1_000_000
    sub = [o[:cal].cpu().numpy() for o in obs_dev]
    out = np.zeros(cal)
    t0 = time.perf_counter()
    pyoracle.linear_regular(dims, starts, steps, vals, sub, out)
    rate = cal / (time.perf_counter() - t0)
    n = sample_points or int(min(obs_dev[0].numel(), max(cal, rate * 12.0)))
    sub = [o[:n].cpu().numpy() for o in obs_dev]
    out = np.zeros(n)
    best = float("inf")
    for _ in range(4):  # ~10 s of single-thread CPU work at 1e8 points
        t0 = time.perf_counter()
        pyoracle.linear_regular(dims, starts, steps, vals, sub, out)
        best = min(best, time.perf_counter() - t0)
    rec = {
        "value": round(n / best / 1e6, 3),
        "unit": "Mpoints/s",
        "cores": 1,
        "kind": "port",
        "sample": f"first {n} points of rank 0's batch, best of 4, single thread, "
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


def main():
    args = parse()
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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")

    P = args.points
    n = args.grid
    g = np.linspace(-1.0, 1.0, n)
    dims = [n] * NDIMS
    starts = np.full(NDIMS, -1.0)
    steps = np.full(NDIMS, g[1] - g[0])

    # Grid: generated on rank 0, replicated read-only on every GPU by ONE RCCL broadcast.
    vals_dev = torch.empty(n**NDIMS, dtype=torch.float64, device=dev)
    if rank == 0:
        vals_host = np.random.default_rng(1).uniform(-1.0, 1.0, n**NDIMS)
        vals_dev.copy_(torch.from_numpy(vals_host))
    if world > 1:
        if args.backend == "nccl":
            dist.broadcast(vals_dev, src=0)  # RCCL over xGMI, device to device
        else:
            stage = vals_dev.cpu()
            dist.broadcast(stage, src=0)
            vals_dev.copy_(stage)
    it = interpn_amd.Interpolator.regular("linear", dims, starts, steps, vals_dev, device=local_rank,
                                          dtype=np.float64)

    # Observation shard of this rank: i.i.d. uniform over the grid extent, device resident.
    gen = torch.Generator(device=dev)
    gen.manual_seed(3 + 1000 * rank)
    obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(NDIMS)]
    out = torch.empty(P, dtype=torch.float64, device=dev)

    def step():
        it.eval_tensors(obs, out)
        it.finish()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()

    # Kernel duration with HIP events on the launch stream (torch's current stream).
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        it.eval_tensors(obs, out)
        ev[k][1].record()
        it.finish()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(t[0]), float(t[1])

    # Measured device-copy bandwidth (1 read + 1 write of 0.8 GB), reported beside the nominal peak.
    copy_gbps = None
    if rank == 0:
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

    if rank == 0:
        total_points = P * world * args.steps
        value = total_points / elapsed / 1e6
        achieved = P * BYTES_PER_POINT / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                if tj.get("points") == P and tj.get("grid") == n:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        rec = {
            "metric": "Mpoints/sec (f64) per GPU + achieved HBM GB/s vs roofline, 3D linear-regular",
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
                "workload": f"3D multilinear::regular, {n}^3 f64 grid, {P:.0e} random obs per GPU (BASELINE configs[1])",
                "points_per_gpu": P,
                "grid": [n] * NDIMS,
                "sharding": "obs sharded contiguously per rank; grid replicated by one RCCL broadcast; no collective in the loop",
                "value_per_gpu": round(value / world, 1),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_note": ("FETCH_SIZE+WRITE_SIZE (rocprofv3 --pmc, gfx950 x2 read correction): fabric-side bytes of "
                                 "the L2, Infinity-Cache hits included; the excess over the algorithmic 3.2 GB is brick "
                                 "lines of the 5.4 MiB table re-fetched from the 256 MiB Infinity Cache, not HBM re-reads "
                                 "(profiles/README.md)"),
                "kernel": "interpn::k_linear_brick<double,3,false,true,1,2,2,0> (bricked grid copy, quad-cooperative gather, 2 points/lane)",
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_point": BYTES_PER_POINT,
                "measured_copy_GBps": round(copy_gbps, 1),
                "frac_of_measured_copy": round(achieved / copy_gbps, 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            vals_host = vals_dev.cpu().numpy()
            cb, cpu_out, ncpu = cpu_baseline(dims, starts, steps, vals_host, obs, args.cpu_sample)
            # the oracle here is the checker, never the thing measured on the GPU side
            same = bool(np.array_equal(out[:ncpu].cpu().numpy(), cpu_out))
            cb["gpu_matches_bitwise"] = same
            rec["cpu_baseline"] = cb
        print(json.dumps(rec), flush=True)

    it.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
