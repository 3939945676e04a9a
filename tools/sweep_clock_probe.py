#!/usr/bin/env python3
"""Sweep evaluation (linear_sweep.h): time against the clock period that aligns the waves' rows, and
the alignment itself from the kernel's own time stamps (STAMPS build): for every XCD, the mean
resultant length R of the sweep positions of its workgroups that are in their rows at a moment
(1 = all at the same slab of the table, ~0 = spread evenly over it), averaged over the launch."""
import ctypes, json, os, sys
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libinterpn_ablate.so"))
lib.ablate_create.restype = ctypes.c_void_p
lib.ablate_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
lib.ablate_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
lib.ablate_launch_sweep.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.ablate_set_sweep_clock.argtypes = [ctypes.c_uint]
lib.ablate_set_sweep_parked.argtypes = [ctypes.c_int]
lib.ablate_set_sweep_abl.argtypes = [ctypes.c_int]
lib.ablate_set_sweep_fastdiv.argtypes = [ctypes.c_uint]
lib.ablate_set_sweep_stamps.argtypes = [ctypes.c_void_p]
lib.ablate_destroy.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
P = 100_000_256 // 512 * 512
gen = torch.Generator(device=dev); gen.manual_seed(3)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
if os.environ.get("SWEEP_PRESORT"):  # points ordered by x0 inside windows of 2^w points (the table then stays in the L2: what is left is everything but the misses)
    w = 1 << int(os.environ["SWEEP_PRESORT"])
    for lo in range(0, P, w):
        idx = torch.argsort(obs[0][lo:lo + w])
        for d in range(3):
            obs[d][lo:lo + w] = obs[d][lo:lo + w][idx]
    del idx
stream = torch.cuda.current_stream(dev).cuda_stream
SHAPES = [(8, 1024), (8, 512), (12, 768), (16, 768), (16, 512)]
if os.environ.get("SWEEP_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SWEEP_SHAPES"].split(",")]  # K x threads [x workgroups per CU]
GRIDS = [(64, 1, 1), (64, 1, 2)]
if os.environ.get("SWEEP_GRIDS"):
    GRIDS = [tuple(int(v) for v in s.split("x")) for s in os.environ["SWEEP_GRIDS"].split(",")]
CLOCKS = [1, 0, 1700, 2100, 2500]  # 1: no clock (rows in sorted order), 0: the period the previous launch measured
if os.environ.get("SWEEP_CLOCKS"):
    CLOCKS = [int(v) for v in os.environ["SWEEP_CLOCKS"].split(",")]


def timed(fn, reps):
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rc = fn(); b.record()
        assert rc == 0, rc
        ev.append((a, b))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def timeline(st, K):
    """st: [nwaves, 8] stamps of the persistent waves -> summary"""
    st = st[st[:, 1] > 0]
    t0 = st[:, 0].min()
    start, end = [(st[:, i] - t0).astype(np.float64) * 0.01 for i in range(2)]  # microseconds
    other, rows = st[:, 2].astype(np.float64) * 0.01, st[:, 3].astype(np.float64) * 0.01
    rounds = np.maximum(st[:, 6].astype(np.float64), 1)
    return {"waves": int(len(st)), "launch_us": round(float(end.max()), 1), "first_wave_done_us": round(float(end.min()), 1),
            "round_us_median": round(float(np.median((end - start) / rounds)), 2), "rows_us_per_round": round(float(np.median(rows / rounds)), 2),
            "other_us_per_round": round(float(np.median(other / rounds)), 2), "in_rows_fraction": round(float(rows.sum() / (end - start).sum()), 3),
            "period_us": round(float(np.median(st[:, 7])) * 0.01, 2), "xcd_end_us": [round(float(end[st[:, 4] == x].mean()), 0) if (st[:, 4] == x).any() else None for x in range(8)],
            "end_us_pct": [round(float(np.percentile(end, q)), 0) for q in (0, 10, 50, 90, 100)]}


for n, si, sj in GRIDS:
    vals = torch.rand(n ** 3, dtype=torch.float64, device=dev, generator=gen)
    h = lib.ablate_create(vals.data_ptr(), n, si, sj, 2.0 / (n - 1))
    ref = torch.empty(P, dtype=torch.float64, device=dev)
    base = lambda: lib.ablate_launch(h, 0, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), ref.data_ptr(), P, stream)
    tb = float(np.median(timed(base, 6)[2:])) * 1e8 / P
    for shape in SHAPES:
        K, th = shape[0], shape[1]
        wgs = shape[2] if len(shape) > 2 else 0
        kl = shape[3] if len(shape) > 3 else 0  # K x threads x workgroups per CU (0: default) x rows parked in LDS [x measurement build: 1 no table access, 2 no streams]
        abl = shape[4] if len(shape) > 4 else 0
        fastdiv = shape[5] if len(shape) > 5 else 1  # 0: the divide sequences (the kernel before step_cell_fast)
        lib.ablate_set_sweep_fastdiv(fastdiv)
        lib.ablate_set_sweep_parked(kl)
        lib.ablate_set_sweep_abl(abl)
        out = torch.full((P,), -7.0, dtype=torch.float64, device=dev)
        sweep = lambda: lib.ablate_launch_sweep(h, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), P, K, th, wgs, stream)
        stamps = torch.zeros(256 * 16 * 4 * 8, dtype=torch.int64, device=dev)
        for ck in CLOCKS:
            lib.ablate_set_sweep_clock(ck)
            lib.ablate_set_sweep_stamps(None)
            out.fill_(-7.0)
            rc = sweep()
            if rc == -1:  # shape / layout not instantiated: the brick kernel's time alone
                print(json.dumps({"grid": n, "layout": [si, sj], "brick_ms_per_1e8": round(tb, 4)}), flush=True)
                break
            assert rc == 0, rc
            torch.cuda.synchronize()
            same = bool(torch.equal(out, ref))
            ms = float(np.median(timed(sweep, 7)[2:])) * 1e8 / P
            lib.ablate_set_sweep_stamps(stamps.data_ptr())
            stamps.zero_()
            assert sweep() == 0
            torch.cuda.synchronize()
            raw = stamps.cpu().numpy().reshape(-1, 8)
            if os.environ.get("SWEEP_DUMP"):
                os.makedirs(os.path.join(ROOT, "gpurun_out", "sweep_dump"), exist_ok=True)
                np.save(os.path.join(ROOT, "gpurun_out", "sweep_dump", f"st_{n}_{si}{sj}_{K}x{th}_{ck}.npy"), raw)
            al = timeline(raw, K)
            lib.ablate_set_sweep_stamps(None)
            print(json.dumps({"grid": n, "layout": [si, sj], "K": K, "threads": th, "wgs_per_cu": wgs, "parked_rows": kl, "ablation": abl, "fastdiv": fastdiv, "clock_us": ck / 100.0, "bitwise_equal": same,
                              "brick_ms_per_1e8": round(tb, 4), "sweep_ms_per_1e8": round(ms, 4), **al}), flush=True)
        del out, stamps
    lib.ablate_destroy(h)
