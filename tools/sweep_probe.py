#!/usr/bin/env python3
"""The sweep evaluation (interpn_amd/csrc/linear_sweep.h) against the brick kernel on the same
table, unordered points: bitwise comparison, HIP-event medians, alternating in one process."""
import ctypes, json, os, sys
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libinterpn_ablate.so"))
lib.ablate_create.restype = ctypes.c_void_p
lib.ablate_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
lib.ablate_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
lib.ablate_launch_sweep.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
lib.ablate_destroy.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
P = int(float(os.environ.get("SWEEP_POINTS", "100000256"))) // 512 * 512
gen = torch.Generator(device=dev); gen.manual_seed(3)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
stream = torch.cuda.current_stream(dev).cuda_stream
SHAPES = [(4, 1024), (8, 1024), (8, 768), (12, 768), (16, 768), (8, 512), (16, 512), (24, 512), (16, 256), (32, 256)]
if os.environ.get("SWEEP_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split("x")) for s in os.environ["SWEEP_SHAPES"].split(",")]
GRIDS = [(64, 1, 1), (64, 1, 2), (48, 1, 1), (80, 1, 1), (128, 1, 1)]
if os.environ.get("SWEEP_GRIDS"):
    GRIDS = [tuple(int(v) for v in s.split("x")) for s in os.environ["SWEEP_GRIDS"].split(",")]


def timed(fn, reps):
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); rc = fn(); b.record()
        assert rc == 0, rc
        ev.append((a, b))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


for n, si, sj in GRIDS:
    vals = torch.rand(n ** 3, dtype=torch.float64, device=dev, generator=gen)
    h = lib.ablate_create(vals.data_ptr(), n, si, sj, 2.0 / (n - 1))
    ref = torch.empty(P, dtype=torch.float64, device=dev)
    base = lambda: lib.ablate_launch(h, 0, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), ref.data_ptr(), P, stream)
    timed(base, 2)
    for K, th in SHAPES:
        out = torch.full((P,), -7.0, dtype=torch.float64, device=dev)
        sweep = lambda: lib.ablate_launch_sweep(h, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), P, K, th, 0, stream)
        rc = sweep()
        if rc == -1:
            continue
        assert rc == 0, rc
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        tb, ts = [], []
        for _ in range(4):
            tb += timed(base, 3)
            ts += timed(sweep, 3)
        print(json.dumps({"grid": n, "layout": [si, sj], "K": K, "threads": th, "points_on_chip": K * th * 256, "bitwise_equal": same,
                          "brick_ms_per_1e8": round(float(np.median(tb)) * 1e8 / P, 4), "sweep_ms_per_1e8": round(float(np.median(ts)) * 1e8 / P, 4)}), flush=True)
        del out
    lib.ablate_destroy(h)
