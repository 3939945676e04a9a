#!/usr/bin/env python3
"""Upper bound of temporal blocking for the 3-D multilinear brick kernel (round-4 review, item 1).

A kernel that orders a chunk of points by leading-index slab on chip and sweeps the slabs in phase
across an XCD makes every table line an L2 hit ONLY while (points per sweep and XCD) >> (lines of the
table): every sweep re-fetches the whole table once.  This probe measures that trade-off with the
EXISTING kernel (tools/libinterpn_ablate.so = the product kernel source): the batch is pre-ordered
(untimed, torch) by the cell index of dimension 0 inside windows of W points, so that consecutive
workgroups — dealt round-robin over the 8 XCDs in dispatch order — sweep the table once per window;
W / 8 is the number of points an XCD sees per sweep.  What any on-chip ordering can hold is bounded
by the LDS: 256 CUs x 160 KiB / 24 B per point = 1.7e6 points chip-wide, i.e. W <= 2^20..2^21.

    python tools/slab_window_probe.py            # times (HIP events, median of 5)
    PROBE_PMC=1 rocprofv3 --pmc ... -- python3 tools/slab_window_probe.py   # one launch per case, plan -> gpurun_out/slab_window/plan.json
"""
import ctypes, json, os, sys
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libinterpn_ablate.so"))
lib.ablate_create.restype = ctypes.c_void_p
lib.ablate_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
lib.ablate_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
lib.ablate_destroy.argtypes = [ctypes.c_void_p]
lib.ablate_set_extra_lds.argtypes = [ctypes.c_size_t]
PMC = bool(int(os.environ.get("PROBE_PMC", "0")))
dev = torch.device("cuda:0")
P = 100_000_256 // 512 * 512
gen = torch.Generator(device=dev); gen.manual_seed(3)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
out = torch.empty(P, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
idx = torch.arange(P, device=dev, dtype=torch.int64)


def ordered(n, log2w, shift):
    """obs sorted by (cell index along dim 0) >> shift inside windows of 2^log2w points"""
    if log2w == 0:
        return obs
    step = 2.0 / (n - 1)
    cell = torch.clamp(torch.floor((obs[0] + 1.0) / step), 0, n - 2).to(torch.int64) >> shift
    key = (idx >> log2w) * 256 + cell
    perm = torch.argsort(key)
    del key, cell
    res = [o[perm].contiguous() for o in obs]
    del perm
    return res


def run(h, o, reps):
    ev = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = lib.ablate_launch(h, 0, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), out.data_ptr(), P, stream)
        b.record(); assert rc == 0, rc
        ev.append((a, b))
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


if PMC:
    CASES = [(64, 1, 1), (64, 1, 2), (128, 1, 1)]
    WINDOWS = [0, 18, 20, 21, 22, 24]
    EXTRA = [0]
else:
    CASES = [(64, 1, 1), (64, 1, 2), (64, 2, 2), (48, 1, 1), (128, 1, 1)]
    WINDOWS = [0, 17, 18, 19, 20, 21, 22, 23, 24, 26]
    EXTRA = [0, 39]  # KiB of extra dynamic LDS: 6 / 2 resident workgroups per CU (fewer points in flight = a tighter phase)
plan = []
for n, si, sj in CASES:
    vals = torch.rand(n ** 3, dtype=torch.float64, device=dev, generator=gen)
    h = lib.ablate_create(vals.data_ptr(), n, si, sj, 2.0 / (n - 1))
    nb = lambda nn, s: (nn - 1) if s == 1 else ((nn - 1) // 2 + 1)
    lines = nb(n, si) * nb(n, sj) * ((n - 2) // 3 + 1)
    for w in WINDOWS:
        for shift in ([0] if w == 0 else [0, 2]):
            o = ordered(n, w, shift)
            for e in EXTRA:
                lib.ablate_set_extra_lds(e * 1024)
                if PMC:
                    run(h, o, 1)
                    plan.append({"grid": n, "layout": [si, sj], "table_lines": lines, "log2_window": w, "key_shift": shift, "extra_lds_KiB": e})
                    continue
                run(h, o, 2)
                ms = float(np.median(run(h, o, 5))) * 1e8 / P
                rec = {"grid": n, "layout": [si, sj], "table_MiB": round(lines * 128 / 2**20, 2), "table_lines": lines,
                       "log2_window": w, "key_shift": shift, "points_per_xcd_and_sweep": (2 ** w) // 8 if w else None,
                       "compulsory_misses_per_point": round(lines * 8 / 2 ** w, 3) if w else None,
                       "workgroups_per_cu": 6 if e == 0 else 2, "ms_per_1e8": round(ms, 4), "frac_of_8TBps": round(32e8 / ms / 1e-3 / 8e12, 4)}
                print(json.dumps(rec), flush=True)
            if o is not obs:
                del o
    lib.ablate_destroy(h)
if PMC:
    d = os.path.join(ROOT, "gpurun_out", "slab_window")
    os.makedirs(d, exist_ok=True)
    json.dump(plan, open(os.path.join(d, "plan.json"), "w"))
