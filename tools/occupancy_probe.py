#!/usr/bin/env python3
"""Is the headline kernel latency-bound?  The product kernel source (tools/libinterpn_ablate.so)
launched with extra dynamic LDS per workgroup, so that fewer workgroups fit a CU (24 KiB of its
own: 6 workgroups = 6 waves per SIMD; +8 KiB -> 5, +16 -> 4, +29 -> 3, +56 -> 2 (64 KiB limit w/o opt-in: 1 beyond))."""
import ctypes, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libinterpn_ablate.so"))
lib.ablate_create.restype = ctypes.c_void_p
lib.ablate_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
lib.ablate_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
lib.ablate_destroy.argtypes = [ctypes.c_void_p]
lib.ablate_set_extra_lds.argtypes = [ctypes.c_size_t]
dev = torch.device("cuda:0")
P = 100_000_256 // 512 * 512
gen = torch.Generator(device=dev); gen.manual_seed(3)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
out = torch.empty(P, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
EXTRA = [0, 8, 16, 29, 39]  # KiB -> 6, 5, 4, 3, 2 workgroups per CU (dynamic LDS <= 64 KiB without opt-in)
MODES = {"full": 0, "stream": 1, "gather": 2}
for n, si, sj in [(64, 1, 2), (48, 1, 1), (128, 1, 1)]:
    vals = torch.rand(n ** 3, dtype=torch.float64, device=dev, generator=gen)
    h = lib.ablate_create(vals.data_ptr(), n, si, sj, 2.0 / (n - 1))
    ms = {(e, m): [] for e in EXTRA for m in MODES}
    for r in range(6):
        for e in EXTRA:
            lib.ablate_set_extra_lds(e * 1024)
            for m, code in MODES.items():
                ev = []
                for k in range(3):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    rc = lib.ablate_launch(h, code, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), P, stream)
                    b.record(); assert rc == 0, rc
                    ev.append((a, b))
                torch.cuda.synchronize()
                if r: ms[(e, m)] += [a.elapsed_time(b) for a, b in ev]
    for e in EXTRA:
        wg = (160 * 1024) // (24 * 1024 + e * 1024)
        print(json.dumps({"grid": n, "layout": [si, sj], "extra_lds_KiB": e, "workgroups_per_CU": min(wg, 6),
                          **{m + "_ms": round(float(np.median(ms[(e, m)])) * 1e8 / P, 4) for m in MODES}}), flush=True)
    lib.ablate_destroy(h)
