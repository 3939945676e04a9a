#!/usr/bin/env python3
"""Interleaved A/B of builds of the headline kernel that differ in a compile-time switch
(tools/_variants/*.so = tools/ablate_linear3d.hip compiled with -D...; the first entry is the
library's own build, tools/libinterpn_ablate.so).  Outputs must be equal.
    python tools/variant_probe.py [variant.so ...]"""
import ctypes, glob, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = [os.path.join(ROOT, "tools", "libinterpn_ablate.so")] + (sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "*.so"))))
libs = []
for p in paths:
    lib = ctypes.CDLL(p)
    lib.ablate_create.restype = ctypes.c_void_p
    lib.ablate_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
    lib.ablate_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
    lib.ablate_destroy.argtypes = [ctypes.c_void_p]
    libs.append((os.path.basename(p), lib))
dev = torch.device("cuda:0")
P = 100_000_256 // 512 * 512
gen = torch.Generator(device=dev); gen.manual_seed(3)
obs = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
out = torch.empty(P, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
MODES = {"full": 0, "gather": 2}
for n, si, sj in [(64, 1, 2), (64, 2, 2), (48, 1, 1), (32, 1, 1), (128, 1, 1)]:
    vals = torch.rand(n ** 3, dtype=torch.float64, device=dev, generator=gen)
    hs = [(name, lib, lib.ablate_create(vals.data_ptr(), n, si, sj, 2.0 / (n - 1))) for name, lib in libs]
    ref, same = None, {}
    for name, lib, h in hs:
        out.zero_()
        assert lib.ablate_launch(h, 0, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), P, stream) == 0
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        same[name] = bool(torch.equal(out, ref))
    ms = {(name, m): [] for name, _, _ in hs for m in MODES}
    for r in range(7):
        for name, lib, h in hs:
            for m, code in MODES.items():
                ev = []
                for k in range(4):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    rc = lib.ablate_launch(h, code, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), P, stream)
                    b.record(); assert rc == 0
                    ev.append((a, b))
                torch.cuda.synchronize()
                if r: ms[(name, m)] += [a.elapsed_time(b) for a, b in ev]
    for name, lib, h in hs:
        print(json.dumps({"grid": n, "layout": [si, sj], "build": name, "equal_to_base": same[name],
                          **{m + "_ms": round(float(np.median(ms[(name, m)])) * 1e8 / P, 4) for m in MODES}}), flush=True)
        lib.ablate_destroy(h)
