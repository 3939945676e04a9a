#!/usr/bin/env python3
"""Matrix of the headline kernel's ablation variants (tools/libinterpn_ablate.so: the product kernel
source with ABL = 0 full / 1 stream-only / 2 gather-only) over grid size, brick layout and batch
size.  A small batch (2e6 points = 64 MB of streams) keeps coordinates and results resident in the
256 MiB Infinity Cache across launches: if the stream half were HBM-bound it would speed up there.
    python tools/ablate_probe.py
"""
import ctypes, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "libinterpn_ablate.so"))
lib.ablate_create.restype = ctypes.c_void_p
lib.ablate_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double]
lib.ablate_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
lib.ablate_destroy.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
PMAX = 100_000_256 // 512 * 512
gen = torch.Generator(device=dev); gen.manual_seed(3)
obs = [torch.rand(PMAX, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(3)]
out = torch.empty(PMAX, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream


MODES = ("full", "stream", "gather", "nodiv", "ldsdma")


def run_interleaved(h, P, rounds=8, per_round=3):
    """Interleaved rounds in ONE process (cdna_hip_programming.md rule 24): every round times each
    variant `per_round` times back to back; the per-variant median over all rounds is reported, so
    clock / thermal drift hits every variant alike."""
    ms = {m: [] for m in MODES}
    for r in range(rounds + 1):
        for i, m in enumerate(MODES):
            for k in range(per_round):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = lib.ablate_launch(h, i, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), P, stream)
                b.record()
                assert rc == 0, rc
                torch.cuda.synchronize()
                if r > 0:
                    ms[m].append(a.elapsed_time(b))
    return {m: float(np.median(v)) for m, v in ms.items()}


for n in (32, 48, 64, 96, 128):
    vals = torch.rand(n ** 3, dtype=torch.float64, device=dev, generator=gen)
    step = 2.0 / (n - 1)
    for si, sj in ((1, 1), (1, 2), (2, 2)):
        h = lib.ablate_create(vals.data_ptr(), n, si, sj, step)
        assert h
        P = PMAX
        r = run_interleaved(h, P)
        scale = 1e8 / P
        print(json.dumps({"grid": n, "layout": [si, sj], "points": P,
                          "ms_per_1e8": {k: round(v * scale, 3) for k, v in r.items()},
                          "sum_minus_full": round((r["stream"] + r["gather"] - r["full"]) * scale, 3)}), flush=True)
        # the LDS-DMA variant computes real results: they must equal the unmodified kernel's
        lib.ablate_launch(h, 0, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), PMAX, stream)
        torch.cuda.synchronize(); ref = out.clone(); out.zero_()
        lib.ablate_launch(h, 4, obs[0].data_ptr(), obs[1].data_ptr(), obs[2].data_ptr(), out.data_ptr(), PMAX, stream)
        torch.cuda.synchronize()
        print(json.dumps({"grid": n, "layout": [si, sj], "ldsdma_equals_full": bool(torch.equal(ref, out))}), flush=True)
        del ref
        lib.ablate_destroy(h)
