#!/usr/bin/env python3
"""Sweep grid sizes x tile layouts for multicubic f64 (validates the layout model of
abi_layout.hip::maybe_build_cubic_tiles).  Run on the GPU box; one process, env re-read per handle."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import interpn_amd

dev = torch.device("cuda:0")
P = 10_000_000
gen = torch.Generator(device=dev); gen.manual_seed(5)
obs_all = [torch.rand(P, dtype=torch.float64, device=dev, generator=gen) * 2 - 1 for _ in range(4)]
out = torch.empty(P, dtype=torch.float64, device=dev)


def run(N, n, lay):
    if lay == "auto": os.environ.pop("INTERPN_HIP_BRICKS", None)
    else: os.environ["INTERPN_HIP_BRICKS"] = lay
    g = np.linspace(-1, 1, n)
    vals = np.random.default_rng(1).uniform(-1, 1, n ** N)
    it = interpn_amd.Interpolator.regular("cubic", [n] * N, np.full(N, -1.0), np.full(N, g[1] - g[0]), vals)
    obs = obs_all[:N]
    it.eval_tensors(obs, out); it.finish()
    ms = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); it.eval_tensors(obs, out); b.record(); it.finish(); ms.append(a.elapsed_time(b))
    it.close()
    return round(sorted(ms)[2], 3)


for N, sizes in ((2, (64, 128, 256, 512, 1024, 2048, 4096)), (3, (16, 24, 32, 48, 64, 96, 128, 192)), (4, (8, 12, 16, 24, 32, 40))):
    for n in sizes:
        row = {}
        for lay in ("off", "44", "24", "22", "14", "11", "auto"):
            try:
                row[lay] = run(N, n, lay)
            except Exception as e:  # noqa: BLE001
                row[lay] = "err"
        best = min((v, k) for k, v in row.items() if isinstance(v, float) and k != "auto")
        row["best"] = best[1]
        row["auto_vs_best"] = round(row["auto"] / best[0], 3) if isinstance(row["auto"], float) else None
        print(N, n, json.dumps(row), flush=True)
