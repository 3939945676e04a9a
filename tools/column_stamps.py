#!/usr/bin/env python3
"""cfg4-style 4-D multicubic (regular n^4, P points): the column evaluation's workgroup shapes side
by side — threads per workgroup x workgroups per CU (K-range phases, cubic_column.h) — each checked
bit for bit against the in-place result, with HIP-event times, per-stage times of the sorted
evaluation and the kernel's own time stamps (option debug_stamps: per workgroup start | part known |
histogram | local order | first sub-column landed | end, 100 MHz ticks, plus XCC / CU ids), reduced
to: mean microseconds per stage, workgroups per CU, and the fraction of the launch a CU had no
workgroup between its first and last.

    python tools/column_stamps.py [n] [points] [f32] [lin]
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def analyse(st):
    st = st[st[:, 5] != 0]
    if len(st) == 0:
        return {}
    t = st[:, :6].astype(np.float64) / 100.0  # microseconds
    hw = st[:, 6] & 0xFFFFFFFF
    xcc = st[:, 6] >> 32
    # HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se, [19:16] tg, ...
    cu = ((xcc & 0xF).astype(np.int64) << 8) | ((hw >> 8) & 0xFF).astype(np.int64)
    t0 = t[:, 0].min()
    span = t[:, 5].max() - t0
    stage = {"lookup": t[:, 1] - t[:, 0], "hist": t[:, 2] - t[:, 1], "order": t[:, 3] - t[:, 2], "fill_wait": t[:, 4] - t[:, 3],
             "phases": t[:, 5] - t[:, 4], "total": t[:, 5] - t[:, 0]}
    res = {"workgroups": int(len(st)), "span_us": round(float(span), 1), "cus_seen": int(len(np.unique(cu)))}
    for k, v in stage.items():
        res[k + "_us"] = [round(float(v.mean()), 2), round(float(np.percentile(v, 95)), 2)]
    # per CU: time covered by at least one workgroup / by at least two, relative to the span
    cov1, cov2, last = [], [], []
    for c in np.unique(cu):
        sel = cu == c
        ev = sorted([(a, 1) for a in t[sel, 0]] + [(b, -1) for b in t[sel, 5]])
        depth, prev, one, two = 0, t0, 0.0, 0.0
        for when, d in ev:
            if depth >= 1: one += when - prev
            if depth >= 2: two += when - prev
            depth += d
            prev = when
        cov1.append(one / span)
        cov2.append(two / span)
        last.append((t[sel, 5].max() - t0) / span)
    res["cu_covered_ge1"] = round(float(np.mean(cov1)), 3)
    res["cu_covered_ge2"] = round(float(np.mean(cov2)), 3)
    res["cu_last_end_mean"] = round(float(np.mean(last)), 3)
    res["points_per_wg"] = round(float((st[:, 7] >> 32).mean()), 1)
    own = ((st[:, 7] >> 16) & 0xFFFF).astype(np.float64) / 100.0  # thread 0's own duration
    res["last_wave_after_wave0_us"] = round(float((stage["total"] - own).mean()), 2)
    # gap on a CU between the end of a workgroup's last wave and thread 0 of the next one
    gaps = []
    for c in np.unique(cu):
        sel = np.where(cu == c)[0]
        sel = sel[np.argsort(t[sel, 0])]
        for i, j in zip(sel[:-1], sel[1:]):
            gaps.append(t[j, 0] - t[i, 5])
    if gaps:
        res["gap_to_next_wg_us"] = [round(float(np.mean(gaps)), 2), round(float(np.percentile(gaps, 95)), 2)]
    return res


def main():
    import torch

    import interpn_amd

    args = [a for a in sys.argv[1:] if a not in ("f32", "lin")]
    n = int(args[0]) if len(args) > 0 else 32
    P = int(float(args[1])) if len(args) > 1 else 10_000_000
    dtype = np.float32 if "f32" in sys.argv else np.float64
    tdt = torch.float32 if dtype == np.float32 else torch.float64
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(4)
    g = np.linspace(-1.0, 1.0, n)
    vals = rng.uniform(-1, 1, n**4).astype(dtype)
    it = interpn_amd.Interpolator.regular("cubic", [n] * 4, np.full(4, -1.0, dtype=dtype), np.full(4, g[1] - g[0], dtype=dtype),
                                          vals, "lin" in sys.argv, 0, dtype)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    obs = [torch.rand(P, dtype=tdt, device=dev, generator=gen) * 2.0 - 1.0 for _ in range(4)]
    out = torch.empty(P, dtype=tdt, device=dev)
    stamps = torch.zeros(1 << 17, dtype=torch.int64, device=dev)  # 16384 workgroups x 8 words

    def timed(reps=15):
        for _ in range(3):
            it.eval_tensors(obs, out)
        it.finish()
        ms = []
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            it.eval_tensors(obs, out)
            b.record()
            it.finish()
            ms.append(a.elapsed_time(b))
        return float(np.median(ms)), float(np.min(ms))

    it.set_option("binned", 0)
    t = timed(5)
    ref = out.clone()
    print(json.dumps({"variant": "in place", "ms": round(t[0], 4), "kernel": it.kernel_name()}), flush=True)
    it.set_option("binned", 1)
    it.set_option("column", 1)
    variants = [(768, 2, 0), (768, 1, 0), (768, 2, 0), (768, 2, 5), (384, 1, 0)]
    if os.environ.get("STAMPS_VARIANTS"):
        variants = [tuple(int(v) for v in x.split("x")) for x in os.environ["STAMPS_VARIANTS"].split(",")]
    for var in variants:
        threads, wgs, cpp = var[:3]
        part = var[3] if len(var) > 3 else 0
        it.set_option("column_part", part)
        it.set_option("column_threads", threads)
        it.set_option("column_groups", wgs)
        it.set_option("column_cpp", cpp)
        it.set_option("debug_stamps", 0)
        it.set_option("stage_timing", 0)
        out.fill_(-3.0)
        t = timed()
        same = bool(torch.equal(out, ref))
        it.set_option("stage_timing", 1)
        it.eval_tensors(obs, out)
        it.finish()
        stage = it.stage_ms() if hasattr(it, "stage_ms") else None
        it.set_option("stage_timing", 0)
        stamps.zero_()
        it.set_option("debug_stamps_bytes", stamps.numel() * stamps.element_size())
        it.set_option("debug_stamps", stamps.data_ptr())
        it.eval_tensors(obs, out)
        it.finish()
        torch.cuda.synchronize()
        it.set_option("debug_stamps", 0)
        st = stamps.cpu().numpy().view(np.uint64).reshape(-1, 8)
        if os.environ.get("STAMPS_DIR"):
            np.save(os.path.join(os.environ["STAMPS_DIR"], f"stamps_{threads}x{wgs}_{len(os.listdir(os.environ['STAMPS_DIR']))}.npy"), st[st[:, 5] != 0])
        print(json.dumps({"threads": threads, "groups": wgs, "cpp": cpp, "part": part, "ms": round(t[0], 4), "min": round(t[1], 4), "stage_ms": stage,
                          "kernel": it.kernel_name(), "bit_identical_to_in_place": same, "stamps": analyse(st)}), flush=True)
    it.close()


if __name__ == "__main__":
    main()
