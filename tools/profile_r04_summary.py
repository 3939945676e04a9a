#!/usr/bin/env python3
"""Reduces gpurun_out/prof_r04 (tools/profile_r04.sh) to the files committed under profiles/:
traffic per configuration (FETCH_SIZE / WRITE_SIZE with the calibrated corrections, TCC hits and
misses), cfg4's SQ counters, the kernel-stats table of the bench run."""
import collections
import csv
import glob
import json
import os
import re

R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TAG = os.environ.get("PROF_TAG", "prof_r04")  # tools/profile_r05.sh: prof_r05
OUT = R + "/gpurun_out/" + TAG


def rows(pattern):
    for f in sorted(glob.glob(f"{OUT}/{pattern}/*/*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            yield row


def counters(pattern):
    """kernel (short name) -> counter -> mean per launch"""
    agg = collections.OrderedDict()
    for r in rows(pattern):
        m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:24]
        agg.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def last_json(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception:
        return None


cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    d = counters("cal_" + c)
    cal[c] = d.get("k_stream16nt", {}).get(c)
P = 100_000_000
fetch_corr = 24.0 * P / (cal["FETCH_SIZE"] * 1024) if cal.get("FETCH_SIZE") else 2.0
write_corr = 8.0 * P / (cal["WRITE_SIZE"] * 1024) if cal.get("WRITE_SIZE") else 1.0
res = {"fetch_correction": fetch_corr, "write_correction": write_corr,
       "correction_source": "stream kernel of known byte count (tools/tune_linear3d cal), same 16-B/lane non-temporal access pattern"
                            if cal.get("FETCH_SIZE") else "MI355X_MICROARCH.md section HBM (calibration kernel not run)",
       "configs": {}}
main_kernel = {"cfg2": "k_linear_brick", "cfg3": "k_linear_brick", "cfg5": "k_linear_brick", "cfg4": None}
for key in ("cfg2", "cfg3", "cfg5", "cfg4", "cfg2brick"):
    t = last_json(f"{OUT}/{key}.time")
    if not t:
        continue
    per = {}
    for i in (1, 2, 3, 4):
        for k, d in counters(f"{key}_p{i}").items():
            per.setdefault(k, {}).update(d)
    entry = {"config": t["config"], "kernel": t["kernel"], "table_bytes": t["table_bytes"], "points": t["points"], "grid": t["grid"],
             "ndims": t["ndims"], "ms_unprofiled": t["ms"], "algorithmic_bytes": t["algorithmic_bytes"], "kernels": {}}
    tot_rd = tot_wr = 0.0
    for k, d in per.items():
        if not (k.startswith("k_linear") or k.startswith("k_cubic") or k.startswith("k_bin")):  # (k_linear_sweep included)
            continue
        rd = d.get("FETCH_SIZE", 0.0) * 1024 * fetch_corr
        wr = d.get("WRITE_SIZE", 0.0) * 1024 * write_corr
        entry["kernels"][k] = {"fabric_read_bytes": rd, "fabric_write_bytes": wr,
                               **{c: v for c, v in d.items() if c.startswith("TCC") or c.startswith("TCP")}}
        tot_rd += rd
        tot_wr += wr
    entry["fabric_read_bytes_per_evaluation"] = tot_rd
    entry["fabric_write_bytes_per_evaluation"] = tot_wr
    entry["fabric_bytes_per_evaluation"] = tot_rd + tot_wr
    entry["ratio_to_algorithmic"] = round((tot_rd + tot_wr) / t["algorithmic_bytes"], 3) if t["algorithmic_bytes"] else None
    res["configs"][key] = entry
# legacy top-level keys (cfg2 = the headline kernel) for readers of the round-1..3 format
c2 = res["configs"].get("cfg2")
if c2:
    res.update(hbm_read_bytes_per_launch=c2["fabric_read_bytes_per_evaluation"], hbm_write_bytes_per_launch=c2["fabric_write_bytes_per_evaluation"],
               hbm_bytes_per_launch=c2["fabric_bytes_per_evaluation"], points=c2["points"], grid=c2["grid"], kernel=c2["kernel"],
               table_bytes=c2["table_bytes"])
res["source"] = ("profiles/" + TAG.replace("prof_", "") + "_traffic.json: FETCH_SIZE / WRITE_SIZE / TCC counters from separate rocprofv3 --pmc passes of "
                 "tools/bench_configs.py per configuration (tools/profile_" + TAG.replace("prof_", "") + ".sh), read side corrected by the factor measured on a stream "
                 "kernel of known byte count; fabric-side bytes of the L2, Infinity-Cache hits included")
json.dump(res, open(OUT + "/traffic.json", "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} for k, v in res["configs"].items()}, indent=1))

with open(OUT + "/cfg4_counters.txt", "w") as f:
    f.write("# cfg4 (4-D multicubic-regular 32^4 f64, 1e7 random points): per-launch counter averages (rocprofv3 --pmc, separate passes), summed over the chip.\n")
    f.write("# default = sorted by saturation-class pair of dims 0,1 (k_bin_hist + k_bin_scan + k_bin_scatter_records) and evaluated by the persistent\n")
    f.write("# LDS-column kernel (k_cubic_column, dim 0 from per-part Hermite coefficients); binned=0 = the tiled kernel on the points as given\n")
    for mode in (1, 0):
        per = {}
        for i in (1, 2, 3):
            for k, d in counters(f"c4_b{mode}_p{i}").items():
                per.setdefault(k, {}).update(d)
        t = last_json(f"{OUT}/c4_b{mode}.time")
        f.write(f'\n## {"default (sorted + column)" if mode else "binned=0 (in place)"}: {t["ms"] if t else None} ms per evaluation (HIP events, unprofiled run)\n')
        for k, d in per.items():
            if k.startswith("k_cubic") or k.startswith("k_bin"):
                for c, v in d.items():
                    f.write("%-24s %-30s %14.5g\n" % (k, c, v))
                if "SQ_INSTS_VALU" in d and k.startswith("k_cubic"):
                    f.write("%-24s %-30s %14.5g\n" % (k, "VALU instr per 64 points", d["SQ_INSTS_VALU"] / (1e7 / 64)))
                if "SQ_ACTIVE_INST_VALU" in d and "SQ_WAVE_CYCLES" in d:
                    f.write("%-24s %-30s %14.4f\n" % (k, "ACTIVE_INST_VALU / WAVE_CYCLES", d["SQ_ACTIVE_INST_VALU"] / d["SQ_WAVE_CYCLES"]))
print(open(OUT + "/cfg4_counters.txt").read())
for fn in glob.glob(f"{OUT}/stats/*/*kernel_stats.csv"):
    for l in open(fn):
        if "interpn" in l or l.startswith('"Name"'):
            print(l[:240].rstrip())
