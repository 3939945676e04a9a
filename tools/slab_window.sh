#!/bin/bash
# Round-5 experiment behind profiles/REJECTED.md (temporal blocking of the brick table):
#   gpurun --timeout 900 -- bash tools/slab_window.sh
# times of the existing kernel on window-ordered points, then one TCC counter pass (one launch per case).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}
OUT="$R/gpurun_out/slab_window"
mkdir -p "$OUT"
timeout -k 10 420 python3 $R/tools/slab_window_probe.py > $OUT/times.jsonl 2> $OUT/times.err || { echo "times failed"; tail -5 $OUT/times.err; exit 1; }
tail -3 $OUT/times.jsonl
export PROBE_PMC=1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/pmc -- python3 $R/tools/slab_window_probe.py > $OUT/pmc.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc.log; exit 1; }
python3 - <<PY
import csv, glob, json, collections
out = "$OUT"
plan = json.load(open(out + "/plan.json"))
rows = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pmc/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_linear_brick" not in r["Kernel_Name"]:
            continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
assert len(ids) == len(plan), (len(ids), len(plan))
with open(out + "/counters.jsonl", "w") as fh:
    for p, i in zip(plan, ids):
        c = rows[i]
        p.update({k: c.get(k) for k in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum")})
        p["misses_per_point"] = round(c.get("TCC_MISS_sum", 0) / 100000256, 4)
        fh.write(json.dumps(p) + "\n")
        print(json.dumps(p))
PY
