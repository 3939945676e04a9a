#!/bin/bash
# TCC counters of the sweep kernel against the brick kernel (one rocprofv3 --pmc pass each set):
#   gpurun --timeout 600 -- bash tools/sweep_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}
OUT="$R/gpurun_out/sweep_pmc"
rm -rf "$OUT" && mkdir -p "$OUT"
export SWEEP_GRIDS=${SWEEP_GRIDS:-64x1x1} SWEEP_SHAPES=${SWEEP_SHAPES:-12x768} SWEEP_CLOCKS=${SWEEP_CLOCKS:-0,0,0}
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/sweep_clock_probe.py > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
out = "$OUT"
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_linear" not in n: continue
        k = n[n.index("k_linear"):].replace("double, ", "").replace("(interpn::SweepArgs<double>)", "")  # one line per instantiation (rows in registers, threads, ..., stamps, cell, parked rows)
        agg.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: (round(sum(v[len(v)//2:]) / len(v[len(v)//2:]) / 1e6, 2), len(v)) for c, v in d.items()}, "(1e6 per launch, later half of the launches)")
PY
