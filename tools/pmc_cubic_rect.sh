#!/bin/bash
# SQ-side counters of the rectilinear 3-D multicubic kernels (tiled kernel in place and the sweep kernel), 64^3 f64, 1e7 points:
#   gpurun --timeout 600 -- bash tools/pmc_cubic_rect.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}
OUT="$R/gpurun_out/pmc_cubic_rect"
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/cubic_sweep_probe.py ${PROBE_ARGS:-rect 64 points=1e7} > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done <<'CNT'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
CNT
python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_cubic" not in n: continue
        k = n[n.index("k_cubic"):].split("(")[0]
        agg.setdefault(r["Counter_Name"], collections.OrderedDict()).setdefault(k, []).append(float(r["Counter_Value"]))
kernels = []
for d in agg.values():
    for k in d:
        if k not in kernels: kernels.append(k)
print("kernels:", kernels)
for c, d in agg.items():
    print("%-24s" % c, "  ".join("%12.4g" % (sum(d[k]) / len(d[k])) if k in d else "%12s" % "-" for k in kernels))
PY
