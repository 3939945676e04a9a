#!/bin/bash
# SQ / LDS counters of k_cubic_column on coefficient columns (tools/coef_probe.py, COEF_ONLY): usage: gpurun -- bash tools/pmc_coef.sh [probe args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT is not set)}
OUT=$R/gpurun_out/pmc_coef
rm -rf "$OUT" && mkdir -p "$OUT"
export COEF_ONLY=1
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/coef_probe.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done <<'CNT'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH
CNT
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ['GRAFT_REPO_ROOT']; OUT = R + '/gpurun_out/pmc_coef'
agg = collections.OrderedDict()
for fn in sorted(glob.glob(f'{OUT}/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1]
        agg.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
with open(OUT + '/summary.txt', 'w') as f:
    for (k, c), v in agg.items():
        if k.startswith('k_cubic_column'):
            f.write('%-18s %-28s %14.5g  (n=%d)\n' % (k, c, sum(v) / len(v), len(v)))
print(open(OUT + '/summary.txt').read())
PY
