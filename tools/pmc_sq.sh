#!/bin/bash
# SQ-side counters (instruction mix, busy / wait cycles, LDS conflicts) of the 3-D linear kernels,
# regular (cfg2) against rectilinear (cfg3): where do the extra 0.3 ms of cfg3 go?
cd /tmp && export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:?run under gpurun}/gpurun_out/pmc_sq
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only "3D linear re" > $OUT/p$i.log 2>&1 || echo "pass $i failed/timeout"
done <<'CNT'
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_CYCLE_sum
CNT
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_sq'
agg=collections.OrderedDict()
for f in sorted(glob.glob(out+'/p*/*/*counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        if 'k_linear_brick' not in row['Kernel_Name']: continue
        k='rect' if 'double, 3, true' in row['Kernel_Name'] else ('reg128' if '1, 1, 2' in row['Kernel_Name'] else 'reg64')
        agg.setdefault(row['Counter_Name'],collections.OrderedDict()).setdefault(k,[]).append(float(row['Counter_Value']))
print("%-28s %14s %14s %14s"%("counter (avg per launch)","reg64","rect64","reg128"))
for c,d in agg.items():
    g=lambda k: sum(d[k])/len(d[k]) if k in d else float('nan')
    print("%-28s %14.4g %14.4g %14.4g"%(c,g('reg64'),g('rect'),g('reg128')))
PY
