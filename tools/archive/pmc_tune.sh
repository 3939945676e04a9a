#!/bin/bash
# Collect PMC counters for the tuning harness kernels (one rocprofv3 pass per counter group).
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tune
mkdir -p $OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- $GRAFT_REPO_ROOT/tools/tune_layout 1e8 64 prof > $OUT/p$i.log 2>&1
done <<'CNT'
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum
SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_WAVES
TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE
TCC_TAG_STALL_sum TCC_BUSY_sum TCC_READ_sum TCC_CYCLE_sum
TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCP_LATENCY_sum
CNT
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_tune'
agg=collections.OrderedDict()
for f in sorted(glob.glob(out+'/p*/*/*counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'][:60]; c=row['Counter_Name']; v=float(row['Counter_Value'])
        agg.setdefault(k,collections.OrderedDict()).setdefault(c,[]).append(v)
with open(out+'/summary.txt','w') as fo:
    for k,d in agg.items():
        fo.write(k+'\n')
        for c,vs in d.items():
            fo.write(f'   {c:45s} {sum(vs)/len(vs):16.0f}  (n={len(vs)})\n')
print(open(out+'/summary.txt').read())
PY
