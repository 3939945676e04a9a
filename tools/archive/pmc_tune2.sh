#!/bin/bash
# L2-side stall counters for the tuning-harness kernels (full vs gather-only), one pass per group.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tune2
rm -rf $OUT && mkdir -p $OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- $GRAFT_REPO_ROOT/tools/tune_layout 1e8 64 prof > $OUT/p$i.log 2>&1 || echo "pass $i failed/timeout"
done <<'CNT'
TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_READ_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_STREAMING_REQ_sum TCC_NORMAL_WRITEBACK_sum
TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_WRITE_sum TCC_NORMAL_EVICT_sum
CNT
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_tune2'
agg=collections.OrderedDict()
for f in sorted(glob.glob(out+'/p*/*/*counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        k=row['Kernel_Name'][:48]; c=row['Counter_Name']; v=float(row['Counter_Value'])
        agg.setdefault(c,collections.OrderedDict()).setdefault(k,[]).append(v)
ks=[]
for c in agg:
    for k in agg[c]:
        if k not in ks: ks.append(k)
for i,k in enumerate(ks): print("K%d = %s"%(i,k))
for c,d in agg.items():
    print("%-42s"%c+"".join("%14.4g"%(sum(d.get(k,[0]))/max(1,len(d.get(k,[0])))) for k in ks))
PY
