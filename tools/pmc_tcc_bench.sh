#!/bin/bash
# L2 (TCC) request / hit / miss counts of the benchmark kernel per channel-cycle: how much of the
# L2's one-request-per-channel-per-clock budget the product kernel uses.
cd /tmp && export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:?run under gpurun}/gpurun_out/pmc_tcc_bench
rm -rf "$OUT" && mkdir -p "$OUT"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_configs.py --only "3D linear re" > $OUT/p$i.log 2>&1 || echo "pass $i failed/timeout"
done <<'CNT'
TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_CYCLE_sum
TCC_HIT_sum TCC_MISS_sum TCC_BUSY_sum TCC_STREAMING_REQ_sum
TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_NORMAL_EVICT_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum
CNT
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_tcc_bench'
agg=collections.OrderedDict()
for f in sorted(glob.glob(out+'/p*/*/*counter_collection.csv')):
    for row in csv.DictReader(open(f)):
        if 'k_linear_brick' not in row['Kernel_Name']: continue
        k='rect64' if 'double, 3, true' in row['Kernel_Name'] else ('reg128' if '1, 1, 2' in row['Kernel_Name'] else 'reg64')
        agg.setdefault(row['Counter_Name'],collections.OrderedDict()).setdefault(k,[]).append(float(row['Counter_Value']))
print("%-30s %14s %14s %14s"%("counter (avg per launch)","reg64","rect64","reg128"))
for c,d in agg.items():
    g=lambda k: sum(d[k])/len(d[k]) if k in d else float('nan')
    print("%-30s %14.4g %14.4g %14.4g"%(c,g('reg64'),g('rect64'),g('reg128')))
PY
