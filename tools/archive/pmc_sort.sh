#!/bin/bash
# TCC write/read counters of the sort kernels (k_bin_hist / k_bin_scan / k_bin_scatter_records) and of the
# column kernel on cfg4, from the files a previous tools/profile_r03.sh left, or collected here.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_sort
rm -rf $OUT && mkdir -p $OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done <<'CNT'
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_WRITEBACK_sum
CNT
python3 - <<'PY'
import csv, glob, os, re, collections
R = os.environ['GRAFT_REPO_ROOT']; OUT = R + '/gpurun_out/pmc_sort'
agg = collections.OrderedDict()
for fn in sorted(glob.glob(f'{OUT}/p*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(fn)):
        m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
        if not m: continue
        agg.setdefault((m.group(1), r['Counter_Name']), []).append(float(r['Counter_Value']))
with open(OUT + '/summary.txt', 'w') as f:
    for (k, c), v in agg.items():
        f.write('%-24s %-26s %14.5g  (n=%d)\n' % (k, c, sum(v) / len(v), len(v)))
print(open(OUT + '/summary.txt').read())
PY
