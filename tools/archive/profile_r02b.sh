#!/bin/bash
# Round-2 (second session) profile: rocprofv3 --kernel-trace --stats of the driver's bench command
# (every configuration's kernels, the binned 4-D multicubic path included), and TCC counter passes
# of cfg4 with the points evaluated in place (INTERPN_HIP_BINNED=0) and binned (=1).
# Every rocprofv3 call is wrapped in `timeout`; PMC passes use --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02b
rm -rf $OUT && mkdir -p $OUT
PY=python3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PY $R/bench.py --steps 20 --warmup 5 > $OUT/stats_bench.json 2> $OUT/stats.err
for mode in 0 1; do
  export INTERPN_HIP_BINNED=$mode
  i=0
  while read -r line; do
    [ -z "$line" ] && continue
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/c4_b${mode}_p$i -- $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_b${mode}_p$i.log 2>&1 || echo "cfg4 binned=$mode pass $i failed/timeout"
  done <<'CNT'
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY
CNT
  timeout 200 $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_b${mode}.time 2>&1
done
unset INTERPN_HIP_BINNED
$PY - <<'PY'
import csv, glob, json, os, collections
R = os.environ['GRAFT_REPO_ROOT']; OUT = R + '/gpurun_out/prof_r02b'
with open(OUT + '/cfg4_binned_counters.txt', 'w') as f:
    f.write('# cfg4 (4-D multicubic-regular 32^4 f64, 1e7 random points): per-launch counter averages (rocprofv3 --pmc, separate passes), summed over the chip,\n')
    f.write('# with the points evaluated in place (binned=0) and counting-sorted by tile position first (binned=1: k_bin_hist + k_bin_scan + k_bin_scatter + k_cubic_brick)\n')
    for mode in (0, 1):
        agg = collections.OrderedDict()
        for fn in sorted(glob.glob(f'{OUT}/c4_b{mode}_p*/*/*counter_collection.csv')):
            for r in csv.DictReader(open(fn)):
                k = r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1]
                agg.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
        ms = None
        try:
            for l in open(f'{OUT}/c4_b{mode}.time'):
                if l.startswith('{'): ms = json.loads(l)['ms']
        except Exception: pass
        f.write(f'\n## binned={mode}: {ms} ms per evaluation (HIP events, unprofiled run)\n')
        for (k, c), v in agg.items():
            if k.startswith('k_cubic') or k.startswith('k_bin'):
                f.write('%-16s %-30s %14.5g\n' % (k, c, sum(v) / len(v)))
print(open(OUT + '/cfg4_binned_counters.txt').read())
for fn in glob.glob(f'{OUT}/stats/*/*kernel_stats.csv'):
    for l in open(fn):
        if 'interpn' in l or l.startswith('"Name"'): print(l[:230].rstrip())
PY
