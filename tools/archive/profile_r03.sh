#!/bin/bash
# Round-3 profile session (gpurun --timeout 1200 -- bash tools/profile_r03.sh).
#   A. rocprofv3 --kernel-trace --stats of the driver's bench command (every configuration's kernels,
#      incl. cfg4's sort + LDS-column pipeline), and of the same command with the process group forced
#      (--force-dist --backend nccl: the RCCL calls of the multi-GPU path on one rank)
#   B. separate --pmc passes FETCH_SIZE / WRITE_SIZE of the headline kernel + the known-byte-count
#      stream kernel (calibration, MI355X_MICROARCH.md section HBM)
#   C. cfg4: TCC / SQ / LDS counter passes of the sort kernels and of k_cubic_column (binned = 1,
#      the default) and of the tiled kernel with the points evaluated in place (binned = 0)
# Every rocprofv3 call is wrapped in `timeout`; PMC passes use --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r03
rm -rf $OUT && mkdir -p $OUT
PY=python3
echo "A" ; date
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PY $R/bench.py --steps 20 --warmup 5 > $OUT/stats_bench.json 2> $OUT/stats.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_nccl -- $PY $R/bench.py --steps 20 --warmup 5 --force-dist --backend nccl --no-configs --no-cpu-baseline --no-ablate > $OUT/stats_nccl_bench.json 2> $OUT/stats_nccl.err
echo "B" ; date
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- $PY $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs --no-ablate --sustain-seconds 0 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
  if [ -x $R/tools/tune_linear3d ]; then
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -- $R/tools/tune_linear3d 1e8 64 cal > $OUT/cal_$c.log 2>&1
  fi
done
echo "C" ; date
for mode in 1 0; do
  export INTERPN_HIP_BINNED=$mode
  [ $mode = 1 ] && unset INTERPN_HIP_BINNED
  i=0
  while read -r line; do
    [ -z "$line" ] && continue
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/c4_b${mode}_p$i -- $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_b${mode}_p$i.log 2>&1 || echo "cfg4 binned=$mode pass $i failed/timeout"
  done <<'CNT'
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES
CNT
  timeout -k 10 200 $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_b${mode}.time 2>&1
done
unset INTERPN_HIP_BINNED
date
$PY - <<'PY'
import csv, glob, json, os, collections
R = os.environ['GRAFT_REPO_ROOT']; OUT = R + '/gpurun_out/prof_r03'
def rows(pattern):
    for f in sorted(glob.glob(f'{OUT}/{pattern}/*/*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            yield row
def counter(dirname, kernel_substr):
    vals = [float(r['Counter_Value']) for r in rows(dirname) if kernel_substr in r['Kernel_Name']]
    return sum(vals) / len(vals) if vals else None
bench = None
try:
    bench = json.loads([l for l in open(OUT + '/pmc_FETCH_SIZE.json') if l.startswith('{')][-1])
except Exception as e:
    print('no bench record from the FETCH_SIZE pass:', e)
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    res[c + '_bench_KiB'] = counter('pmc_' + c, 'k_linear_brick<double, 3')
    res[c + '_stream_cal_KiB'] = counter('cal_' + c, 'k_stream16nt')
P = 100_000_000
if res.get('FETCH_SIZE_stream_cal_KiB'):
    res['fetch_correction'] = 24.0 * P / (res['FETCH_SIZE_stream_cal_KiB'] * 1024)
else:
    res['fetch_correction'] = 2.0
    res['fetch_correction_source'] = 'MI355X_MICROARCH.md section HBM (calibration kernel not run)'
if res.get('WRITE_SIZE_stream_cal_KiB'):
    res['write_correction'] = 8.0 * P / (res['WRITE_SIZE_stream_cal_KiB'] * 1024)
if res.get('FETCH_SIZE_bench_KiB') and res.get('WRITE_SIZE_bench_KiB'):
    rd = res['FETCH_SIZE_bench_KiB'] * 1024 * res['fetch_correction']
    wr = res['WRITE_SIZE_bench_KiB'] * 1024 * res.get('write_correction', 1.0)
    res.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, hbm_bytes_per_launch=rd + wr, points=P, grid=64)
    if bench:
        res['kernel'] = bench['roofline']['kernel']
        res['table_bytes'] = bench['roofline'].get('table_bytes')
    res['source'] = ('profiles/r03_bench_traffic.json: FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes of bench.py '
                     '(tools/profile_r03.sh), read side corrected by the factor measured on a stream kernel of known byte count '
                     'with the same 16-B/lane non-temporal access pattern; fabric-side bytes of the L2, Infinity-Cache hits included')
json.dump(res, open(OUT + '/traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
with open(OUT + '/cfg4_counters.txt', 'w') as f:
    f.write('# cfg4 (4-D multicubic-regular 32^4 f64, 1e7 random points): per-launch counter averages (rocprofv3 --pmc, separate passes), summed over the chip.\n')
    f.write('# default = sorted by saturation-class pair of dims 0,1 (k_bin_hist + k_bin_scan + k_bin_scatter_records) and evaluated out of an LDS-resident\n')
    f.write('# table column (k_cubic_column); binned=0 = the tiled kernel on the points as given\n')
    for mode in (1, 0):
        agg = collections.OrderedDict()
        for r in rows(f'c4_b{mode}_p*'):
            import re
            m = re.search(r'(k_[a-z0-9_]+)', r['Kernel_Name'])
            k = m.group(1) if m else r['Kernel_Name'][:24]
            agg.setdefault((k, r['Counter_Name']), []).append(float(r['Counter_Value']))
        ms = None
        try:
            for l in open(f'{OUT}/c4_b{mode}.time'):
                if l.startswith('{'): ms = json.loads(l)['ms']
        except Exception: pass
        f.write(f'\n## {"default (sorted + column)" if mode else "binned=0 (in place)"}: {ms} ms per evaluation (HIP events, unprofiled run)\n')
        for (k, c), v in agg.items():
            if k.startswith('k_cubic') or k.startswith('k_bin'):
                f.write('%-24s %-30s %14.5g\n' % (k, c, sum(v) / len(v)))
print(open(OUT + '/cfg4_counters.txt').read())
for d in ('stats', 'stats_nccl'):
    for fn in glob.glob(f'{OUT}/{d}/*/*kernel_stats.csv'):
        print('==', d)
        for l in open(fn):
            if 'interpn' in l or 'ccl' in l.lower() or l.startswith('"Name"'): print(l[:240].rstrip())
PY
