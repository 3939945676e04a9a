#!/bin/bash
# Round-2 profile session (run on the GPU box: gpurun --timeout 2400 -- ./tools/profile_r02.sh).
#   A. rocprofv3 --kernel-trace --stats of the driver's bench command (every configuration's kernel)
#   B. separate --pmc passes FETCH_SIZE / WRITE_SIZE of the headline kernel + a stream kernel of
#      known byte count in the same access pattern (calibration, MI355X_MICROARCH.md section HBM)
#   C. cfg4 (4-D multicubic 32^4): TCC and SQ counter passes of the product kernel
#   D. cfg4 under every tile layout (INTERPN_HIP_BRICKS=off|44|24|22|14|11): HIP-event time, and
#      TCC hit / miss / fabric-read counters per layout
# Every rocprofv3 call is wrapped in `timeout`; PMC passes use --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r02
rm -rf $OUT && mkdir -p $OUT
PY=python3

# ---- A
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $PY $R/bench.py --steps 20 --warmup 5 > $OUT/stats_bench.json 2> $OUT/stats.err
# ---- B
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- $PY $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-configs --no-ablate --sustain-seconds 0 > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err
  if [ -x $R/tools/tune_linear3d ]; then
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -- $R/tools/tune_linear3d 1e8 64 cal > $OUT/cal_$c.log 2>&1
  fi
done
# ---- C
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/c4_p$i -- $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_p$i.log 2>&1 || echo "cfg4 pass $i failed/timeout"
done <<'CNT'
TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum
TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_NC_READ_REQ_sum
SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
GRBM_GUI_ACTIVE
CNT
# ---- D
for lay in off 44 24 22 14 11 auto; do
  if [ $lay = auto ]; then unset INTERPN_HIP_BRICKS; else export INTERPN_HIP_BRICKS=$lay; fi
  timeout 200 $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_layout_$lay.time 2>&1
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/c4_layout_$lay -- $PY $R/tools/bench_configs.py --only "cfg4 4D cubic regular 32^4 1e7 (linearize=false)" > $OUT/c4_layout_$lay.log 2>&1 || echo "layout $lay pmc failed"
done
unset INTERPN_HIP_BRICKS
# ---- E: host-pointer paths (PCIe-inclusive) and small-call latency; 4-D linear after the cell bricks
timeout 300 $PY $R/tools/bench_host_path.py > $OUT/host_path.txt 2>&1
timeout 300 $PY $R/tools/bench_configs.py > $OUT/bench_configs.txt 2>&1

$PY - <<'PY'
import csv, glob, json, os, collections
R = os.environ['GRAFT_REPO_ROOT']; OUT = R + '/gpurun_out/prof_r02'
def rows(pattern):
    for f in sorted(glob.glob(f'{OUT}/{pattern}/*/*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            yield row
def counter(dirname, kernel_substr):
    vals = [float(r['Counter_Value']) for r in rows(dirname) if kernel_substr in r['Kernel_Name']]
    return sum(vals) / len(vals) if vals else None
# B: traffic record
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
    res['source'] = ('profiles/r02_bench_traffic.json: FETCH_SIZE / WRITE_SIZE from separate rocprofv3 --pmc passes of bench.py '
                     '(tools/profile_r02.sh), read side corrected by the factor measured on a stream kernel of known byte count '
                     'with the same 16-B/lane non-temporal access pattern; fabric-side bytes of the L2, Infinity-Cache hits included')
json.dump(res, open(OUT + '/traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
# C: cfg4 counters
agg = collections.OrderedDict()
for r in rows('c4_p*'):
    if 'k_cubic_brick' in r['Kernel_Name']:
        agg.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
with open(OUT + '/cfg4_counters.txt', 'w') as f:
    f.write('# cfg4: k_cubic_brick<double, 4, false, true, 1, 1>, 32^4 f64 grid, 1e7 random points; averages per launch, summed over the chip\n')
    for c, v in agg.items():
        f.write('%-32s %14.5g\n' % (c, sum(v) / len(v)))
print(open(OUT + '/cfg4_counters.txt').read())
# D: layouts
with open(OUT + '/cfg4_layouts.txt', 'w') as f:
    f.write('# cfg4 (4-D multicubic-regular 32^4 f64, 1e7 random points) under every tile layout of dims 0,1: HIP-event median (unprofiled run),\n')
    f.write('# and per launch TCC_REQ / TCC_HIT / TCC_MISS / TCC_EA0_RDREQ (separate rocprofv3 --pmc run)\n')
    f.write('%-6s %9s %12s %12s %12s %12s %10s\n' % ('layout', 'ms', 'TCC_REQ', 'TCC_HIT', 'TCC_MISS', 'EA0_RDREQ', 'lines/pt'))
    for lay in ('off', '44', '24', '22', '14', '11', 'auto'):
        ms = None
        try:
            for l in open(f'{OUT}/c4_layout_{lay}.time'):
                if l.startswith('{'):
                    ms = json.loads(l)['ms']
        except Exception:
            pass
        c = {}
        for r in rows(f'c4_layout_{lay}'):
            if 'k_cubic' in r['Kernel_Name']:
                c.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
        g = lambda k: (sum(c[k]) / len(c[k])) if k in c else float('nan')
        f.write('%-6s %9s %12.4g %12.4g %12.4g %12.4g %10.2f\n' % (lay, ms, g('TCC_REQ_sum'), g('TCC_HIT_sum'), g('TCC_MISS_sum'),
                                                                   g('TCC_EA0_RDREQ_sum'), g('TCC_REQ_sum') / 1e7))
print(open(OUT + '/cfg4_layouts.txt').read())
for f in glob.glob(f'{OUT}/stats/*/*kernel_stats.csv'):
    print(open(f).read()[:3000])
PY
