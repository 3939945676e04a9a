#!/bin/bash
# Round profile of the benchmark (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats of bench.py  -> per-kernel average duration
#   2. separate --pmc passes for FETCH_SIZE and WRITE_SIZE (bench.py and a known-byte-count
#      stream kernel with the same 16-B/lane non-temporal access pattern, for calibration)
# Every rocprofv3 call is wrapped in `timeout`; summaries land in gpurun_out/prof_bench/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_bench
rm -rf $OUT && mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -- $R/tools/tune_linear3d 1e8 64 cal > $OUT/cal_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os
R = os.environ['GRAFT_REPO_ROOT']; OUT = R + '/gpurun_out/prof_bench'
def counter(dirname, kernel_substr):
    vals = []
    for f in glob.glob(f'{OUT}/{dirname}/*/*counter_collection.csv'):
        for row in csv.DictReader(open(f)):
            if kernel_substr in row['Kernel_Name']:
                vals.append(float(row['Counter_Value']))
    return sum(vals) / len(vals) if vals else None
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    res[c + '_bench_KiB'] = counter('pmc_' + c, 'k_linear_brick<double, 3') or counter('pmc_' + c, 'k_linear_regular')
    res[c + '_stream_cal_KiB'] = counter('cal_' + c, 'k_stream16nt')
    res[c + '_stream8_cal_KiB'] = counter('cal_' + c, 'k_stream(')
# known byte count of the calibration kernel: reads 3 x 8 B, writes 8 B per point, 1e8 points
P = 100_000_000
if res.get('FETCH_SIZE_stream_cal_KiB'):
    res['fetch_correction'] = 24.0 * P / (res['FETCH_SIZE_stream_cal_KiB'] * 1024)
if res.get('WRITE_SIZE_stream_cal_KiB'):
    res['write_correction'] = 8.0 * P / (res['WRITE_SIZE_stream_cal_KiB'] * 1024)
if res.get('FETCH_SIZE_bench_KiB') and res.get('WRITE_SIZE_bench_KiB') and 'fetch_correction' in res:
    rd = res['FETCH_SIZE_bench_KiB'] * 1024 * res['fetch_correction']
    wr = res['WRITE_SIZE_bench_KiB'] * 1024 * res.get('write_correction', 1.0)
    res['hbm_read_bytes_per_launch'] = rd
    res['hbm_write_bytes_per_launch'] = wr
    res['hbm_bytes_per_launch'] = rd + wr
    res['points'] = P; res['grid'] = 64
    res['note'] = ('FETCH_SIZE/WRITE_SIZE from separate rocprofv3 --pmc passes of bench.py; corrected by the factor '
                   'measured on a stream kernel of known byte count with the same 16-B/lane non-temporal access pattern; FETCH_SIZE counts L2 fabric-side requests, so bricks served by the Infinity Cache are included '
                   '(MI355X_MICROARCH.md: gfx950 FETCH_SIZE under-reports wide coalesced reads)')
json.dump(res, open(OUT + '/traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
for f in glob.glob(f'{OUT}/stats/*/*kernel_stats.csv'):
    print(open(f).read()[:1500])
PY
