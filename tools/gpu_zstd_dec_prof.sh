#!/bin/bash
# GPU: phase cycles of the decoder's execution kernel + rocprofv3 kernel stats of one decode
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
GC_ZD_PROF=1 timeout 200 python tools/gpu_zstd_dec_rate.py ${1:-268435456} 2>&1 | grep -v "Exception ignored\|Traceback\|File \|AttributeError" | awk '!seen[$0]++' | tail -24
mkdir -p gpurun_out/zdprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/zdprof -o zd -- python tools/gpu_zstd_dec_rate.py ${1:-268435456} > gpurun_out/zdprof/run.log 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/zdprof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if 'zstd_dec' in r['Name']:
            print(r['Name'][:40], 'calls', r['Calls'], 'total ms %.2f' % (float(r['TotalDurationNs'])/1e6), 'avg ms %.3f' % (float(r['AverageNs'])/1e6))
PY
