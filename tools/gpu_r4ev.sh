#!/bin/bash
# round 4 evidence visit: the default bench line (the metric, with real_data and the CPU baselines), bench lines of configs C2 / C4 / C5's share, rocprofv3 kernel
# stats of the metric run.  Outputs under gpurun_out/<tag>/.   usage: tools/gpu_r4ev.sh <tag>
TAG=${1:-r4ev}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 6000 $OUT/bench.json; tail -2 $OUT/bench.err
timeout 400 python bench.py --codec zstd --bytes 100000000 --steps 10 --no-cpu-baseline > $OUT/bench_c2.json 2>> $OUT/bench.err
timeout 600 python bench.py --codec zstd --level 19 --bytes 125000000 --steps 3 --no-cpu-baseline > $OUT/bench_c4.json 2>> $OUT/bench.err
timeout 600 python bench.py --codec brotli --bytes 1250000000 --steps 3 --no-cpu-baseline > $OUT/bench_c5.json 2>> $OUT/bench.err
for f in c2 c4 c5; do python - <<PY
import json
try:
    d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['metric'], d['value'], d['ms_per_step'], d['ratio'])
except Exception as e: print('$f failed', e)
PY
done
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-check > $OLDPWD/$OUT/bench_prof.json 2> $OLDPWD/$OUT/prof.err; cd $OLDPWD
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md && head -40 $OUT/kernel_stats.md
find $OUT/prof -name '*.db' -size +20M -delete
