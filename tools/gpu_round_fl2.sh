#!/bin/bash
# FLZMA2 GPU visit: parity tests, per-level kernel timings, bench line, rocprofv3 kernel stats.  usage: tools/gpu_round_fl2.sh <tag>
TAG=${1:-fl2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_flzma2.py tests/test_plugin.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for lv in 1 3 5 9; do timeout 300 python tools/gpu_profile.py --codec flzma2 --level $lv --corpus silesia-like >> $OUT/levels.json 2>> $OUT/levels.err; done
timeout 300 python tools/gpu_profile.py --codec flzma2 --level 5 --corpus text-zipf >> $OUT/levels.json 2>> $OUT/levels.err; cat $OUT/levels.json
timeout 900 python bench.py --codec flzma2 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -- python bench.py --codec flzma2 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md && cat $OUT/kernel_stats.md
find $OUT/prof -name '*.db' -size +20M -delete
