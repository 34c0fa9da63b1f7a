#!/bin/bash
# BROTLI GPU visit: parity tests, kernel timings, bench line, rocprofv3 kernel stats.  usage: tools/gpu_round_br.sh <tag>
TAG=${1:-br}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_brotli.py tests/test_plugin.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
for c in web-text text-zipf silesia-like; do timeout 300 python tools/gpu_profile.py --codec brotli --corpus $c >> $OUT/levels.json 2>> $OUT/levels.err; done; cat $OUT/levels.json
timeout 900 python bench.py --codec brotli --bytes 500000000 > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -- python bench.py --codec brotli --bytes 500000000 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md && cat $OUT/kernel_stats.md
find $OUT/prof -name '*.db' -size +20M -delete
