#!/bin/bash
# One GPU-box visit: parity tests, phase profile, bench line, rocprofv3 kernel stats.  Outputs under gpurun_out/<tag>/.
# usage: tools/gpu_round.sh <tag> [skip-tests|zstd-tests]
TAG=${1:-run}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" == "zstd-tests" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log
elif [ "$2" != "skip-tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log
fi
timeout 300 python tools/gpu_profile.py > $OUT/phase.json 2> $OUT/phase.err; cat $OUT/phase.json

timeout 300 python tools/gpu_profile.py --level 1 > $OUT/phase_l1.json 2>> $OUT/phase.err; cat $OUT/phase_l1.json
for c in silesia-like lz-7zip; do timeout 300 python tools/gpu_profile.py --corpus $c >> $OUT/phase_other.json 2>> $OUT/phase.err; done; cat $OUT/phase_other.json
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -3 $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_prof.json 2> $OUT/prof.err
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md && cat $OUT/kernel_stats.md
find $OUT/prof -name '*.db' -size +20M -delete
