#!/bin/bash
# One GPU-box visit at the end of a round: the whole -m gpu suite, smoke, the default bench line (the metric), the other codecs' bench lines,
# rocprofv3 kernel stats of the metric run.  Outputs under gpurun_out/<tag>/.
# usage: tools/gpu_final.sh <tag>
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -2 $OUT/bench.err
timeout 600 python bench.py --codec brotli --steps 5 > $OUT/bench_brotli.json 2>> $OUT/bench.err; cat $OUT/bench_brotli.json
timeout 600 python bench.py --codec zstd --level 19 --bytes 125000000 --steps 3 > $OUT/bench_c4.json 2>> $OUT/bench.err; cat $OUT/bench_c4.json
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-check > $OLDPWD/$OUT/bench_prof.json 2> $OLDPWD/$OUT/prof.err; cd $OLDPWD
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md && cat $OUT/kernel_stats.md
find $OUT/prof -name '*.db' -size +20M -delete
