#!/bin/bash
# One GPU-box visit covering the three codecs: parity tests, bench lines, rocprofv3 kernel stats, FETCH/WRITE PMC passes.
# usage: tools/gpu_full.sh <tag>
TAG=${1:-full}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for c in zstd flzma2 brotli; do
  EXTRA=""; [ $c == brotli ] && EXTRA="--bytes 500000000"
  timeout 900 python bench.py --codec $c $EXTRA > $OUT/bench_$c.json 2> $OUT/bench_$c.err; cat $OUT/bench_$c.json; tail -2 $OUT/bench_$c.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$c -- python bench.py --codec $c $EXTRA --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_prof_$c.json 2> $OUT/prof_$c.err
  DB=$(find $OUT/prof_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats_$c.md && cat $OUT/kernel_stats_$c.md
  rm -rf $OUT/prof_$c
  PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/gpu_pmc.sh $TAG/pmc_$c --codec $c $EXTRA > /dev/null 2>&1
done
python tools/gpu_ratio.py > $OUT/ratio.jsonl 2> $OUT/ratio.err; cat $OUT/ratio.jsonl
