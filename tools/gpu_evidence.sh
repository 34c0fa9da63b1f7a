#!/bin/bash
# One short GPU-box visit for the judged numbers of the state at HEAD: the default bench line (the metric, both legs), rocprofv3 kernel
# stats of the same command, FETCH_SIZE / WRITE_SIZE passes per codec.  Outputs under gpurun_out/<tag>/.
# usage: tools/gpu_evidence.sh <tag> [codecs...]
TAG=${1:-evidence}; shift; CODECS=${@:-zstd flzma2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -2 $OUT/bench.err
R=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-check > $R/$OUT/bench_prof.json 2> $R/$OUT/prof.err; cd $R
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md && cat $OUT/kernel_stats.md
rm -rf $OUT/prof
for c in $CODECS; do
  EXTRA=""; [ $c == brotli ] && EXTRA="--bytes 500000000"
  PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/gpu_pmc.sh $TAG/pmc_$c --codec $c $EXTRA > /dev/null 2>&1
  cat $OUT/pmc_$c/pmc.md
done
