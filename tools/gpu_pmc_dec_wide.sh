#!/bin/bash
# Decoder on the metric's own stream (wide execution): rocprofv3 kernel stats, then FETCH_SIZE and WRITE_SIZE in separate passes (kernel trace only).
TAG=${1:-pmcdecw}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -- python $R/tools/gpu_zstd_dec_once.py > $R/$OUT/stats.log 2>&1 )
DB=$(find $OUT/prof -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py $DB | grep "zstd_dec\|fillBuffer\|^| kernel\|^|---" > $OUT/kernel_stats.md
rm -rf $OUT/prof; rm -f $OUT/pmc.md
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --pmc $ctr --kernel-trace -d $R/$OUT/$ctr -- python $R/tools/gpu_zstd_dec_once.py > $R/$OUT/$ctr.log 2> $R/$OUT/$ctr.err )
  DB=$(find $OUT/$ctr -name '*.db' | head -1)
  if [ -n "$DB" ]; then echo "## $ctr" >> $OUT/pmc.md; python tools/rocpd_pmc.py $DB | grep "zstd_dec\|^| kernel\|^|---" >> $OUT/pmc.md; echo >> $OUT/pmc.md; else echo "## $ctr: no result" >> $OUT/pmc.md; tail -3 $OUT/$ctr.err >> $OUT/pmc.md; fi
  rm -rf $OUT/$ctr
done
cat $OUT/kernel_stats.md $OUT/pmc.md
