#!/bin/bash
# kernel stats of the FLZMA2 leg only
OUT=gpurun_out/${1:-r4k}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -- python $OLDPWD/bench.py --codec flzma2 --steps 3 --warmup 1 --no-cpu-baseline --no-decode-check > $OLDPWD/$OUT/bench_prof.json 2> $OLDPWD/$OUT/prof.err; cd $OLDPWD
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md && head -12 $OUT/kernel_stats.md
find $OUT/prof -name '*.db' -size +20M -delete
