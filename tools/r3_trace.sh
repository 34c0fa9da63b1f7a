#!/bin/bash
# round 3: rocprofv3 kernel trace of a tools/gpu_variants.py run -> per-kernel table.  usage: tools/r3_trace.sh <tag> <gpu_variants args...>
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -- python $GRAFT_REPO_ROOT/tools/gpu_variants.py "$@" > $GRAFT_REPO_ROOT/$OUT/run.jsonl 2> $GRAFT_REPO_ROOT/$OUT/prof.err
cd $GRAFT_REPO_ROOT
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_summary.py $DB > $OUT/kernel_stats.md && cat $OUT/kernel_stats.md
find $OUT/prof -name '*.db' -size +20M -delete
cat $OUT/run.jsonl
