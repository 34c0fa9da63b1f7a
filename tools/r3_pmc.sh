#!/bin/bash
# round 3: hardware counters of one tools/gpu_variants.py run, one counter group per rocprofv3 pass (--pmc with --kernel-trace only).
# usage: tools/r3_pmc.sh <tag> "<group1>;<group2>;..." <gpu_variants args...>
TAG=$1; GROUPS_STR=$2; shift 2; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
IFS=';' read -ra GROUPS_ARR <<< "$GROUPS_STR"
rm -f $OUT/pmc.md
cd /tmp
for ctr in "${GROUPS_ARR[@]}"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$name -- python $GRAFT_REPO_ROOT/tools/gpu_variants.py "$@" > $OUT/$name.json 2> $OUT/$name.err
  DB=$(find $OUT/$name -name '*.db' | head -1)
  if [ -n "$DB" ]; then echo "## $ctr" >> $OUT/pmc.md; python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB >> $OUT/pmc.md; echo >> $OUT/pmc.md; else echo "## $ctr: no result" >> $OUT/pmc.md; tail -3 $OUT/$name.err >> $OUT/pmc.md; fi
  rm -rf $OUT/$name
done
cat $OUT/pmc.md
