#!/bin/bash
# Hardware counters of the bench workload, one counter group per rocprofv3 pass (separate --pmc passes, kernel trace only).
# usage: tools/gpu_pmc.sh <tag> [bench args...]        env PMC_GROUPS="A B;C D" overrides the counter groups (';' separated)
TAG=${1:-pmc}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
GROUPS_DEFAULT="FETCH_SIZE;WRITE_SIZE;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum;TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
IFS=';' read -ra GROUPS_ARR <<< "${PMC_GROUPS:-$GROUPS_DEFAULT}"
rm -f $OUT/pmc.md
for ctr in "${GROUPS_ARR[@]}"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-check "$@" > $OUT/$name.json 2> $OUT/$name.err
  DB=$(find $OUT/$name -name '*.db' | head -1)
  if [ -n "$DB" ]; then echo "## $ctr" >> $OUT/pmc.md; python tools/rocpd_pmc.py $DB >> $OUT/pmc.md; echo >> $OUT/pmc.md; else echo "## $ctr: no result" >> $OUT/pmc.md; tail -3 $OUT/$name.err >> $OUT/pmc.md; fi
  rm -rf $OUT/$name
done
cat $OUT/pmc.md
