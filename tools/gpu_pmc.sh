#!/bin/bash
# HBM traffic counters of the bench workload, one counter group per pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2).
# usage: tools/gpu_pmc.sh <tag> [bench args...]
TAG=${1:-pmc}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$name -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err
  DB=$(find $OUT/$name -name '*.db' | head -1)
  if [ -n "$DB" ]; then echo "## $ctr" >> $OUT/pmc.md; python tools/rocpd_pmc.py $DB >> $OUT/pmc.md; echo >> $OUT/pmc.md; fi
  rm -rf $OUT/$name
done
cat $OUT/pmc.md
