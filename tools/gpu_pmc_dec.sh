#!/bin/bash
# HBM traffic counters of the decoder kernels on the metric's own stream: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (kernel trace only).
TAG=${1:-pmcdec}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rm -f $OUT/pmc.md
for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  name=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$name -- python tools/gpu_zstd_dec_once.py > $OUT/$name.log 2> $OUT/$name.err
  DB=$(find $OUT/$name -name '*.db' | head -1)
  if [ -n "$DB" ]; then echo "## $ctr" >> $OUT/pmc.md; python tools/rocpd_pmc.py $DB | grep -v "gc_mf_\|gc_zstd_lz\|gc_zstd_huf\|gc_zstd_seq_kernel \|gc_zstd_plan\|gc_zstd_emit\|rocclr" >> $OUT/pmc.md; echo >> $OUT/pmc.md; else echo "## $ctr: no result" >> $OUT/pmc.md; tail -3 $OUT/$name.err >> $OUT/pmc.md; fi
  rm -rf $OUT/$name
done
cat $OUT/pmc.md
