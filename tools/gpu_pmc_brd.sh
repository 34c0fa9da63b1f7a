mkdir -p gpurun_out/brd7; export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/brd7/p_$tag -- python tools/gpu_brotli_dec.py --only-own --reps 1 > gpurun_out/brd7/run_$tag.log 2>&1
  DB=$(find gpurun_out/brd7/p_$tag -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_pmc.py $DB | grep -i "brotli_dec_kernel\|counter" >> gpurun_out/brd7/pmc.md
  rm -rf gpurun_out/brd7/p_$tag
done
cat gpurun_out/brd7/pmc.md
