#!/bin/bash
# round 4: the decoder after the serial retry of an overlapped batch -- GPU decoder tests, then the decoder's kernel stats + FETCH / WRITE passes
# (tools/gpu_pmc_dec_wide.sh: under --pmc the launches are serialised; the self-check frame must now pass and gc_zstd_dec_seqv_kernel must be the one measured)
TAG=${1:-r4dec}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 200 python -m pytest tests/test_zstd_dec.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/tests_dec.log 2>&1; tail -3 $OUT/tests_dec.log
bash tools/gpu_pmc_dec_wide.sh $TAG > $OUT/pmc_run.log 2>&1; cat $OUT/pmc.md | cut -c1-160
