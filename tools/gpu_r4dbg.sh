#!/bin/bash
# round 4: the failing order of tests again (parity, then the price-parse file), after the window-cost fix
OUT=gpurun_out/r4dbg; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_price_parse.py -m gpu -q > $OUT/after_fix.log 2>&1; tail -5 $OUT/after_fix.log
timeout 300 python tools/gpu_diag_fl2.py > $OUT/diag_after_fix.log 2>&1; head -3 $OUT/diag_after_fix.log
