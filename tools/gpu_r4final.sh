#!/bin/bash
# round 4, evidence at the round's last code: the whole GPU suite, then tools/gpu_r4ev.sh (smoke, default bench line, C2 / C4 / C5 lines, rocprofv3 kernel stats)
TAG=${1:-r4final}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/tests_full.log 2>&1; tail -4 $OUT/tests_full.log
bash tools/gpu_r4ev.sh $TAG
