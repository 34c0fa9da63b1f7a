#!/bin/bash
# round 4, last GPU seconds: brotli GPU tests (device bytes == emulator bytes, reference decoder) with B1's last-distance substitution, then the speed on web-text
TAG=${1:-r4brs}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 45 python -m pytest tests/test_brotli.py -m gpu -q -x > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 30 python tools/gpu_profile.py --codec brotli --bytes 500000000 --corpus web-text --reps 3 2>&1 | cut -c1-420 | tee $OUT/speed.log
