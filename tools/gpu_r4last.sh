#!/bin/bash
# round 4, after the decoder's serial retry: its GPU test, then the default bench line at this code
TAG=${1:-r4last}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 150 python -m pytest tests/test_zstd_dec.py -m gpu -q -k "serial_retry or selfcheck or golden" > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 240 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -c 6000 $OUT/bench.log
