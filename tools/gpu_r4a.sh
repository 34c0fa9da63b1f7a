#!/bin/bash
# round 4, first GPU visit: the lane-per-window price parse (W7L) on the device -- FLZMA2 tests, sizes on real data and stand-ins, timing.  usage: tools/gpu_r4a.sh <tag>
TAG=${1:-r4a}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_flzma2.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
{
timeout 300 python tools/gpu_ratio.py --bytes 211900000 --codecs flzma2 --corpora real-bin,silesia-like
timeout 300 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs flzma2 --corpora real-src,real-py,text-zipf,lz-7zip,web-text
} > $OUT/ratio.jsonl 2> $OUT/ratio.err
cat $OUT/ratio.jsonl; tail -3 $OUT/ratio.err
timeout 600 python bench.py --codec flzma2 --no-cpu-baseline > $OUT/bench_fl2.json 2> $OUT/bench_fl2.err; tail -c 3000 $OUT/bench_fl2.json; tail -2 $OUT/bench_fl2.err
