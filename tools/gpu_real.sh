#!/bin/bash
# The BASELINE levels on the real-data corpora + the default bench line.  usage: tools/gpu_real.sh <tag> [nobench]
TAG=${1:-real}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
timeout 240 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs zstd,brotli,flzma2 --corpora real-src,real-py
timeout 240 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs zstd,brotli --corpora real-bin
timeout 240 python tools/gpu_ratio.py --bytes 211900000 --codecs flzma2 --corpora real-bin
timeout 240 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs zstd,flzma2,brotli --corpora text-zipf,lz-7zip,web-text
} > $OUT/real.jsonl 2> $OUT/real.err
cat $OUT/real.jsonl; tail -3 $OUT/real.err
[ "$2" == nobench ] || { timeout 900 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json; tail -2 $OUT/bench.err; }
