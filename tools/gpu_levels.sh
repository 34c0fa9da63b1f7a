#!/bin/bash
# Sizes of every accepted level against the reference encoder at the same level, and of the BASELINE levels on the real-data corpora
# (tools/gpu_ratio.py lines) -> gpurun_out/<tag>/levels.jsonl.   usage: tools/gpu_levels.sh <tag>
TAG=${1:-levels}; OUT=gpurun_out/$TAG; mkdir -p $OUT
N=$((32*1024*1024))
{
timeout 240 python tools/gpu_ratio.py --bytes $N --codecs zstd --levels 1,2,16,22 --corpora text-zipf,lz-7zip
timeout 240 python tools/gpu_ratio.py --bytes $N --codecs flzma2 --levels 1,3,7,9 --corpora text-zipf,lz-7zip,silesia-like
timeout 240 python tools/gpu_ratio.py --bytes $N --codecs brotli --levels 1,4,9,11 --corpora text-zipf,lz-7zip,web-text
timeout 240 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs zstd,brotli --corpora real-src,real-bin,real-py
timeout 240 python tools/gpu_ratio.py --bytes 211900000 --codecs flzma2 --corpora real-bin
timeout 240 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs flzma2 --corpora real-src,real-py
} > $OUT/levels.jsonl 2> $OUT/levels.err
cat $OUT/levels.jsonl; tail -3 $OUT/levels.err
