#!/bin/bash
# A short, bounded GPU visit: named ratio runs + the bench line.  usage: tools/gpu_quick.sh <tag>
TAG=${1:-quick}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
timeout 150 python tools/gpu_ratio.py --bytes $((32*1024*1024)) --codecs flzma2 --levels 1,2 --corpora silesia-like,random
timeout 150 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs brotli --corpora real-src,real-bin,real-py
timeout 150 python tools/gpu_ratio.py --bytes $((32*1024*1024)) --codecs brotli --levels 1,4,9 --corpora web-text,lz-7zip
} > $OUT/ratio.jsonl 2> $OUT/ratio.err
cat $OUT/ratio.jsonl; tail -2 $OUT/ratio.err
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json; tail -2 $OUT/bench.err
