#!/bin/bash
TAG=${1:-quick3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
timeout 200 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs brotli --corpora real-src,real-bin,real-py,web-text,lz-7zip
timeout 200 python tools/gpu_ratio.py --bytes $((32*1024*1024)) --codecs brotli --levels 5,9 --corpora web-text,lz-7zip
} > $OUT/ratio.jsonl 2> $OUT/ratio.err
cat $OUT/ratio.jsonl; tail -2 $OUT/ratio.err
timeout 300 python bench.py --codec brotli --bytes 500000000 --no-cpu-baseline --steps 5 > $OUT/bench_br.json 2> $OUT/bench.err; tail -c 1200 $OUT/bench_br.json
