#!/bin/bash
TAG=${1:-quick2}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
timeout 200 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs flzma2 --corpora real-src,real-py
timeout 200 python tools/gpu_ratio.py --bytes 211900000 --codecs flzma2 --corpora real-bin
timeout 200 python tools/gpu_ratio.py --bytes $((32*1024*1024)) --codecs zstd --levels 7,12 --corpora text-zipf,lz-7zip
timeout 200 python tools/gpu_ratio.py --bytes $((32*1024*1024)) --codecs brotli --levels 9 --corpora web-text,text-zipf
} > $OUT/ratio.jsonl 2> $OUT/ratio.err
cat $OUT/ratio.jsonl; tail -2 $OUT/ratio.err
timeout 300 python bench.py --codec flzma2 --no-cpu-baseline --steps 5 > $OUT/bench_fl2.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench_fl2.json
PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" bash tools/gpu_pmc.sh $TAG/pmc_flzma2 --codec flzma2 > /dev/null 2>&1
grep "deepen" $OUT/pmc_flzma2/pmc.md
