#!/bin/bash
TAG=${1:-z12}; OUT=gpurun_out/$TAG; mkdir -p $OUT
H=7-zip-zstd_amd/csrc/libgpucodec_hooks.so
run() { echo "== $*" ; env "$@" timeout 150 python tools/gpu_ratio.py --lib $H --bytes $((32*1024*1024)) --codecs zstd --levels $LV --corpora lz-7zip,text-zipf 2>&1 | cut -c1-220; }
{
LV=12; run GC_PRICE_PARSE=1 GC_SHORT_PASS=1; run GC_PRICE_PARSE=1 GC_SHORT_PASS=0
LV=10; run GC_PRICE_PARSE=0
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
timeout 200 python tools/gpu_ratio.py --bytes $((32*1024*1024)) --codecs flzma2 --levels 2,3,4 --corpora silesia-like,text-zipf,lz-7zip
