#!/bin/bash
# A/B of FLZMA2 level 2-4 configurations through the test hooks (the hooks build).  usage: tools/gpu_fl2_levels_ab.sh <tag>
TAG=${1:-fl2ab}; OUT=gpurun_out/$TAG; mkdir -p $OUT
H=7-zip-zstd_amd/csrc/libgpucodec_hooks.so
run() { echo "== $*" ; env "$@" timeout 120 python tools/gpu_ratio.py --lib $H --bytes $((32*1024*1024)) --codecs flzma2 --levels $LV --corpora silesia-like,text-zipf 2>&1 | cut -c1-220; }
{
LV=2; run GC_SEG_LOG=15; run GC_SEG_LOG=17
LV=3; run GC_SEG_LOG=17; run GC_PRICE_PARSE=1 GC_SHORT_PASS=1; run GC_PRICE_PARSE=1 GC_SHORT_PASS=1 GC_SEG_LOG=17; run GC_PRICE_PARSE=1 GC_SHORT_PASS=0 GC_SEG_LOG=17
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
