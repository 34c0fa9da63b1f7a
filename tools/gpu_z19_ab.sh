#!/bin/bash
TAG=${1:-z19}; OUT=gpurun_out/$TAG; mkdir -p $OUT
H=7-zip-zstd_amd/csrc/libgpucodec_hooks.so
run() { echo "== $*" ; env "$@" timeout 200 python tools/gpu_ratio.py --lib $H --bytes $((32*1024*1024)) --codecs $CODEC --levels $LV --corpora $CORP 2>&1 | cut -c1-220; }
{
CODEC=zstd; LV=19; CORP=text-zipf,lz-7zip
run GC_SEARCH_DEPTH=8; run GC_SEARCH_DEPTH=24; run GC_SEARCH_DEPTH=64
CODEC=flzma2; LV=5; CORP=real-bin,real-src
run GC_SEARCH_DEPTH=2; run GC_SEARCH_DEPTH=16; run GC_SEARCH_DEPTH=64
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
