#!/bin/bash
TAG=${1:-depth}; OUT=gpurun_out/$TAG; mkdir -p $OUT
H=7-zip-zstd_amd/csrc/libgpucodec_hooks.so
{
for D in 2 6 12; do
  echo "== GC_SEARCH_DEPTH=$D"
  GC_SEARCH_DEPTH=$D timeout 200 python tools/gpu_ratio.py --lib $H --bytes $((64*1024*1024)) --codecs flzma2 --levels 5 --corpora real-src,real-py,silesia-like 2>&1 | cut -c1-220
  GC_SEARCH_DEPTH=$D timeout 200 python tools/gpu_profile.py --lib $H --codec flzma2 --bytes 211900000 --corpus silesia-like --reps 3 2>&1 | cut -c1-900
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
