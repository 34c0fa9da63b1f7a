#!/bin/bash
TAG=${1:-brdepth}; OUT=gpurun_out/$TAG; mkdir -p $OUT
H=7-zip-zstd_amd/csrc/libgpucodec_hooks.so
{
for D in 0 4 8; do
  echo "== brotli q6 GC_SEARCH_DEPTH=$D"
  GC_SEARCH_DEPTH=$D timeout 200 python tools/gpu_ratio.py --lib $H --bytes $((64*1024*1024)) --codecs brotli --levels 6 --corpora real-src,real-bin,real-py,web-text 2>&1 | cut -c1-220
  GC_SEARCH_DEPTH=$D timeout 200 python tools/gpu_profile.py --lib $H --codec brotli --bytes 500000000 --corpus web-text --reps 3 2>&1 | cut -c1-400
done
echo "== flzma2 L5 as built (depth 6, capped records skipped)"
timeout 200 python tools/gpu_profile.py --codec flzma2 --bytes 211900000 --corpus silesia-like --reps 3 2>&1 | cut -c1-700
timeout 200 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs flzma2 --levels 5 --corpora real-src,real-py 2>&1 | cut -c1-220
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
