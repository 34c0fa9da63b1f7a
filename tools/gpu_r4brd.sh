#!/bin/bash
# round 4: brotli quality 6 with ONE / TWO links followed (hook GC_SEARCH_DEPTH; round 3 measured 0 / 4 / 8): size on the real corpora, speed on web-text
TAG=${1:-r4brd}; OUT=gpurun_out/$TAG; mkdir -p $OUT
H=7-zip-zstd_amd/csrc/libgpucodec_hooks.so
{
for D in 1 2; do
  echo "== brotli q6 GC_SEARCH_DEPTH=$D"
  GC_SEARCH_DEPTH=$D timeout 120 python tools/gpu_ratio.py --lib $H --bytes $((64*1024*1024)) --codecs brotli --levels 6 --corpora real-py,real-src 2>&1 | cut -c1-220
  GC_SEARCH_DEPTH=$D timeout 100 python tools/gpu_profile.py --lib $H --codec brotli --bytes 500000000 --corpus web-text --reps 3 2>&1 | cut -c1-400
done
} > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
