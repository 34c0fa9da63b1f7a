#!/bin/bash
# 2 KiB windows (hook GC_DPL_WIN2K, hooks library) on inputs whose window groups do not fill the device: W7L time
OUT=gpurun_out/r4w3; mkdir -p $OUT
cp 7-zip-zstd_amd/csrc/libgpucodec.so /tmp/keep.so; cp tools/_variants/libgpucodec_hooks.so 7-zip-zstd_amd/csrc/libgpucodec.so
for w in 0 1; do for n in 33554432 100663296 134217728; do
GC_DPL_WIN2K=$w timeout 300 python bench.py --codec flzma2 --bytes $n --no-cpu-baseline --no-decode-check --steps 3 > $OUT/b_${w}_$n.json 2> $OUT/b_${w}_$n.err
python - <<PY
import json
d=json.loads(open('$OUT/b_${w}_$n.json').read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('win2k=$w', $n, 'ms', d['ms_per_step'], 'mf.dp', k['mf.dp'], 'ratio', d['ratio'])
PY
done; done
cp /tmp/keep.so 7-zip-zstd_amd/csrc/libgpucodec.so
