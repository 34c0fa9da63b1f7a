#!/bin/bash
# experiment builds of W7L timed on the FLZMA2 leg: usage tools/gpu_r4v.sh <tag> <lib tags...>
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
cp 7-zip-zstd_amd/csrc/libgpucodec.so /tmp/keep.so
for v in "$@"; do
cp tools/_variants/libgpucodec_$v.so 7-zip-zstd_amd/csrc/libgpucodec.so
timeout 600 python bench.py --codec flzma2 --no-cpu-baseline --no-decode-check --steps 5 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], d['ratio'], d['roofline']['kernel_ms']['mf.dp'])
PY
done
cp /tmp/keep.so 7-zip-zstd_amd/csrc/libgpucodec.so
