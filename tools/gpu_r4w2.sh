#!/bin/bash
# the three-wave build (168 registers): W7L time against the number of window groups
OUT=gpurun_out/r4w2; mkdir -p $OUT
cp 7-zip-zstd_amd/csrc/libgpucodec.so /tmp/keep.so; cp tools/_variants/libgpucodec_t192w3.so 7-zip-zstd_amd/csrc/libgpucodec.so
for n in 67108864 134217728 167772160 201326592 211900000; do
timeout 300 python bench.py --codec flzma2 --bytes $n --no-cpu-baseline --no-decode-check --steps 3 > $OUT/b_$n.json 2> $OUT/b_$n.err
python - <<PY
import json
d=json.loads(open('$OUT/b_$n.json').read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print($n, 'groups', ($n + 262143) // 262144, 'ms', d['ms_per_step'], 'mf.dp', k['mf.dp'])
PY
done
cp /tmp/keep.so 7-zip-zstd_amd/csrc/libgpucodec.so
