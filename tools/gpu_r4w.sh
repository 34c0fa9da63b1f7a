#!/bin/bash
# W7L time against the number of window groups (is the launch one round of resident groups?): the FLZMA2 leg on inputs of several sizes
OUT=gpurun_out/r4w; mkdir -p $OUT
for n in 33554432 67108864 100663296 134217728 167772160 211900000 423800000; do
timeout 300 python bench.py --codec flzma2 --bytes $n --no-cpu-baseline --no-decode-check --steps 3 > $OUT/b_$n.json 2> $OUT/b_$n.err
python - <<PY
import json
d=json.loads(open('$OUT/b_$n.json').read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print($n, 'groups', ($n + 262143) // 262144, 'ms', d['ms_per_step'], 'mf.dp', k['mf.dp'], 'model', k['model'])
PY
done
