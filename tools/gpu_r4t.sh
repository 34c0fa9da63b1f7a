#!/bin/bash
# round 4: quick timing of the FLZMA2 leg + its GPU tests.  usage: tools/gpu_r4t.sh <tag> [notests]
TAG=${1:-r4t}; OUT=gpurun_out/$TAG; mkdir -p $OUT
[ "$2" == notests ] || { timeout 600 python -m pytest tests/test_flzma2.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log; }
timeout 300 python tools/gpu_ratio.py --bytes 211900000 --codecs flzma2 --corpora real-bin > $OUT/ratio.jsonl 2> $OUT/ratio.err; cat $OUT/ratio.jsonl
timeout 600 python bench.py --codec flzma2 --no-cpu-baseline --steps 5 > $OUT/bench_fl2.json 2> $OUT/bench_fl2.err; python - <<PY
import json
d=json.loads(open('$OUT/bench_fl2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])
PY
