#!/bin/bash
# experiment visit: bench lines of one codec under a list of environment settings.  usage: tools/gpu_exp.sh <tag> <codec> "<ENV=..>" "<ENV=..>" ...
TAG=$1; C=$2; shift; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for e in "$@"; do
  echo "== $e" | tee -a $OUT/exp.log
  env $e timeout 600 python bench.py --codec $C --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'MBps': d['value'], 'ms': d['ms_per_step'], 'comp': d['compressed_bytes'], 'kernel_ms': d['roofline']['kernel_ms']}))" | tee -a $OUT/exp.log
done
