#!/bin/bash
# One GPU-box visit for the ratio work: sizes against the reference (tools/gpu_ratio.py) + the default bench line.
# usage: tools/gpu_ratio_round.sh <tag> [ratio args...]
TAG=${1:-ratio}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python tools/gpu_ratio.py "$@" > $OUT/ratio.jsonl 2> $OUT/ratio.err; cat $OUT/ratio.jsonl; tail -3 $OUT/ratio.err
timeout 600 python bench.py --no-decode-check --no-cpu-baseline --steps 5 > $OUT/bench.json 2> $OUT/bench.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json"))
    print("zstd", d["value"], d["ms_per_step"], {k:v for k,v in d["roofline"]["kernel_ms"].items()})
    f=d["flzma2_l5_silesia"]; print("fl2", f["value"], f["ms_per_step"], f["compressed_bytes"], {k:v for k,v in f["roofline"]["kernel_ms"].items()})
except Exception as e: print("bench failed", e); print(open("$OUT/bench.err").read()[-2000:])
PY
