#!/bin/bash
# round 4: phase B per block in W7 or W7L -- price-parse / parity / FLZMA2 tests, FLZMA2 bench (default, GC_DPL=1 through the hooks library), bench line of config C4, sizes on real data
TAG=${1:-r4s}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_price_parse.py tests/test_flzma2.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
timeout 600 python bench.py --codec flzma2 --no-cpu-baseline --steps 5 > $OUT/bench_fl2.json 2> $OUT/bench_fl2.err
timeout 600 python bench.py --codec zstd --level 19 --bytes 125000000 --steps 3 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
for f in fl2 c4; do python - <<PY
import json
d=json.loads(open('$OUT/bench_$f.json').read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']; print('$f', d['value'], d['ms_per_step'], d['ratio'], 'mf.dp', k.get('mf.dp'))
PY
done
timeout 900 python tools/gpu_ratio.py --bytes 211900000 --codecs flzma2 --corpora real-bin,silesia-like > $OUT/ratio_fl2.jsonl 2> $OUT/ratio.err; cut -c1-170 $OUT/ratio_fl2.jsonl
timeout 900 python tools/gpu_ratio.py --bytes $((64*1024*1024)) --codecs flzma2 --corpora real-src,real-py,text-zipf,lz-7zip >> $OUT/ratio_fl2.jsonl 2>> $OUT/ratio.err; tail -4 $OUT/ratio_fl2.jsonl | cut -c1-170
