#!/bin/bash
# round 4: zstd levels 16-22 through W7L on the device (tests, size against the reference on real data, bench line of config C4) + an experiment build of the
# FLZMA2 leg: 2 KiB windows with a repeat ring of four nodes (21 KiB of LDS: seven waves per CU) timed on the bench.   usage: tools/gpu_r4z.sh <tag>
TAG=${1:-r4z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_price_parse.py tests/test_gpu_ratio_bars.py -m gpu -x -q > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
timeout 900 python tools/gpu_ratio.py --bytes $((32*1024*1024)) --codecs zstd --levels 16,19,22 --corpora real-src,real-bin,text-zipf,silesia-like > $OUT/ratio_z.jsonl 2> $OUT/ratio_z.err; cut -c1-170 $OUT/ratio_z.jsonl
timeout 600 python bench.py --codec zstd --level 19 --bytes 125000000 --steps 3 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_c4.json').read().strip().splitlines()[-1]); print('c4', d['value'], d['ms_per_step'], d['ratio'], d['roofline']['kernel_ms'])
PY
python - <<PY
import os, sys
sys.path.insert(0, '.')
import __graft_entry__ as g
out = 'tools/_variants'; os.makedirs(out, exist_ok=True)
objs = g.compile_hip_objects(os.path.join(g.CSRC, '_obj'))
pobjs = g.compile_hip_objects(os.path.join(out, '_obj'), only={'gc_lz_dpl.hip': ['-DDPL_MR=4u'], 'gc_api.hip': ['-DGC_TEST_HOOKS']})
g.link_hip([p or o for p, o in zip(pobjs, objs)], os.path.join(out, 'libgpucodec_x.so'))
PY
cp 7-zip-zstd_amd/csrc/libgpucodec.so /tmp/keep.so; cp tools/_variants/libgpucodec_x.so 7-zip-zstd_amd/csrc/libgpucodec.so
for w in 0 1; do
GC_DPL_WIN2K=$w timeout 600 python bench.py --codec flzma2 --no-cpu-baseline --steps 5 > $OUT/bench_x$w.json 2> $OUT/bench_x$w.err
python - <<PY
import json
d=json.loads(open('$OUT/bench_x$w.json').read().strip().splitlines()[-1])
print('MR4 win2k=$w', d['value'], d['ms_per_step'], d['ratio'], d['roofline']['kernel_ms']['mf.dp'])
PY
done
cp /tmp/keep.so 7-zip-zstd_amd/csrc/libgpucodec.so
