#!/bin/bash
# round 4: the FLZMA2 level table against the reference (32 MiB per corpus) + an experiment build of W7L (named -D flag) timed on the bench.  usage: tools/gpu_r4lv.sh <tag> [flag]
TAG=${1:-r4lv}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 python tools/gpu_ratio.py --bytes $((32*1024*1024)) --codecs flzma2 --levels 1,2,3,7,9 --corpora text-zipf,lz-7zip,silesia-like > $OUT/levels.jsonl 2> $OUT/levels.err; cat $OUT/levels.jsonl
if [ -n "$2" ]; then
python - <<PY
import os, sys
sys.path.insert(0, '.')
import __graft_entry__ as g
out = 'tools/_variants'; os.makedirs(out, exist_ok=True)
objs = g.compile_hip_objects(os.path.join(g.CSRC, '_obj'))
pobjs = g.compile_hip_objects(os.path.join(out, '_obj'), only={'gc_lz_dpl.hip': ['-D$2']})
g.link_hip([p or o for p, o in zip(pobjs, objs)], os.path.join(out, 'libgpucodec_x.so'))
PY
cp 7-zip-zstd_amd/csrc/libgpucodec.so /tmp/keep.so; cp tools/_variants/libgpucodec_x.so 7-zip-zstd_amd/csrc/libgpucodec.so
timeout 600 python bench.py --codec flzma2 --no-cpu-baseline --steps 5 > $OUT/bench_x.json 2> $OUT/bench_x.err
cp /tmp/keep.so 7-zip-zstd_amd/csrc/libgpucodec.so
python - <<PY
import json
d=json.loads(open('$OUT/bench_x.json').read().strip().splitlines()[-1])
print('$2', d['value'], d['ms_per_step'], d['ratio'], d['roofline']['kernel_ms']['mf.dp'])
PY
fi
