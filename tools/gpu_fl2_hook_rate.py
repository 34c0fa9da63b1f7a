#!/usr/bin/env python3
"""Kernel times of Fast-LZMA2 level 5 on the Silesia stand-in through the TEST build of the library (csrc/libgpucodec_hooks.so), so that GC_* hooks in the
environment select code paths: A / B timing of experiments.  usage: GC_DPL_WIN2K=1 python tools/gpu_fl2_hook_rate.py [corpus] [bytes]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import __graft_entry__ as g
import oracle as O
pkg = g.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else 'silesia-like'; n = int(sys.argv[2]) if len(sys.argv) > 2 else 211_900_000
x = O.corpus(kind, n)
enc = pkg.Flzma2Encoder(level=5, device=0, lib_path=pkg.HOOKS_LIB_PATH)
d_src = torch.from_numpy(x).to('cuda:0'); cap = enc.compress_bound(x.size); d_dst = torch.empty(cap, dtype=torch.uint8, device='cuda:0')
for _ in range(3):
    enc.code_device(d_src.data_ptr(), x.size, d_dst.data_ptr(), cap); c = enc.finish()
ms = enc.last_timing_ms(); ms.update(enc.mf_timing_ms())
print(json.dumps({'hooks': {k: v for k, v in os.environ.items() if k.startswith('GC_')}, 'corpus': kind, 'compressed': int(c), 'total_ms': round(ms['total'], 2), 'mf.dp': round(ms.get('mf.dp', 0), 2)}))
