#!/usr/bin/env python3
"""Kernel times of zstd level 3 and Fast-LZMA2 level 5 on REAL bytes tiled to the metric's sizes (whole 8 MiB frames repeated), next to the stand-in corpora.
usage: python tools/gpu_real_rate.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import __graft_entry__ as g
import oracle as O
pkg = g.load_package()
def tiled(kind, n):
    full = O.corpus(kind, 211_900_000)
    if not kind.startswith('real'): return O.corpus(kind, n)
    return np.ascontiguousarray(np.resize(full[: full.size - full.size % (8 << 20)], n))
for codec, level, n, kinds in (('zstd', 3, 1_000_000_000, ('text-zipf', 'real-src', 'real-bin')), ('flzma2', 5, 211_900_000, ('silesia-like', 'real-src', 'real-bin'))):
    for kind in kinds:
        x = tiled(kind, n)
        enc = (pkg.ZstdEncoder if codec == 'zstd' else pkg.Flzma2Encoder)(level=level, device=0)
        d_src = torch.from_numpy(x).to('cuda:0'); cap = enc.compress_bound(x.size); d_dst = torch.empty(cap, dtype=torch.uint8, device='cuda:0')
        for _ in range(2):
            enc.code_device(d_src.data_ptr(), x.size, d_dst.data_ptr(), cap); c = enc.finish()
        ms = enc.last_timing_ms()
        try: ms.update(enc.mf_timing_ms())
        except Exception as e: ms['mf'] = repr(e)
        print(json.dumps({'codec': codec, 'level': level, 'corpus': kind, 'bytes': int(x.size), 'compressed': int(c), 'ratio': round(x.size / c, 3), 'kernel_ms': {k: round(v, 3) if isinstance(v, float) else v for k, v in ms.items()}}), flush=True)
        enc.close(); del d_src, d_dst
