#!/usr/bin/env python3
"""One codec / level / corpus, a few calls on data resident in HBM (for rocprofv3 --kernel-trace --stats around it).  usage: python tools/gpu_one.py codec level corpus bytes [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import __graft_entry__ as g
import oracle as O
pkg = g.load_package()
codec, level, kind, n = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
full = O.corpus(kind, min(n, 211_900_000))
x = np.ascontiguousarray(np.resize(full[: full.size - full.size % (8 << 20)] if full.size >= (8 << 20) else full, n)) if kind.startswith("real") else O.corpus(kind, n)
enc = {"zstd": pkg.ZstdEncoder, "flzma2": pkg.Flzma2Encoder, "brotli": pkg.BrotliEncoder}[codec](level=level, device=0)
d = torch.from_numpy(x).cuda(); cap = enc.compress_bound(x.size); out = torch.empty(cap, dtype=torch.uint8, device="cuda")
for _ in range(reps):
    enc.code_device(d.data_ptr(), x.size, out.data_ptr(), cap); c = enc.finish()
print(codec, level, kind, x.size, c, enc.last_timing_ms())
