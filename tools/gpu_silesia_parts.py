#!/usr/bin/env python3
"""FLZMA2 level 5 on the five components of the Silesia stand-in, each compressed on its own by the GPU path and by the reference."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
import oracle as O
pkg = g.load_package()
n = 211_900_000
x = O.corpus("silesia-like", n)
kinds = [0,1,0,2,1,0,3,0,1,2, 0,1,0,4,1,0,3,2,1,0]     # corpus_gen.c gc_corpus_silesia_like: 20 segments
names = ["text", "lz-7zip", "pcm", "opcodes", "random"]
seg = n // 20
tot = {k: [0, 0, 0] for k in range(5)}
e = pkg.Flzma2Encoder(level=5)
for i, k in enumerate(kinds):
    a = i * seg; b = n if i == 19 else a + seg
    s_ = x[a:b]
    c = e.code(s_); r, _ = O.ref_fl2_compress(s_, 5, threads=64)
    tot[k][0] += b - a; tot[k][1] += len(c); tot[k][2] += len(r)
e.close()
for k in range(5):
    print(json.dumps({"kind": names[k], "bytes": tot[k][0], "ours": tot[k][1], "ref": tot[k][2], "ours_over_ref": round(tot[k][1] / tot[k][2], 4),
                      "excess_share_of_total": round((tot[k][1] - tot[k][2]) / sum(v[2] for v in tot.values()), 5)}), flush=True)
