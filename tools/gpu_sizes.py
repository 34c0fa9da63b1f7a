#!/usr/bin/env python3
"""Device sizes of this engine for (codec, level, corpus, bytes) next to the cached reference sizes of tools/ref_sizes.py (no reference encoder runs on the GPU box).
usage: python tools/gpu_sizes.py [--lib hooks] [--decode] codec:levels:corpora:MiB [...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import __graft_entry__ as g
import oracle as O
import ref_sizes as RS
args = sys.argv[1:]
kw = {"device": 0}; decode = False
pkg = g.load_package()
while args and args[0].startswith("--"):
    if args[0] == "--lib": kw = {"device": 0, "lib_path": pkg.HOOKS_LIB_PATH}; args = args[2:] if args[1] != "hooks" and not ":" in args[1] else args[2:]
    elif args[0] == "--decode": decode = True; args = args[1:]
    else: args = args[1:]
cache = RS.load(); thr = min(os.cpu_count() or 1, 64)
cur = None
for codec, level, kind, n in RS.parse(args):
    if cur != (kind, n): x = O.corpus(kind, n); cur = (kind, n)
    t0 = time.time()
    if codec == "zstd": e = pkg.ZstdEncoder(level=level, **kw)
    elif codec == "flzma2": e = pkg.Flzma2Encoder(level=level, **kw)
    else: e = pkg.BrotliEncoder(level=level, **kw)
    c = e.code(x); prop = e.coder_props()[0] if codec == "flzma2" else 0; e.close()
    ok = None
    if decode:
        y = O.ref_zstd_decompress(c, x.size) if codec == "zstd" else (O.ref_lzma2_decode(c, x.size, prop) if codec == "flzma2" else O.ref_brotlimt_decompress(c, x.size, thr))
        ok = bool(np.array_equal(y, x))
    r = cache.get(RS.key(codec, level, kind, n))
    print(json.dumps({"codec": codec, "level": level, "corpus": kind, "bytes": int(x.size), "ours": int(len(c)), "ref": r, "ours_over_ref": round(len(c) / r, 4) if r else None,
                      "decodes": ok, "s": round(time.time() - t0, 2)}), flush=True)
