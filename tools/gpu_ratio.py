#!/usr/bin/env python3
"""Compressed size of the three GPU codecs next to the reference codecs (oracle/_ref) on the same bytes, + decode check.
usage: python tools/gpu_ratio.py [--bytes N] [--codecs zstd,flzma2,brotli] [--corpora a,b]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=64 * 1024 * 1024)
ap.add_argument("--codecs", default="zstd,flzma2,brotli")
ap.add_argument("--corpora", default="silesia-like,text-zipf,lz-7zip,web-text")
ap.add_argument("--levels", default="")
ap.add_argument("--lib", default="")
a = ap.parse_args()
pkg = g.load_package()
kw = {"lib_path": a.lib} if a.lib else {"device": 0}
thr = min(os.cpu_count() or 1, 64)
for kind in a.corpora.split(","):
    x = O.corpus(kind, a.bytes)
    for codec in a.codecs.split(","):
        lv = [int(v) for v in a.levels.split(",")] if a.levels else [{"zstd": 3, "flzma2": 5, "brotli": 6}[codec]]
        for level in lv:
            t0 = time.time()
            if codec == "zstd":
                e = pkg.ZstdEncoder(level=level, **kw); c = e.code(x); e.close()
                ok = np.array_equal(O.ref_zstd_decompress(c, x.size), x); r = O.ref_zstd_compress(x, level)
            elif codec == "flzma2":
                e = pkg.Flzma2Encoder(level=level, **kw); c = e.code(x); prop = e.coder_props()[0]; e.close()
                ok = np.array_equal(O.ref_lzma2_decode(c, x.size, prop), x); r, _ = O.ref_fl2_compress(x, level, threads=thr)
            else:
                e = pkg.BrotliEncoder(level=level, **kw); c = e.code(x); e.close()
                ok = np.array_equal(O.ref_brotlimt_decompress(c, x.size, thr), x); r = O.ref_brotlimt_compress(x, level, thr)
            print(json.dumps({"corpus": kind, "bytes": a.bytes, "codec": codec, "level": level, "ours": len(c), "ref": len(r),
                              "ours_over_ref": round(len(c) / len(r), 4), "ratio": round(x.size / len(c), 4), "decodes": bool(ok),
                              "s": round(time.time() - t0, 1)}), flush=True)
