#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (gc_*_compress_host: pageable host input -> H2D -> kernels -> D2H ->
host output; what the 7-Zip plugin's Code() uses).  Reported in DESIGN.md section 6; never bench.py's `value`.
usage: python tools/gpu_host_rate.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as g
import oracle as O
pkg = g.load_package()
for codec, level, kind, n in (("zstd", 3, "text-zipf", 100_000_000), ("flzma2", 5, "silesia-like", 211_900_000), ("brotli", 6, "web-text", 500_000_000)):
    x = O.corpus(kind, n)
    enc = {"zstd": pkg.ZstdEncoder, "flzma2": pkg.Flzma2Encoder, "brotli": pkg.BrotliEncoder}[codec](device=0, level=level)
    enc.code(x)                                  # warm-up: workspace allocation
    best = None
    for _ in range(3):
        t0 = time.perf_counter(); c = enc.code(x); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    enc.close()
    print(json.dumps({"codec": codec, "level": level, "bytes": n, "host_to_host_MBps": round(n / best / 1e6, 1), "ms": round(best * 1e3, 2), "compressed": len(c)}), flush=True)
