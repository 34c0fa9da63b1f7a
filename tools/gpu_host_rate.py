#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry points (what the plugin's Code() uses): bytes in host memory -> compressed bytes in host memory.
  single  gc_*_compress_host on one context: H2D, kernels, D2H one after the other
  multi   gc_multi_compress_host: 64 MiB pieces over two contexts per GPU (copies of one piece overlap the kernels of another),
          from pageable (numpy) and from pinned (gc_host_alloc) buffers
usage: python tools/gpu_host_rate.py [--bytes N]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
import oracle as O
ap = argparse.ArgumentParser(); ap.add_argument("--bytes", type=int, default=1_000_000_000); a = ap.parse_args()
pkg = g.load_package(); lib = pkg.load_library()
for codec, level, kind in (("zstd", 3, "text-zipf"), ("flzma2", 5, "silesia-like"), ("brotli", 6, "web-text")):
    n = a.bytes if codec != "flzma2" else min(a.bytes, 211_900_000)
    x = O.corpus(kind, n)
    res = {"codec": codec, "level": level, "bytes": n}
    enc = {"zstd": pkg.ZstdEncoder, "flzma2": pkg.Flzma2Encoder, "brotli": pkg.BrotliEncoder}[codec](level=level)
    enc.code(x[: 64 << 20])
    t0 = time.perf_counter(); c = enc.code(x); res["single_ctx_GBps"] = round(n / (time.perf_counter() - t0) / 1e9, 2); enc.close()
    m = pkg.MultiEncoder(codec, level)
    m.code(x[: 256 << 20])
    t0 = time.perf_counter(); c2 = m.code(x); res["multi_pageable_GBps"] = round(n / (time.perf_counter() - t0) / 1e9, 2)
    # pinned source and destination
    cid = pkg.CODEC_IDS[codec]; cap = lib.gc_codec_compress_bound(cid, n) + 1
    src = lib.gc_host_alloc(n); dst = lib.gc_host_alloc(cap)
    C.memmove(src, x.ctypes.data, n)
    out = C.c_size_t(0)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter()
        rc = lib.gc_multi_compress_host(m._m, cid, src, n, dst, cap, level, 0, 0, C.byref(out))
        dt = time.perf_counter() - t0
        assert rc == 0
        best = max(best, n / dt / 1e9)
    res["multi_pinned_GBps"] = round(best, 2); res["workers"] = m.workers(); res["compressed"] = out.value
    y = np.ctypeslib.as_array((C.c_ubyte * out.value).from_address(dst)).copy()
    res["pinned_equals_pageable"] = bool(np.array_equal(y, c2))
    lib.gc_host_free(src); lib.gc_host_free(dst); m.close()
    print(json.dumps(res), flush=True)
