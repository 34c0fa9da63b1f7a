#!/usr/bin/env python3
"""Reference-encoder sizes (oracle/_ref) for (codec, level, corpus, bytes), cached in tools/ref_sizes_cache.json -- computed on the build container so that a GPU visit
only has to run THIS engine (tools/gpu_sizes.py prints ours / ref from the cache).  The corpora are deterministic (generators with a fixed seed; real-* = files of the image).
usage: python tools/ref_sizes.py codec:levels:corpora:MiB [...]      e.g.  zstd:16,19,22:text-zipf,lz-7zip:32   flzma2:5:real-bin:211900000B"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
CACHE = os.path.join(ROOT, "tools", "ref_sizes_cache.json")

def load():
    return json.load(open(CACHE)) if os.path.exists(CACHE) else {}

def key(codec, level, kind, n): return "%s:%d:%s:%d" % (codec, level, kind, n)

def nbytes(s): return int(s[:-1]) if s.endswith("B") else int(float(s) * 1024 * 1024)

def parse(specs):
    out = []
    for s in specs:
        codec, levels, corpora, size = s.split(":")[:4]
        for kind in corpora.split(","):
            for lv in levels.split(","):
                out.append((codec, int(lv), kind, nbytes(size)))
    return out

if __name__ == "__main__":
    import oracle as O
    thr = 64          # what the GPU box's tests use (min(cpu_count, 64) = 64 there): FL2 slices its dictionary blocks by thread, so the size depends on it
    cache = load()
    for codec, level, kind, n in parse(sys.argv[1:]):
        k = key(codec, level, kind, n)
        if k in cache: print(k, cache[k]); continue
        x = O.corpus(kind, n); t0 = time.time()
        if codec == "zstd": r = len(O.ref_zstd_compress(x, level, workers=thr if level >= 16 else 0))
        elif codec == "flzma2": r = len(O.ref_fl2_compress(x, level, threads=thr)[0])
        else: r = len(O.ref_brotlimt_compress(x, level, thr))
        cache[k] = r; json.dump(cache, open(CACHE, "w"), indent=0, sort_keys=True)
        print(k, r, "%.1fs" % (time.time() - t0), flush=True)
