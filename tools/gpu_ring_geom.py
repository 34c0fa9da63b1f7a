#!/usr/bin/env python3
"""W6r (ring-aware parse, brotli qualities 5-7) geometries on the device through the hooks library: parse time on 1 GB / 32 MiB of web-text, sizes on real data.
usage: python tools/gpu_ring_geom.py [T:warm16:quiet ...]     (T threads per block = T / 16 sub-blocks; warm16 = warm-up positions / 16; quiet = single steps behind a copy before literal runs are skipped)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import __graft_entry__ as g
import oracle as O
import ref_sizes as RS
pkg = g.load_package()
cache = RS.load()
geoms = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(128, 16, 4), (256, 16, 4)]
dev = torch.device("cuda", 0)
big = torch.from_numpy(O.corpus("web-text", 1000000000)).to(dev)
small = big[:33554432].clone()
reals = {k: O.corpus(k, 64 << 20) for k in ("real-src", "real-bin")}
def rate(x):
    e = pkg.BrotliEncoder(level=6, device=0, lib_path=pkg.HOOKS_LIB_PATH)
    cap = e.compress_bound(x.numel()) + 16; d = torch.empty(cap, dtype=torch.uint8, device=dev)
    ms = []; pm = []
    for i in range(4):
        e.code_device(x.data_ptr(), x.numel(), d.data_ptr(), cap); e.finish()
        if i: ms.append(e.last_timing_ms()["total"]); pm.append(e.mf_timing_ms()["mf.parse"])
    e.close()
    return round(sum(ms) / len(ms), 3), round(sum(pm) / len(pm), 3)
for T, warm, quiet in [(0, 0, 0)] + geoms:
    if T: os.environ["GC_BR_RING_GEOM"] = str(T); os.environ["GC_BR_RING"] = str(2 | (quiet << 16) | (warm << 24))
    else: os.environ["GC_BR_RING"] = "0"
    row = {"threads": T, "warm": warm * 16, "quiet": quiet}
    row["1GB total / parse ms"] = rate(big); row["32MiB total / parse ms"] = rate(small)
    for k, x in reals.items():
        e = pkg.BrotliEncoder(level=6, device=0, lib_path=pkg.HOOKS_LIB_PATH); c = e.code(x); e.close()
        assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 64), x), (k, T)
        row[k] = round(len(c) / cache[RS.key("brotli", 6, k, x.size)], 4)
    print(json.dumps(row), flush=True)
