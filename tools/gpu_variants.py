#!/usr/bin/env python3
"""A/B of library variants (tools/build_variants.py) and test hooks on one GPU, one process, one corpus.
usage: python tools/gpu_variants.py [--bytes N] [--corpus K] [--codec zstd|flzma2|brotli] [--level L] [--reps R] [--phases] spec...
spec = libname[@ENV=V[,ENV=V...]]   libname `shipped` = csrc/libgpucodec.so, else tools/_variants/lib_<libname>.so
Prints one JSON line per spec: compressed size, mean kernel times (HIP events), optional in-kernel phase cycles."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=1_000_000_000)
ap.add_argument("--corpus", default="text-zipf")
ap.add_argument("--codec", default="zstd")
ap.add_argument("--level", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--phases", action="store_true")
ap.add_argument("--check", action="store_true", help="decode every variant's stream under the reference decoder (zstd only)")
ap.add_argument("specs", nargs="+")
a = ap.parse_args()
g.build_hip()
pkg = g.load_package()
from importlib import util as _u
spec = _u.spec_from_file_location("c", os.path.join(ROOT, "7-zip-zstd_amd", "corpus", "__init__.py"))
cm = _u.module_from_spec(spec); spec.loader.exec_module(cm)
if a.corpus in cm.REAL_KINDS:                          # real bytes tiled to the size asked for (whole 8 MiB frames repeated), as tools/gpu_real_rate.py does
    full = cm.real_corpus(a.corpus, 211_900_000)
    x = np.ascontiguousarray(np.resize(full[: full.size - full.size % (8 << 20)], a.bytes))
else:
    x = cm.corpus(a.corpus, a.bytes)
d_src = torch.from_numpy(x).cuda()
for sp in a.specs:
    name, _, envs = sp.partition("@")
    lib = None if name == "shipped" else os.path.join(ROOT, "tools", "_variants", "lib_%s.so" % name)
    if name == "shipped" and envs: lib = pkg.HOOKS_LIB_PATH        # (the shipped library reads no environment: hooks live in the test build)
    saved = {}
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("="); saved[k] = os.environ.get(k); os.environ[k] = v
    try:
        cls = {"zstd": pkg.ZstdEncoder, "flzma2": pkg.Flzma2Encoder, "brotli": pkg.BrotliEncoder}[a.codec]
        enc = cls(device=0, level=a.level or {"zstd": 3, "flzma2": 5, "brotli": 6}[a.codec], lib_path=lib)
        cap = enc.compress_bound(x.size); d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
        def run():
            enc.code_device(d_src.data_ptr(), x.size, d_dst.data_ptr(), cap); return enc.finish()
        for _ in range(2): size = run()
        acc = {}
        for _ in range(a.reps):
            size = run()
            for k, v in enc.last_timing_ms().items(): acc[k] = acc.get(k, 0) + v / a.reps
            for k, v in (enc.mf_timing_ms() or {}).items(): acc[k] = acc.get(k, 0) + v / a.reps
        out = {"spec": sp, "compressed": int(size), "GBps": round(a.bytes / acc["total"] / 1e6, 2), "ms": {k: round(v, 3) for k, v in acc.items()}}
        if a.phases and a.codec == "zstd":
            enc.set_phase_profile(True); run(); out["phase_cycles_per_block"] = {k: round(v) for k, v in enc.phase_profile().items() if v}; enc.set_phase_profile(False)
        if a.check and a.codec == "zstd":
            sys.path.insert(0, os.path.join(ROOT, "oracle")); import oracle as O
            y = O.ref_zstd_decompress(d_dst[:size].cpu().numpy(), x.size); out["decodes"] = bool(np.array_equal(x, y))
        enc.close(); del d_dst
    except Exception as e:                                    # a variant that fails must not end the run
        out = {"spec": sp, "error": repr(e)}
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    print(json.dumps(out), flush=True)
