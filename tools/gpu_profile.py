#!/usr/bin/env python3
"""Kernel timings (HIP events) + in-kernel phase profile of the zstd path on one GPU.
usage: python tools/gpu_profile.py [--bytes N] [--corpus KIND] [--reps R]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=100_000_000)
ap.add_argument("--corpus", default="text-zipf")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--codec", default="zstd")
ap.add_argument("--level", type=int, default=0)
ap.add_argument("--lib", default=None, help="library variant (tools/_variants/lib_<name>.so)")
a = ap.parse_args()
if not a.lib: g.build_hip()
pkg = g.load_package()
from importlib import util as _u
spec = _u.spec_from_file_location("c", os.path.join(ROOT, "7-zip-zstd_amd", "corpus", "__init__.py"))
cm = _u.module_from_spec(spec); spec.loader.exec_module(cm)
x = cm.real_corpus(a.corpus, a.bytes) if a.corpus in cm.REAL_KINDS else cm.corpus(a.corpus, a.bytes)
if x.size < a.bytes: x = np.resize(x[:(x.size >> 23) << 23], a.bytes)      # real bytes tiled frame by frame to the size asked for
fl2 = a.codec == "flzma2"
br = a.codec == "brotli"
enc = pkg.Flzma2Encoder(device=0, level=a.level or 5, lib_path=a.lib) if fl2 else (pkg.BrotliEncoder(device=0, level=a.level or 6, lib_path=a.lib) if br else pkg.ZstdEncoder(device=0, level=a.level or 3, lib_path=a.lib))
d_src = torch.from_numpy(x).cuda(); cap = enc.compress_bound(x.size); d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
def run():
    enc.code_device(d_src.data_ptr(), x.size, d_dst.data_ptr(), cap); return enc.finish()
for _ in range(2): size = run()
acc = {}
for _ in range(a.reps):
    size = run()
    for k, v in enc.last_timing_ms().items(): acc[k] = acc.get(k, 0) + v / a.reps
    for k, v in (enc.mf_timing_ms() or {}).items(): acc[k] = acc.get(k, 0) + v / a.reps
if fl2:
    enc.set_phase_profile(True)
    run(); ph = enc.phase_profile(); tp = acc
elif br:
    ph, tp = {}, acc
else:
    enc.set_phase_profile(True)
    run(); ph = enc.phase_profile(); tp = enc.last_timing_ms()
    enc.set_phase_profile(False)
print(json.dumps({"codec": a.codec, "level": enc.level, "corpus": a.corpus, "bytes": a.bytes, "compressed": size, "ratio": round(a.bytes / size, 4),
                  "GBps_total": round(a.bytes / acc["total"] / 1e6, 2), "kernel_ms": {k: round(v, 4) for k, v in acc.items()},
                  "kernel_ms_profiled": {k: round(v, 4) for k, v in tp.items()},
                  "phase_cycles_per_block": {k: round(v) for k, v in ph.items()}}))
