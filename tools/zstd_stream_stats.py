#!/usr/bin/env python3
"""Where the bytes of a zstd stream go: this engine's stream (emulator build by default, --gpu for the device) next to the reference encoder's at the
same level on the same bytes -- sequences, literal bytes, bytes of the literals / sequences sections (analysis aid; test infrastructure only).
usage: python tools/zstd_stream_stats.py --corpus real-src --bytes 8388608 --level 3 [--skip N] [--gpu]"""
import argparse, ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--corpus", default="real-src"); ap.add_argument("--bytes", type=int, default=8 << 20); ap.add_argument("--skip", type=int, default=0)
ap.add_argument("--level", type=int, default=3); ap.add_argument("--gpu", action="store_true")
a = ap.parse_args()
pkg = g.load_package()
x = O.corpus(a.corpus, a.skip + a.bytes)[a.skip:]
kw = {"device": 0} if a.gpu else {"lib_path": os.path.join(ROOT, "tests", "emu", "_build", "libgpucodec_emu.so")}
e = pkg.ZstdEncoder(level=a.level, **kw); ours = e.code(x); e.close()
ref = O.ref_zstd_compress(x, a.level)
lib = O.port()
lib.gco_zstd_stats.argtypes = [C.POINTER(C.c_ulonglong)]
def stats(c):
    y = O.port_zstd_decompress(c, x.size); assert np.array_equal(y, x)
    s = (C.c_ulonglong * 16)(); lib.gco_zstd_stats(s); return list(s)
names = ["blocks with sequences", "sequences", "literal bytes", "bytes of literals sections", "bytes of sequences sections", "match bytes", "repeat-offset sequences",
         "sum of offset codes", "-", "matches < 8 bytes"]
so, sr = stats(ours), stats(ref)
print("%-30s %12s %12s" % ("", "ours", "reference"))
print("%-30s %12d %12d   %.4f" % ("stream bytes", len(ours), len(ref), len(ours) / len(ref)))
for i, nm in enumerate(names):
    if nm != "-": print("%-30s %12d %12d" % (nm, so[i], sr[i]))
print("%-30s %12.2f %12.2f" % ("bits per sequence", 8.0 * so[4] / max(so[1], 1), 8.0 * sr[4] / max(sr[1], 1)))
print("%-30s %12.2f %12.2f" % ("bits per literal", 8.0 * so[3] / max(so[2], 1), 8.0 * sr[3] / max(sr[2], 1)))
print("%-30s %12.2f %12.2f" % ("mean match length", so[5] / max(so[1], 1), sr[5] / max(sr[1], 1)))
