"""Generates tests/golden/bt4_matches.npz from the REFERENCE's match finder (C/LzFind.c compiled into oracle/_ref/liblzfind_ref.so by oracle/Makefile):
what Bt4_MatchFinder_GetMatches returns for every position of three small inputs (the binary-tree counterpart of make_hc4_fixture.py).  Run in the
container that has /root/reference; the fixture pins the device kernels of csrc/gc_lzfind.hip wherever the compiled reference is absent.
usage: python tests/golden/make_bt4_fixture.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
out = {}
for name, kind, n, hist, cut, nice in (("text", "text-zipf", 6000, 1 << 16, 32, 64), ("lz", "lz-7zip", 6000, 2048, 8, 273), ("sil", "silesia-like", 6000, 1 << 20, 4, 32)):
    x = O.corpus(kind, 1 << 16)[:n]
    counts, pairs = O.ref_lzfind_matches(x, hist, True, 4, cut, nice)
    out[name + "_input"] = x; out[name + "_params"] = np.array([hist, cut, nice], dtype=np.uint32)
    out[name + "_counts"] = counts.astype(np.uint16); out[name + "_pairs"] = pairs
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bt4_matches.npz"), **out)
print({k: v.shape for k, v in out.items()})
