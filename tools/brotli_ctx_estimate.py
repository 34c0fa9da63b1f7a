#!/usr/bin/env python3
"""RESEARCH TOOL: what would literal context modelling (RFC 7932 section 7.1, MSB6 mode: context = previous byte >> 2) buy B1?
Entropy of the bytes of a 128 KiB block under T prefix codes -- the T - 1 most frequent contexts get a tree of their own, the rest share
one -- relative to a single code.  Round 2, synthetic corpora: T = 4 -> 0.96-0.97, T = 8 -> 0.92-0.93 of the literal bits, i.e. 1.5-4 % of
the meta-block, against 0.3-0.8 KB of extra tree descriptions per 33 KB meta-block (1-2.4 %): no net gain at 128 KiB meta-blocks, which is
why B1 keeps one literal tree (the reference's DecideOverLiteralContextModeling, br_encode.c:410, makes the same kind of estimate).
usage: python tools/brotli_ctx_estimate.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
def H(c):
    c = c[c > 0].astype(float); n = c.sum(); return -(c * np.log2(c / n)).sum()
for kind in ("web-text", "text-zipf", "lz-7zip", "silesia-like"):
    x = O.corpus(kind, 4 << 20)
    for blk in (0, 10):
        s = x[blk * 131072:(blk + 1) * 131072]; cur = s[1:].astype(int); ctx = s[:-1].astype(int) >> 2
        base = H(np.bincount(cur, minlength=256)); order = np.argsort(-np.bincount(ctx, minlength=64)); res = []
        for T in (2, 4, 8, 16):
            cmap = np.full(64, T - 1)
            for i, c in enumerate(order[:T - 1]): cmap[c] = i
            tree = cmap[ctx]
            res.append((T, round(sum(H(np.bincount(cur[tree == t], minlength=256)) for t in range(T)) / base, 3)))
        print(kind, blk, "bits/byte", round(base / len(cur), 2), res)
