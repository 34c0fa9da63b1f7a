"""Deterministic stand-in corpora for the benchmark inputs that are not available offline (enwik8/enwik9,
Silesia, web text).  Harness utility shared by bench.py and tests/ -- no codec logic.
The generators are plain C (corpus_gen.c, built on demand with gcc into libgc_corpus.so)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libgc_corpus.so")
        src = os.path.join(HERE, "corpus_gen.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src, "-lm"], check=True)
        _lib = C.CDLL(so)
        for name in ("gc_corpus_text_zipf", "gc_corpus_webtext", "gc_corpus_silesia_like"):
            getattr(_lib, name).argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        _lib.gc_corpus_lz7zip.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint32]
    return _lib


KINDS = ("text-zipf", "web-text", "lz-7zip", "silesia-like", "random", "zeros")


def corpus(kind, n, seed=20260921):
    """n bytes of the named synthetic corpus as a numpy uint8 array (pure function of (kind, n, seed))."""
    out = np.empty(n, dtype=np.uint8)
    lib = _load()
    if kind == "text-zipf":
        lib.gc_corpus_text_zipf(out.ctypes.data, n, seed)
    elif kind == "web-text":
        lib.gc_corpus_webtext(out.ctypes.data, n, seed)
    elif kind == "lz-7zip":
        lib.gc_corpus_lz7zip(out.ctypes.data, n, 24, 0 if seed == 20260921 else seed & 0xFFFFFFFF)
    elif kind == "silesia-like":
        lib.gc_corpus_silesia_like(out.ctypes.data, n, seed)
    elif kind == "random":
        out[:] = np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8)
    elif kind == "zeros":
        out[:] = 0
    else:
        raise ValueError(kind)
    return out
