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
            tmp = "%s.%d.tmp" % (so, os.getpid())            # several test workers may get here at once: build aside, rename into place
            subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", tmp, src, "-lm"], check=True)
            os.replace(tmp, so)
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


# ------------------------------------------------------------------------------------------------------------------------------------
# REAL data (round 3).  The generators above are stand-ins; the size bars measured on them say how the encoders do on one generator.  The
# image itself -- the same on the build container and on the GPU box -- holds real bytes of the kinds the benchmark corpora consist of:
#   real-src   C / C++ headers of /opt/rocm/include, then the .py sources of the site-packages (source text, as enwik / a source tarball)
#   real-bin   the shared objects of /opt/rocm/lib: x86-64 code + gfx code objects + symbol and string tables (up to 16 MiB of each
#              file, so that no single library is the corpus), as Silesia's mozilla / ooffice / samba
#   real-py    the Python standard library, .py and .pyc (text + marshalled byte code: a mixed archive)
# Files in sorted order, concatenated, cut at n bytes: a pure function of the image.  Nothing of /root/reference is read.
REAL_KINDS = ("real-src", "real-bin", "real-py")
_REAL_ROOTS = {
    "real-src": [("/opt/rocm/include", (".h", ".hpp", ".inc", ".cuh")), ("/usr/local/lib/python3.10/dist-packages", (".py",)), ("/usr/lib/python3/dist-packages", (".py",))],
    "real-bin": [("/opt/rocm/lib", (".so",))],
    "real-py": [("/usr/lib/python3.10", (".py", ".pyc"))],
}
_real_cache = {}


def real_corpus(kind, n):
    """up to n bytes of real data from the image (numpy uint8; shorter if the image holds less); see above"""
    if kind not in _REAL_ROOTS:
        raise ValueError("unknown real corpus %r" % (kind,))
    have = _real_cache.get(kind)
    if have is not None and have.size >= n:
        return have[:n].copy()
    per_file = (16 << 20) if kind == "real-bin" else (1 << 40)
    parts, total = [], 0
    for root, exts in _REAL_ROOTS[kind]:
        root = os.path.realpath(root)
        if total >= n or not os.path.isdir(root):
            continue
        for d, dirs, files in os.walk(root):
            dirs.sort()
            if kind == "real-bin":
                dirs[:] = []                                  # the top level only: the libraries themselves, not their kernel data bases
            for f in sorted(files):
                if total >= n:
                    break
                p = os.path.join(d, f)
                if os.path.islink(p) or not os.path.isfile(p):
                    continue
                if not (f.endswith(exts) or (kind == "real-bin" and ".so." in f)):
                    continue
                try:
                    with open(p, "rb") as fh:
                        b = fh.read(min(per_file, n - total))
                except OSError:
                    continue
                parts.append(np.frombuffer(b, dtype=np.uint8)); total += len(b)
            if total >= n:
                break
    out = np.concatenate(parts) if parts else np.empty(0, dtype=np.uint8)
    _real_cache[kind] = out
    return out[:n].copy()
