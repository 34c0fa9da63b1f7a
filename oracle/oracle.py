"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY: ctypes access to the checker libraries.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product path under 7-zip-zstd_amd/ never does).

  port : oracle/libgc_oracle.so        -- this repo's plain-C restatement (zstd_frame_dec.c)
  ref  : oracle/_ref/lib*_ref.so       -- the reference's own C codecs compiled from /root/reference by oracle/Makefile
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_SZ = C.c_size_t
_VP = C.c_void_p


def build(verbose=False):
    """Compile the restatement and, if /root/reference is present, the reference libraries."""
    r = subprocess.run(["make", "-C", HERE, "-j8", "all"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(r.stdout)


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


_port = None
_ref = {}


def port():
    global _port
    if _port is None:
        p = os.path.join(HERE, "libgc_oracle.so")
        if not os.path.exists(p):
            build()
        _port = C.CDLL(p)
        _port.gco_zstd_decompress.argtypes = [_VP, _SZ, _VP, _SZ, C.POINTER(_SZ)]
        _port.gco_zstd_decompress.restype = C.c_int
        _port.gco_last_diag.restype = C.POINTER(_Diag)
        _port.gco_xxh64.argtypes = [_VP, _SZ, C.c_uint64]
        _port.gco_xxh64.restype = C.c_uint64
        _port.gco_lzma2_decode.argtypes = [_VP, _SZ, _VP, _SZ, C.c_ubyte]
        _port.gco_lzma2_decode.restype = _SZ
    return _port


class _Diag(C.Structure):
    _fields_ = [("code", C.c_int), ("line", C.c_int), ("src_pos", _SZ), ("dst_pos", _SZ),
                ("frames", C.c_uint), ("blocks", C.c_uint)]


def ref(name):
    """name in {'zstd','flzma2','brotli','bra','lzfind'}; returns CDLL or None when the prebuilt .so is absent."""
    if name not in _ref:
        lib = _load(os.path.join(HERE, "_ref", "lib%s_ref.so" % name))
        if lib is not None:
            if name == "zstd":
                lib.ref_zstd_compress.argtypes = [_VP, _SZ, _VP, _SZ, C.c_int, C.c_int]
                lib.ref_zstd_compress_pieces.argtypes = [_VP, _SZ, _VP, _SZ, C.c_int, _SZ]
                lib.ref_zstd_decompress.argtypes = [_VP, _SZ, _VP, _SZ]
                lib.ref_zstd_compress_bound.argtypes = [_SZ]
                lib.ref_zstd_compress_sequences.argtypes = [_VP, _SZ, _VP, _VP, _VP, _SZ, _VP, _SZ, C.c_int]
                for f in ("ref_zstd_compress", "ref_zstd_compress_pieces", "ref_zstd_decompress",
                          "ref_zstd_compress_bound", "ref_zstd_compress_sequences"):
                    getattr(lib, f).restype = _SZ
            elif name == "flzma2":
                lib.ref_fl2_compress.argtypes = [_VP, _SZ, _VP, _SZ, C.c_int, C.c_uint, C.POINTER(C.c_ubyte)]
                lib.ref_fl2_compress.restype = _SZ
                lib.ref_lzma2_decode.argtypes = [_VP, _SZ, _VP, _SZ, C.c_ubyte]
                lib.ref_lzma2_decode.restype = _SZ
            elif name == "bra":
                lib.ref_bra_convert.argtypes = [C.c_int, _VP, _SZ, C.c_uint, C.c_int]
                lib.ref_bra_convert.restype = _SZ
                lib.ref_bra_x86_convert.argtypes = [_VP, _SZ, C.c_uint, C.c_int, C.POINTER(C.c_uint)]
                lib.ref_bra_x86_convert.restype = _SZ
                lib.ref_delta_convert.argtypes = [_VP, _SZ, C.c_uint, C.c_int, _VP]
                lib.ref_delta_convert.restype = None
                if hasattr(lib, "ref_crc32"):
                    lib.ref_crc32.argtypes = [_VP, _SZ]; lib.ref_crc32.restype = C.c_uint
            elif name == "lzfind":
                lib.ref_lzfind_matches.argtypes = [_VP, _SZ, C.c_uint, C.c_int, C.c_int, C.c_uint, C.c_uint, _VP, _VP, _SZ, C.POINTER(_SZ)]
                lib.ref_lzfind_matches.restype = C.c_int
            elif name == "brotli":
                lib.ref_brotli_compress.argtypes = [_VP, _SZ, _VP, _SZ, C.c_int, C.c_int]
                lib.ref_brotli_decompress.argtypes = [_VP, _SZ, _VP, _SZ]
                lib.ref_brotlimt_compress.argtypes = [_VP, _SZ, _VP, _SZ, C.c_int, C.c_int]
                lib.ref_brotlimt_decompress.argtypes = [_VP, _SZ, _VP, _SZ, C.c_int]
                for f in ("ref_brotli_compress", "ref_brotli_decompress", "ref_brotlimt_compress",
                          "ref_brotlimt_decompress"):
                    getattr(lib, f).restype = _SZ
        _ref[name] = lib
    return _ref[name]


_BAD = (1 << 64) - 1


def _buf(x):
    a = np.ascontiguousarray(np.frombuffer(x, dtype=np.uint8) if not isinstance(x, np.ndarray) else x, dtype=np.uint8)
    return a, a.ctypes.data


# ----------------------------------------------------------------------------- corpora
def corpus(kind, n, seed=20260921):
    """Synthetic corpora live with the harness (7-zip-zstd_amd/corpus); re-exported here for the tests."""
    import importlib.util
    import sys
    name = "sevenzip_zstd_amd_corpus"
    if name not in sys.modules:
        spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(HERE), "7-zip-zstd_amd", "corpus", "__init__.py"))
        mod = importlib.util.module_from_spec(spec); sys.modules[name] = mod; spec.loader.exec_module(mod)
    mod = sys.modules[name]
    if kind in mod.REAL_KINDS:                       # real bytes from the image (round 3): real-src / real-bin / real-py
        return mod.real_corpus(kind, n)
    return mod.corpus(kind, n, seed)


# ----------------------------------------------------------------------------- zstd
def port_zstd_decompress(comp, cap):
    a, ap = _buf(comp)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    n = _SZ(0)
    rc = port().gco_zstd_decompress(out.ctypes.data, cap, ap, a.size, C.byref(n))
    if rc != 0:
        d = port().gco_last_diag().contents
        raise ValueError("port zstd decode failed: code=%d line=%d src_pos=%d dst_pos=%d frames=%d blocks=%d"
                         % (d.code, d.line, d.src_pos, d.dst_pos, d.frames, d.blocks))
    return out[:n.value]


def ref_zstd_decompress(comp, cap):
    a, ap = _buf(comp)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    r = ref("zstd").ref_zstd_decompress(out.ctypes.data, cap, ap, a.size)
    if r == _BAD:
        raise ValueError("reference zstd decoder rejected the stream")
    return out[:r]


def ref_zstd_compress(data, level=3, workers=0, piece=0):
    a, ap = _buf(data)
    lib = ref("zstd")
    npieces = 1 if not piece else (a.size + piece - 1) // piece + 1
    cap = lib.ref_zstd_compress_bound(a.size) + 64 * npieces + 64
    out = np.empty(cap, dtype=np.uint8)
    if piece:
        r = lib.ref_zstd_compress_pieces(out.ctypes.data, cap, ap, a.size, level, piece)
    else:
        r = lib.ref_zstd_compress(out.ctypes.data, cap, ap, a.size, level, workers)
    if r == _BAD:
        raise RuntimeError("reference zstd compress failed")
    return out[:r].copy()


def ref_bra_convert(kind, data, pc=0, encoding=True):
    """The reference's branch converter (C/Bra.c) on a copy of data, one call: (converted array, processed bytes).  kind 0 ARM64 1 ARM 2 ARMT 3 PPC 4 SPARC"""
    a = np.array(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8, copy=True)
    buf = np.zeros(a.size + 64, dtype=np.uint8)          # 16-byte aligned start, room behind the end
    off = (-buf.ctypes.data) % 16
    view = buf[off:off + a.size]; view[:] = a
    done = ref("bra").ref_bra_convert(kind, view.ctypes.data, a.size, pc & 0xFFFFFFFF, 1 if encoding else 0)
    return view.copy(), int(done)


def ref_crc32(data):
    """The reference's CRC-32 (C/7zCrc.c CrcCalc) of data"""
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    return int(ref("bra").ref_crc32(a.ctypes.data if a.size else None, a.size)) & 0xFFFFFFFF


def ref_delta_convert(data, delta, encoding=True, state=None):
    """The reference's Delta filter (C/Delta.c) on a copy: (converted array, state out as bytes)"""
    a = np.array(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8, copy=True)
    st = np.frombuffer(bytes(state) if state is not None else bytes(256), dtype=np.uint8).copy()
    ref("bra").ref_delta_convert(a.ctypes.data if a.size else None, a.size, int(delta), 1 if encoding else 0, st.ctypes.data)
    return a, st.tobytes()


def ref_bra_x86_convert(data, pc=0, encoding=True, state=0):
    """The reference's X86 converter (C/Bra86.c), one call on a copy: (converted array, processed bytes, state out)"""
    a = np.array(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8, copy=True)
    st = C.c_uint(state)
    done = ref("bra").ref_bra_x86_convert(a.ctypes.data if a.size else None, a.size, pc & 0xFFFFFFFF, 1 if encoding else 0, C.byref(st))
    return a, int(done), int(st.value)


def ref_zstd_compress_opts(data, level=3, checksum=False, streamed=False, ldm=False):
    """One frame from the reference encoder with a content checksum / without a content size field / with long-distance matching."""
    a, ap = _buf(data)
    lib = ref("zstd")
    cap = lib.ref_zstd_compress_bound(a.size) + 1024
    out = np.empty(cap, dtype=np.uint8)
    lib.ref_zstd_compress_opts.restype = C.c_size_t
    lib.ref_zstd_compress_opts.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint]
    r = lib.ref_zstd_compress_opts(out.ctypes.data, cap, ap, a.size, level, (1 if checksum else 0) | (2 if streamed else 0) | (4 if ldm else 0))
    if r == _BAD:
        raise RuntimeError("reference zstd compress failed")
    return out[:r].copy()


def ref_zstd_compress_sequences(data, offs, lls, mls, level=3):
    a, ap = _buf(data)
    lib = ref("zstd")
    cap = lib.ref_zstd_compress_bound(a.size) + 1024
    out = np.empty(cap, dtype=np.uint8)
    o = np.ascontiguousarray(offs, dtype=np.uint32); l = np.ascontiguousarray(lls, dtype=np.uint32)
    m = np.ascontiguousarray(mls, dtype=np.uint32)
    r = lib.ref_zstd_compress_sequences(out.ctypes.data, cap, o.ctypes.data, l.ctypes.data, m.ctypes.data,
                                        o.size, ap, a.size, level)
    if r == _BAD:
        raise RuntimeError("reference ZSTD_compressSequences rejected the sequences")
    return out[:r].copy()


# ----------------------------------------------------------------------------- flzma2 / brotli
def ref_fl2_compress(data, level=5, threads=1):
    a, ap = _buf(data)
    cap = a.size + a.size // 8 + 4096
    out = np.empty(cap, dtype=np.uint8)
    prop = C.c_ubyte(0)
    r = ref("flzma2").ref_fl2_compress(out.ctypes.data, cap, ap, a.size, level, threads, C.byref(prop))
    if r == _BAD:
        raise RuntimeError("reference FL2 compress failed")
    return out[:r].copy(), prop.value


def port_lzma2_decode(comp, cap, prop):
    """oracle/lzma2_dec.c: the plain-C restatement of the LZMA2 decoder."""
    a, ap = _buf(comp)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    r = port().gco_lzma2_decode(ap, a.size, out.ctypes.data, cap, prop)
    if r > (1 << 63):
        raise ValueError("port LZMA2 decoder rejected the stream (check %d)" % ((1 << 64) - 1 - r))
    return out[:r]


def ref_lzma2_decode(comp, cap, prop):
    a, ap = _buf(comp)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    r = ref("flzma2").ref_lzma2_decode(out.ctypes.data, cap, ap, a.size, prop)
    if r >= _BAD - 1:
        raise ValueError("reference LZMA2 decoder rejected the stream (%d)" % (r - (1 << 64)))
    return out[:r]


def ref_brotlimt_compress(data, level=6, threads=1):
    a, ap = _buf(data)
    cap = a.size + a.size // 4 + 4096
    out = np.empty(cap, dtype=np.uint8)
    r = ref("brotli").ref_brotlimt_compress(out.ctypes.data, cap, ap, a.size, level, threads)
    if r == _BAD:
        raise RuntimeError("reference brotli-mt compress failed")
    return out[:r].copy()


def ref_brotlimt_decompress(comp, cap, threads=1):
    a, ap = _buf(comp)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    r = ref("brotli").ref_brotlimt_decompress(out.ctypes.data, cap, ap, a.size, threads)
    if r == _BAD:
        raise ValueError("reference brotli-mt decoder rejected the stream")
    return out[:r]


def ref_brotli_decompress(comp, cap):
    a, ap = _buf(comp)
    out = np.empty(max(cap, 1), dtype=np.uint8)
    r = ref("brotli").ref_brotli_decompress(out.ctypes.data, cap, ap, a.size)
    if r == _BAD:
        raise ValueError("reference brotli decoder rejected the stream")
    return out[:r]


def ref_brotli_compress(data, level=6, lgwin=22):
    """one bare RFC 7932 stream of the reference encoder (BrotliEncoderCompress)"""
    a, ap = _buf(data)
    cap = a.size + a.size // 4 + 4096
    out = np.empty(cap, dtype=np.uint8)
    r = ref("brotli").ref_brotli_compress(out.ctypes.data, cap, ap, a.size, level, lgwin)
    if r == _BAD:
        raise RuntimeError("reference brotli compress failed")
    return out[:r].copy()


def ref_brotli_dictionary():
    """the static dictionary of RFC 7932 Appendix A as the reference holds it (C/brotli/common/dictionary.h: BrotliGetDictionary()->data, 122 784 bytes)"""
    class _D(C.Structure):
        _fields_ = [("size_bits_by_length", C.c_uint8 * 32), ("offsets_by_length", C.c_uint32 * 32), ("data_size", C.c_size_t), ("data", C.POINTER(C.c_uint8))]
    lib = ref("brotli")
    lib.BrotliGetDictionary.restype = C.POINTER(_D)
    d = lib.BrotliGetDictionary().contents
    return np.frombuffer(C.string_at(d.data, d.data_size), dtype=np.uint8).copy()


def _match_lists(fn, data, history, cut, nice, extra):
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
    counts = np.zeros(max(a.size, 1), dtype=np.uint32)
    cap = 2 * a.size * 8 + 64
    pairs = np.zeros(cap, dtype=np.uint32)
    used = _SZ(0)
    rc = fn(a.ctypes.data if a.size else None, a.size, int(history), *extra, int(cut), int(nice), counts.ctypes.data, pairs.ctypes.data, cap, C.byref(used))
    if rc != 0:
        raise RuntimeError("match finder oracle failed: %d" % rc)
    return counts[:a.size].copy(), pairs[:used.value].copy()


def ref_lzfind_matches(data, history=1 << 20, bt=False, hash_bytes=4, cut=32, nice=64):
    """The reference's mainline match finder (C/LzFind.c through ref_shim_lzfind.c): (values per position, all (length, distance - 1) values in order)."""
    return _match_lists(ref("lzfind").ref_lzfind_matches, data, history, cut, nice, (1 if bt else 0, int(hash_bytes)))


def port_hc4_matches(data, history=1 << 20, cut=32, nice=64):
    """The restatement oracle/lzfind_hc4.c of Hc4_MatchFinder_GetMatches, same output layout."""
    lib = port()
    lib.gc_oracle_hc4_matches.argtypes = [_VP, _SZ, C.c_uint, C.c_uint, C.c_uint, _VP, _VP, _SZ, C.POINTER(_SZ)]
    lib.gc_oracle_hc4_matches.restype = C.c_int
    return _match_lists(lib.gc_oracle_hc4_matches, data, history, cut, nice, ())
