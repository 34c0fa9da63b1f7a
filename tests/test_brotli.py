"""BROTLI path (7-Zip method id 0x4F71102): brotli-mt framed streams produced by the HIP kernels must regenerate the input
bit-exactly under the reference's own decoder (C/brotli + C/zstdmt/brotli-mt_decompress.c compiled into oracle/_ref).

CPU tests run the unmodified kernel sources under the SIMT emulator; -m gpu tests run the product library on the MI355X and
additionally require GPU bytes == emulator bytes."""
import os
import struct

import numpy as np
import pytest

BLK = 128 * 1024


def _need_ref(O):
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref/libbrotli_ref.so not built")


@pytest.fixture(scope="module")
def emu_br(pkg, emu_lib_path):
    encs = {lv: pkg.BrotliEncoder(lib_path=emu_lib_path, level=lv) for lv in (1, 6)}
    yield encs
    for e in encs.values():
        e.close()


@pytest.fixture(scope="module")
def gpu_br(pkg, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    encs = {lv: pkg.BrotliEncoder(device=0, level=lv) for lv in (1, 6)}
    yield encs
    for e in encs.values():
        e.close()


def _frames(c):
    """walk the brotli-mt frames: (header fields, brotli payload)"""
    out, p = [], 0
    while p < len(c):
        magic, eight, csize, br, hint = struct.unpack_from("<IIIHH", bytes(c[p:p + 16]))
        assert magic == 0x184D2A50 and eight == 8 and br == 0x5242          # brotli-mt_compress.c:299-321
        out.append((csize, hint, c[p + 16:p + 16 + csize]))
        p += 16 + csize
    assert p == len(c)
    return out


def _roundtrip(O, enc, x):
    _need_ref(O)
    c = enc.code(x)
    assert np.array_equal(O.ref_brotlimt_decompress(c, x.size), x)
    # every frame is a complete brotli stream that the plain (non-mt) reference decoder accepts, and the hint covers it
    pos = 0
    for csize, hint, payload in _frames(c):
        part = O.ref_brotli_decompress(payload, hint << 16)
        assert np.array_equal(part, x[pos:pos + part.size])
        pos += part.size
    assert pos == x.size
    return c


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 64, 255, 1000, 4097, BLK - 1, BLK, BLK + 1])
def test_emu_edge_sizes(O, emu_br, n):
    _roundtrip(O, emu_br[6], O.corpus("text-zipf", n))


@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like", "web-text", "random", "zeros"])
def test_emu_corpora(O, emu_br, kind):
    x = O.corpus(kind, BLK + 70_000)
    c = _roundtrip(O, emu_br[6], x)
    if kind == "random":
        assert len(c) <= x.size + 64                    # stored meta-blocks
    if kind in ("text-zipf", "web-text"):
        ref = O.ref_brotlimt_compress(x, 6, 1)
        assert len(c) <= 1.15 * len(ref), (len(c), len(ref))


def test_emu_patterns_and_degenerate_alphabets(O, emu_br):
    enc = emu_br[6]
    _roundtrip(O, enc, np.tile(np.arange(7, dtype=np.uint8), (BLK + 50) // 7 + 1)[:BLK + 50].copy())
    r = O.corpus("random", 70_000)
    _roundtrip(O, enc, np.concatenate([r, r[:50_000]]))
    rng = np.random.default_rng(7)
    p = 1.0 / np.arange(1, 257) ** 1.2; p /= p.sum()
    _roundtrip(O, enc, rng.choice(256, size=BLK, p=p).astype(np.uint8))          # 256-symbol literal code, deep tree
    _roundtrip(O, enc, np.full(5000, 65, dtype=np.uint8))                          # one literal symbol: NSYM = 1 simple code
    _roundtrip(O, enc, np.frombuffer(b"ab" * 3000, dtype=np.uint8).copy())         # two literal symbols
    q = 1.0 / 2.0 ** np.arange(1, 41); q /= q.sum()
    _roundtrip(O, enc, rng.choice(40, size=60_000, p=q).astype(np.uint8))           # Fibonacci-like counts: 15-bit length limit


def test_emu_multi_chunk_framing(O, emu_br):
    _need_ref(O)
    x = O.corpus("text-zipf", 9 * BLK + 100)             # level 1: 1 MiB chunks = 8 blocks -> 2 frames
    c = _roundtrip(O, emu_br[1], x)
    fr = _frames(c)
    assert len(fr) == 2 and fr[0][1] == (8 * BLK >> 16) + 1


def test_emu_deterministic(O, emu_br):
    x = O.corpus("silesia-like", BLK)
    assert np.array_equal(emu_br[6].code(x), emu_br[6].code(x))


def test_emu_last_distance_substitution(pkg, O, emu_lib_path, monkeypatch):
    """B1 gives a copy the distance of one of the three commands in front when its bytes match there too (same bytes out, same parse, a cheaper distance code).  On
    records of a fixed stride -- where the finder's nearest candidate and the parse's last distance differ all the time -- the stream must get smaller than with the
    step switched off (hook GC_BR_REPSUB=0), on text it must stay where it was (within 0.05 %), and every stream must decode under the reference decoder."""
    _need_ref(O)
    rng = np.random.default_rng(5)
    rec = rng.integers(0, 256, size=(6000, 24), dtype=np.uint8)
    rec[:, :10] = rec[0, :10]; rec[:, 14:20] = (np.arange(6000)[:, None] >> np.array([0, 8, 16, 0, 8, 16])) & 0xFF      # tables of records: constant fields, counters, noise
    cases = (("records", rec.reshape(-1).copy()), ("text", O.corpus("text-zipf", 3 * BLK + 5)), ("objects", O.corpus("real-bin", 4 * BLK)))
    for name, x in cases:
        if x.size < BLK:
            continue                                                                                                        # (the image holds no such data)
        sizes = {}
        monkeypatch.setenv("GC_BR_RING", "0")            # W6's parse (W6r tries the ring distances itself: behind it the step finds nothing on these inputs)
        for passes in ("0", "1", "2"):
            monkeypatch.setenv("GC_BR_REPSUB", passes)
            e = pkg.BrotliEncoder(lib_path=emu_lib_path, level=6)
            try:
                c = e.code(x)
            finally:
                e.close()
            assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 2), x), (name, passes)
            sizes[passes] = len(c)
        assert sizes["1"] <= sizes["0"] * 1.0005 and sizes["2"] <= sizes["0"] * 1.0005, (name, sizes)       # (the prefix codes move with the symbols: a few bytes either way where nothing is gained)
        if name != "text":
            assert sizes["1"] < sizes["0"], (name, sizes)


def _ring_word(ring_min=2, select=8, quiet=4, warm16=16):
    """hook GC_BR_RING as gc_api.hip reads it: shortest copy at a ring distance | block selection (1 in 2^select sequences of W6's parse, 0 = every block) << 8 |
    single steps behind a copy << 16 | warm-up positions / 16 << 24; 0 = W6 alone"""
    return str(ring_min | (select << 8) | (quiet << 16) | (warm16 << 24))


def _code_with(pkg, emu_lib_path, monkeypatch, x, level=6, **env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = pkg.BrotliEncoder(lib_path=emu_lib_path, level=level)
    try:
        return e.code(x)
    finally:
        e.close()
        for k in env:
            monkeypatch.delenv(k, raising=False)


def test_emu_ring_parse_on_tables_and_machine_code(pkg, O, emu_lib_path, monkeypatch):
    """W6r (gc_lz_window.hip, qualities 5-7): the parse walks the finder's records in order and tries its own last four distances first, as the reference's hasher does
    (hash_longest_match64_inc.h:185-222).  Tables of records and machine code must get clearly smaller than with W6 alone (hook GC_BR_RING=0), text must come out as W6
    left it (the block selection finds no block that returns to its distances), and every stream decodes under the reference decoder."""
    _need_ref(O)
    rng = np.random.default_rng(11)
    rec = rng.integers(0, 256, size=(12000, 24), dtype=np.uint8)
    rec[:, :10] = rec[0, :10]; rec[:, 14:20] = (np.arange(12000)[:, None] >> np.array([0, 8, 16, 0, 8, 16])) & 0xFF
    cases = [("records", rec.reshape(-1).copy(), 0.97), ("text", O.corpus("text-zipf", 3 * BLK + 5), None)]
    obj = O.corpus("real-bin", 8 * BLK)
    if obj.size >= 8 * BLK:
        cases.append(("objects", obj, 0.975))
    for name, x, bar in cases:
        w6 = _code_with(pkg, emu_lib_path, monkeypatch, x, GC_BR_RING="0")
        ring = _code_with(pkg, emu_lib_path, monkeypatch, x)
        forced = _code_with(pkg, emu_lib_path, monkeypatch, x, GC_BR_RING=_ring_word(select=0))
        for c in (w6, ring, forced):
            assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 2), x), name
        if bar is None:
            assert np.array_equal(ring, w6), name                       # no block selected: W6's parse, byte for byte
            assert len(forced) <= 1.02 * len(w6), (name, len(forced), len(w6))        # (W6 looks two positions ahead at this quality, W6r one: 1 % on text -- which is why text stays with W6)
        else:
            assert len(ring) <= bar * len(w6), (name, len(ring), len(w6))
            assert len(forced) <= 1.001 * len(ring), (name, len(forced), len(ring))     # (the selection reads W6's parse: a block whose W6 parse never returns to a distance is left alone although W6r would gain there)


@pytest.mark.parametrize("n", [1, 2, 3, 17, 95, 96, 97, 160, 1000, 4097, BLK - 1, BLK, BLK + 1, 2 * BLK + 12345])
def test_emu_ring_parse_edge_sizes(pkg, O, emu_lib_path, monkeypatch, n):
    """W6r forced on every block (selection off) at the sizes where its sub-blocks are ragged or empty, the input ends inside the 96 bytes in which no ring distance is
    tried, or a block has one position; the three workgroup sizes (4 / 8 / 16 sub-blocks) and no warm-up / no single steps."""
    _need_ref(O)
    rng = np.random.default_rng(n)
    rec = rng.integers(0, 256, size=(n // 12 + 2, 12), dtype=np.uint8); rec[:, :7] = rec[0, :7]
    x = rec.reshape(-1)[:n].copy()
    for geom, word in (("256", _ring_word(select=0)), ("128", _ring_word(select=0, warm16=0)), ("64", _ring_word(select=0, quiet=0, ring_min=3))):
        for level in (5, 7):
            c = _code_with(pkg, emu_lib_path, monkeypatch, x, level=level, GC_BR_RING=word, GC_BR_RING_GEOM=geom)
            assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 2), x), (n, geom, level)


def test_emu_ring_parse_literal_runs_and_long_copies(pkg, O, emu_lib_path, monkeypatch):
    """runs of literals longer than a step sees (random bytes between repeats), copies longer than a step compares (64 bytes) at one distance, a sub-block that is all
    literals beside one that is one copy, and a share of the sequence array that fills up (copies of two bytes at the last distance)"""
    _need_ref(O)
    rng = np.random.default_rng(3)
    r = rng.integers(0, 256, size=40_000, dtype=np.uint8)
    parts = [r, r[:30_000], np.zeros(20_000, dtype=np.uint8), r[5_000:9_000], rng.integers(0, 256, size=25_000, dtype=np.uint8), r[100:30_100]]
    two = np.tile(np.array([1, 2, 0, 0], dtype=np.uint8), 12_000); two[2::4] = rng.integers(0, 256, size=12_000); two[3::4] = rng.integers(0, 256, size=12_000)    # "ab??" repeated: two-byte copies at distance 4
    for x in (np.concatenate(parts), two, np.concatenate([two, r, two])):
        c = _code_with(pkg, emu_lib_path, monkeypatch, x, GC_BR_RING=_ring_word(select=0))
        assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 2), x)


def test_context_tables_equal_the_reference_tables(O, emu_lib_path):
    """B1 computes the CONTEXT_UTF8 lookup of RFC 7932 section 7.1 from the rule it encodes; the reference holds it as a table that its library exports
    (_kBrotliContextLookupTable, C/brotli/common/context.h:99, br_context.c:79-120).  Value for value."""
    import ctypes
    _need_ref(O)
    emu = ctypes.CDLL(emu_lib_path)
    lut = (ctypes.c_uint8 * 512)(); tree = (ctypes.c_uint8 * 64)()
    emu.gc_brotli_context_tables(lut, tree)
    ref = (ctypes.c_uint8 * 2048).in_dll(O.ref("brotli"), "_kBrotliContextLookupTable")
    assert bytes(lut) == bytes(ref)[2 * 512:3 * 512]                                  # CONTEXT_UTF8 = 2
    assert max(tree) == 12 and set(tree) == set(range(13))                           # thirteen trees, every one of them reachable


def test_emu_literal_context_modelling(pkg, O, emu_lib_path, monkeypatch):
    """From quality 5 B1 may code a meta-block's literals with thirteen trees chosen by the two bytes in front (CONTEXT_UTF8 + a static context map).  Every stream must decode
    under the reference decoder -- incl. multi-byte UTF-8 (context ids 0-3), the first bytes of a brotli-mt chunk (no bytes in front), a later piece of a plain stream -- and data
    with context structure must get smaller than with the hook GC_BR_CTX=0 (one tree)."""
    _need_ref(O)
    rng = np.random.default_rng(11)
    words = ["der", "die", "und", "Größe", "Straße", "naïve", "日本語", "текст", "für", "zwölf", "Äpfel", "0123", "x = y;", "\n\t", "{", "}", "(a, b)", "E=mc²", "…"]
    utf8 = np.frombuffer(" ".join(words[i] for i in rng.integers(0, len(words), size=60_000)).encode("utf-8"), dtype=np.uint8).copy()
    # text whose letters depend on what stands in front of them: a word per line, capitals after full stops, digits after '=': what the static map separates
    struct_txt = np.frombuffer("".join("Key%d = %d. Next line follows here\n" % (i % 97, (i * 7919) % 100003) for i in range(40_000)).encode(), dtype=np.uint8).copy()
    cases = (("utf8", utf8), ("structured", struct_txt[:9 * BLK + 77]), ("silesia", O.corpus("silesia-like", 3 * BLK + 5)), ("text", O.corpus("text-zipf", 2 * BLK)))
    for name, x in cases:
        sizes = {}
        for ctx in ("0", "1"):
            monkeypatch.setenv("GC_BR_CTX", ctx)
            e = pkg.BrotliEncoder(lib_path=emu_lib_path, level=1 if name == "structured" else 6)      # (level 1: chunks of 8 blocks, so the nine blocks hold a chunk boundary; the hook switches the modelling on below quality 5 too)
            try:
                c = e.code(x)
            finally:
                e.close()
            assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 2), x), (name, ctx)
            sizes[ctx] = len(c)
        assert sizes["1"] <= sizes["0"] * 1.001, (name, sizes)                          # (one tree is always among the choices; the choice is made from an estimate)
        if name == "silesia":
            assert sizes["1"] < sizes["0"] * 0.995, (name, sizes)
    # plain stream in pieces: the first meta-block of a later piece keeps one tree (the bytes in front of it are another call's)
    monkeypatch.setenv("GC_BR_CTX", "1"); monkeypatch.setenv("HIPEMU_DEVICES", "2")
    x = struct_txt[:2 * 8 * BLK + 4321]                                    # three pieces at quality 1 (chunk = 1 MiB) through the host scheduler
    m = pkg.MultiEncoder("brotli", 1, lib_path=emu_lib_path)
    try:
        c = m.code(x, flags=1)                                              # GC_BROTLI_PLAIN
    finally:
        m.close()
    assert np.array_equal(O.ref_brotli_decompress(c, x.size), x)


def test_golden_fixtures_decode_under_reference(O):
    """the reference's own regression fixtures (tests/regr-arc/test.txt.br, .br-mt.br) pin the decoder side of the oracle"""
    _need_ref(O)
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    plain = np.fromfile(os.path.join(g, "test.txt.br"), dtype=np.uint8)
    mt = np.fromfile(os.path.join(g, "test.txt.br-mt.br"), dtype=np.uint8)
    a = O.ref_brotli_decompress(plain, 1 << 20)
    b = O.ref_brotlimt_decompress(mt, 1 << 20)
    assert a.size > 0 and np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("n", [0, 1, 3, 64, 4097, BLK - 1, BLK, BLK + 1, 3 * BLK + 17])
def test_gpu_edge_sizes(O, gpu_br, n):
    _roundtrip(O, gpu_br[6], O.corpus("text-zipf", n))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like", "web-text", "random", "zeros"])
def test_gpu_corpora(O, gpu_br, kind):
    x = O.corpus(kind, 8 * 1024 * 1024 + 999)
    for lv in (1, 6):
        _roundtrip(O, gpu_br[lv], x)


@pytest.mark.gpu
def test_gpu_bytes_equal_emulator_bytes(O, gpu_br, emu_br):
    for kind in ("text-zipf", "silesia-like"):
        x = O.corpus(kind, 2 * BLK + 1234)
        assert np.array_equal(gpu_br[6].code(x), emu_br[6].code(x)), kind


@pytest.mark.gpu
def test_gpu_100mb_web_text_device_api(O, gpu_br):
    import torch
    _need_ref(O)
    n = 100_000_000
    x = O.corpus("web-text", n)
    enc = gpu_br[6]
    d_src = torch.from_numpy(x).to("cuda:0")
    cap = enc.compress_bound(n)
    d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    enc.code_device(d_src.data_ptr(), n, d_dst.data_ptr(), cap)
    size = enc.finish()
    comp = d_dst[:size].cpu().numpy()
    assert np.array_equal(O.ref_brotlimt_decompress(comp, n, threads=8), x)
    assert enc.last_timing_ms()["total"] > 0
