"""The checker itself: the plain-C zstd decoder restatement (oracle/zstd_frame_dec.c) is pinned to the
reference's golden fixture and to the reference's own encoder/decoder (oracle/_ref, when present)."""
import hashlib
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# tests/regression.test:31-89 of the reference: decoded test.txt is 1 000 000 bytes with this SHA-256
TEST_TXT_SHA256 = "aeda0f81c8376d1678af53927a08cf641cafab8b68aef509c881eb0be0bc3c97"


def test_golden_fixture_port_decoder(O):
    comp = open(os.path.join(GOLD, "test.txt.zstd"), "rb").read()
    out = O.port_zstd_decompress(comp, 2_000_000)
    assert out.size == 1_000_000
    assert hashlib.sha256(out.tobytes()).hexdigest() == TEST_TXT_SHA256
    assert out[:5].tobytes() == b"TEST\n" and out[-5:].tobytes() == b"\nEND."


def test_golden_fixture_reference_decoder(O):
    if O.ref("zstd") is None:
        pytest.skip("oracle/_ref not built")
    comp = open(os.path.join(GOLD, "test.txt.zstd"), "rb").read()
    out = O.ref_zstd_decompress(comp, 2_000_000)
    assert hashlib.sha256(out.tobytes()).hexdigest() == TEST_TXT_SHA256


def test_golden_brotli_fixtures_reference_decoder(O):
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref not built")
    plain = O.ref_brotli_decompress(open(os.path.join(GOLD, "test.txt.br"), "rb").read(), 2_000_000)
    assert hashlib.sha256(plain.tobytes()).hexdigest() == TEST_TXT_SHA256
    framed = O.ref_brotlimt_decompress(open(os.path.join(GOLD, "test.txt.br-mt.br"), "rb").read(), 2_000_000)
    assert hashlib.sha256(framed.tobytes()).hexdigest() == TEST_TXT_SHA256


@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like", "random", "zeros"])
def test_port_decoder_matches_reference_encoder(O, kind):
    if O.ref("zstd") is None:
        pytest.skip("oracle/_ref not built")
    x = O.corpus(kind, 1_500_000)
    for level in (1, 3, 6, 12, 19):
        c = O.ref_zstd_compress(x, level)
        assert np.array_equal(O.port_zstd_decompress(c, x.size), x)
        assert np.array_equal(O.ref_zstd_decompress(c, x.size), x)
    c = O.ref_zstd_compress(x, 3, piece=128 * 1024)          # concatenated independent frames
    assert np.array_equal(O.port_zstd_decompress(c, x.size), x)


def test_port_decoder_rejects_corruption(O):
    if O.ref("zstd") is None:
        pytest.skip("oracle/_ref not built")
    x = O.corpus("text-zipf", 200_000)
    c = O.ref_zstd_compress(x, 3).copy()
    bad = c.copy(); bad[len(bad) // 2] ^= 0x55
    try:
        y = O.port_zstd_decompress(bad, x.size)
        assert not np.array_equal(y, x)
    except ValueError:
        pass
    with pytest.raises(ValueError):
        O.port_zstd_decompress(c[:-3], x.size)


def test_skippable_and_empty_frames(O):
    skippable = bytes([0x50, 0x2A, 0x4D, 0x18, 4, 0, 0, 0, 1, 2, 3, 4])
    empty = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00])
    out = O.port_zstd_decompress(skippable + empty + skippable, 16)
    assert out.size == 0
    if O.ref("zstd") is not None:
        assert O.ref_zstd_decompress(skippable + empty, 16).size == 0


def test_xxh64_known_answers(O):
    p = O.port()
    assert p.gco_xxh64(None, 0, 0) == 0xEF46DB3751D8E999            # XXH64("") seed 0
    a = np.frombuffer(b"abc", dtype=np.uint8)
    assert p.gco_xxh64(a.ctypes.data, 3, 0) == 0x44BC2CF5AD770999    # XXH64("abc")


def test_corpus_generators_are_deterministic(O):
    for kind in ("text-zipf", "lz-7zip", "silesia-like", "web-text"):
        a = O.corpus(kind, 300_000); b = O.corpus(kind, 300_000)
        assert np.array_equal(a, b)
    # lz-7zip restates CBenchRandomGenerator::GenerateLz (Bench.cpp:191-256): first 1024 bytes are the MWC stream's low bytes
    a = O.corpus("lz-7zip", 4096)
    a1, a2 = 362436069, 521288629
    for i in range(8):
        a1 = (36969 * (a1 & 0xffff) + (a1 >> 16)) & 0xFFFFFFFF
        a2 = (18000 * (a2 & 0xffff) + (a2 >> 16)) & 0xFFFFFFFF
        r = (((a1 << 16) & 0xFFFFFFFF) + a2) & 0xFFFFFFFF
        assert a[i] == ((r >> 1) & 0xFF)


# ---------------------------------------------------------------------------------------------- match finder oracle (the next row)
def test_hc4_restatement_matches_the_golden_lists(O):
    """oracle/lzfind_hc4.c (what Hc4_MatchFinder_GetMatches returns for every position, restated in data-parallel form) against lists the reference's
    own C/LzFind.c produced (tests/golden/make_hc4_fixture.py)."""
    g = np.load(os.path.join(GOLD, "hc4_matches.npz"))
    for name in ("text", "lz", "sil"):
        hist, cut, nice = (int(v) for v in g[name + "_params"])
        counts, pairs = O.port_hc4_matches(g[name + "_input"], hist, cut, nice)
        assert np.array_equal(counts, g[name + "_counts"].astype(np.uint32)), name
        assert np.array_equal(pairs, g[name + "_pairs"]), name


@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like", "random", "zeros"])
def test_hc4_restatement_matches_the_reference(O, kind):
    if O.ref("lzfind") is None:
        pytest.skip("oracle/_ref not built")
    x = np.zeros(40_000, dtype=np.uint8) if kind == "zeros" else O.corpus(kind, 150_000)
    for hist, cut, nice in ((1 << 20, 32, 64), (4096, 8, 273), (65536, 1, 32), (1 << 16, 48, 5), (300, 16, 128)):     # window wraps, cut 1, shortest / longest nice length
        c1, p1 = O.ref_lzfind_matches(x, hist, False, 4, cut, nice)
        c2, p2 = O.port_hc4_matches(x, hist, cut, nice)
        assert np.array_equal(c1, c2) and np.array_equal(p1, p2), (kind, hist, cut, nice)
    for n in (0, 1, 3, 4, 5, 9):
        a, b = O.ref_lzfind_matches(x[:n], 1 << 16, False, 4, 32, 64), O.port_hc4_matches(x[:n], 1 << 16, 32, 64)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
