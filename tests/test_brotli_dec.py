"""BROTLI decoding on the device (`gc_brotli_dec.hip`): the content must be bit-exact for brotli-mt streams of the REFERENCE's encoder at every quality (incl. the
static dictionary with its transforms, hundreds of prefix codes per meta-block at qualities 10-11, uncompressed and empty meta-blocks) and of this engine's encoder; damaged streams
must be refused or decode to the same content, never crash.  CPU: the kernel under the SIMT emulator; GPU: the product library, larger inputs.  The oracle is the reference's own
decoder (oracle/_ref) -- and the reference's golden vectors tests/regr-arc/test.txt.br / .br-mt.br (tests/golden)."""
import hashlib
import os
import struct
import sys

import numpy as np
import pytest

MiB = 1 << 20
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
# the reference's own golden vector (tests/regression.test): test.txt decodes to 1 000 000 bytes with this SHA-256 (the zstd fixture of test_zstd_dec.py holds the same file)
TEST_TXT_SHA256 = "aeda0f81c8376d1678af53927a08cf641cafab8b68aef509c881eb0be0bc3c97"
UNSUPPORTED = "GC_ERR_UNSUPPORTED"


def _need_ref(O):
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref (reference brotli) is not built")


@pytest.fixture(scope="module")
def emu_dec(pkg, O, emu_lib_path):
    _need_ref(O)
    d = pkg.BrotliDecoder(lib_path=emu_lib_path)
    d.set_dictionary(O.ref_brotli_dictionary())
    yield d
    d.close()


@pytest.fixture(scope="module")
def gpu_dec(pkg, O, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    d = pkg.BrotliDecoder(device=0)              # raises loudly if libgpucodec.so is missing or no gfx950 device opens
    if O.ref("brotli") is not None:
        d.set_dictionary(O.ref_brotli_dictionary())
    yield d
    d.close()


def _check(dec, comp, want, capacity=None):
    out = dec.code(comp, capacity=capacity)
    assert out.size == len(want)
    assert out.tobytes() == bytes(want)


# ---------------------------------------------------------------------------------------------- CPU: tables, scan
def test_transforms_header_equals_the_reference_table(O):
    """gc_brotli_transforms.h (generated, committed) against BrotliGetTransforms() of the reference: 121 x {prefix, elementary transform, suffix}"""
    _need_ref(O)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_brotli_transforms as G
    pool, tri = G.layout(G.table())
    src = open(os.path.join(ROOT, "7-zip-zstd_amd", "csrc", "gc_brotli_transforms.h")).read()
    assert "kdAffixPool[%d] = {%s}" % (len(pool), ",".join(str(b) for b in pool)) in src
    assert "kdTransforms[121][3] = {%s}" % ",".join("{%d,%d,%d}" % t for t in tri) in src


def test_dictionary_is_checked_against_the_rfc_crc(pkg, O, emu_lib_path):
    _need_ref(O)
    d = pkg.BrotliDecoder(lib_path=emu_lib_path)
    try:
        good = O.ref_brotli_dictionary()
        assert good.size == 122784
        bad = good.copy(); bad[1000] ^= 1
        with pytest.raises(pkg.GpuCodecError):
            d.set_dictionary(bad)
        with pytest.raises(pkg.GpuCodecError):
            d.set_dictionary(good[:-1])
        d.set_dictionary(good)
        assert d.has_dictionary()
    finally:
        d.close()


def test_scan_walks_brotli_mt_frames(pkg, O, emu_lib_path):
    _need_ref(O)
    d = pkg.BrotliDecoder(lib_path=emu_lib_path)
    try:
        x = O.corpus("text-zipf", 3 * MiB + 5)
        c = O.ref_brotlimt_compress(x, 1, 3)
        chunks, n, cap, used = d.scan(c)
        assert n >= 1 and used == len(c) and cap >= x.size
        off = 0
        for i in range(n):
            magic, eight, csize, br, hint = struct.unpack_from("<IIIHH", c, off)
            assert (magic, eight, br) == (0x184D2A50, 8, 0x5242)
            assert (chunks[i].src_off, chunks[i].src_size, chunks[i].capacity) == (off + 16, csize, hint << 16)
            off += 16 + csize
        # an input that ends inside a frame: the whole frames in front of it
        _, n2, _, used2 = d.scan(c[:len(c) - 7])
        assert n2 == n - 1 and used2 == chunks[n - 1].src_off - 16
        with pytest.raises(pkg.GpuCodecError):
            d.scan(b"\x00" * 32)
    finally:
        d.close()


# ---------------------------------------------------------------------------------------------- CPU: the kernel under the emulator
def test_emu_golden_vectors(emu_dec):
    mt = np.fromfile(os.path.join(GOLD, "test.txt.br-mt.br"), dtype=np.uint8)
    plain = np.fromfile(os.path.join(GOLD, "test.txt.br"), dtype=np.uint8)
    a = emu_dec.code(mt)
    assert a.size == 1_000_000 and hashlib.sha256(a.tobytes()).hexdigest() == TEST_TXT_SHA256
    b = emu_dec.code(plain, capacity=1_000_000)                   # a bare RFC 7932 stream: one chunk
    assert b.tobytes() == a.tobytes()
    with pytest.raises(Exception):
        emu_dec.code(plain, capacity=999_999)


@pytest.mark.parametrize("kind,q,n", [("text-zipf", 0, 150_000), ("text-zipf", 1, 200_000), ("real-src", 2, 150_000), ("silesia-like", 4, 200_000), ("real-src", 5, 200_000),
                                      ("text-zipf", 6, 200_000), ("lz-7zip", 6, 150_000), ("web-text", 9, 200_000), ("real-src", 10, 150_000), ("silesia-like", 11, 200_000),
                                      ("random", 6, 70_000), ("zeros", 6, 300_000), ("text-zipf", 6, 0), ("text-zipf", 6, 1), ("text-zipf", 11, 5)])
def test_emu_reference_streams(O, emu_dec, kind, q, n):
    x = np.zeros(n, dtype=np.uint8) if kind == "zeros" else (np.random.default_rng(3).integers(0, 256, size=n, dtype=np.uint8) if kind == "random" else O.corpus(kind, n))
    if n and x.size < n:
        pytest.skip("the image holds no %s data" % kind)
    c = O.ref_brotlimt_compress(x, q, 2)
    _check(emu_dec, c, x)


def test_emu_meta_blocks_that_outgrow_lds_take_hbm_pages(O, emu_dec, monkeypatch):
    """A meta-block with more prefix codes than the wave's LDS arena holds (the reference's qualities 10-11 write up to 256 literal and distance trees) has its header read again
    with a page of HBM behind the arena.  The hook shrinks the arena so that small inputs get there; 70 chunks that all need a page outgrow the first pool of 64 (second round)."""
    monkeypatch.setenv("GC_BRD_LDS", "3072")                     # (context maps, modes and directories stay in LDS: they are read per symbol)
    x = np.concatenate([O.corpus(k, 130_000) for k in ("silesia-like", "real-src", "lz-7zip", "web-text")])
    _check(emu_dec, O.ref_brotlimt_compress(x, 11, 1), x)
    y = O.corpus("text-zipf", 70 * 6000)
    c = np.concatenate([O.ref_brotlimt_compress(y[i * 6000:(i + 1) * 6000], 5 + i % 3, 1) for i in range(70)])
    assert emu_dec.scan(c)[1] == 70
    _check(emu_dec, c, y)


@pytest.mark.parametrize("instance", [1, 2, 3, 4])
def test_emu_every_kernel_instance(O, emu_dec, monkeypatch, instance):
    """The decoder has four instances of its kernel (LDS arena / output ring of 64, 32, 16, 4 KiB), chosen by the number of chunks; the hook forces one.  Inputs that take every
    way a copy can go: near (out of the ring), far (out of HBM, behind what the ring has handed over), longer than half the ring (HBM to HBM), reaching into itself at every size
    (runs of one byte, a period of 3 and of 70 000), literal runs longer than the ring (random bytes), the dictionary."""
    monkeypatch.setenv("GC_BRD_INSTANCE", str(instance))
    rng = np.random.default_rng(instance)
    block = O.corpus("text-zipf", 70_000)
    x = np.concatenate([O.corpus("real-src", 120_000), np.zeros(150_000, dtype=np.uint8), np.tile(np.frombuffer(b"abc", dtype=np.uint8), 30_000), block, block, block[:50_000],
                        rng.integers(0, 256, size=80_000, dtype=np.uint8), O.corpus("lz-7zip", 150_000), block[10_000:30_000]])
    for q in (1, 5, 9, 11):
        _check(emu_dec, O.ref_brotlimt_compress(x, q, 1), x)
    _check(emu_dec, O.ref_brotli_compress(x, 6, 16), x, capacity=x.size)        # a window of 64 KiB: every distance is near


def test_emu_bare_streams_and_windows(O, emu_dec):
    x = O.corpus("text-zipf", 180_000)
    for q, lgwin in ((1, 10), (5, 16), (6, 22), (9, 24), (11, 18)):
        c = O.ref_brotli_compress(x, q, lgwin)
        _check(emu_dec, c, x, capacity=x.size)
    with pytest.raises(Exception):
        emu_dec.code(O.ref_brotli_compress(x, 6, 22), capacity=x.size - 1)


def test_emu_own_streams(pkg, O, emu_dec, emu_lib_path):
    for level, n in ((1, 300_000), (6, 400_000), (9, 300_000)):
        x = O.corpus("silesia-like" if level == 6 else "text-zipf", n)
        e = pkg.BrotliEncoder(lib_path=emu_lib_path, level=level)
        try:
            c = e.code(x)
        finally:
            e.close()
        _check(emu_dec, c, x)


def test_emu_without_the_dictionary_a_reference_to_it_is_unsupported(pkg, O, emu_lib_path):
    _need_ref(O)
    d = pkg.BrotliDecoder(lib_path=emu_lib_path)
    try:
        d.set_dictionary(b"")                                     # forget it
        x = O.corpus("real-src", 200_000)
        if x.size < 200_000:
            pytest.skip("the image holds no real-src data")
        with pytest.raises(pkg.GpuCodecError, match=UNSUPPORTED):
            d.code(O.ref_brotlimt_compress(x, 6, 1))
        e = pkg.BrotliEncoder(lib_path=emu_lib_path, level=6)     # this engine's own streams never refer to it
        try:
            c = e.code(x)
        finally:
            e.close()
        _check(d, c, x)
    finally:
        d.set_dictionary(O.ref_brotli_dictionary())
        d.close()


def _damage(dec, c, x, rng, rounds):
    refused = same = 0
    for _ in range(rounds):
        bad = c.copy()
        k = int(rng.integers(0, 4))
        if k == 0:
            bad[int(rng.integers(16, bad.size))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            bad = bad[:int(rng.integers(1, bad.size))]
        elif k == 2:
            i = int(rng.integers(16, bad.size)); bad[i:i + 8] = rng.integers(0, 256, size=bad[i:i + 8].size, dtype=np.uint8)
        else:
            struct.pack_into("<H", bad, 14, max(0, struct.unpack_from("<H", bad, 14)[0] - 1))        # the hint one unit too small
        try:
            y = dec.code(bad, capacity=x.size + (1 << 16))
        except Exception:
            refused += 1
            continue
        if y.size == x.size and y.tobytes() == x.tobytes():
            same += 1
        # (a flipped bit may still be a valid stream of other content: brotli carries no checksum -- what must not happen is a crash or a hang)
    return refused, same


def test_emu_damaged_streams_are_refused_or_harmless(O, emu_dec):
    rng = np.random.default_rng(5)
    x = O.corpus("text-zipf", 90_000)
    refused = 0
    for q in (1, 6, 11):
        c = O.ref_brotlimt_compress(x, q, 1)
        r, _ = _damage(emu_dec, c, x, rng, 12)
        refused += r
    assert refused > 0


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_golden_vectors(gpu_dec):
    mt = np.fromfile(os.path.join(GOLD, "test.txt.br-mt.br"), dtype=np.uint8)
    a = gpu_dec.code(mt)
    assert a.size == 1_000_000 and hashlib.sha256(a.tobytes()).hexdigest() == TEST_TXT_SHA256
    if gpu_dec.has_dictionary():
        b = gpu_dec.code(np.fromfile(os.path.join(GOLD, "test.txt.br"), dtype=np.uint8), capacity=1_000_000)
        assert b.tobytes() == a.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["text-zipf", "silesia-like", "lz-7zip", "real-src", "real-bin", "web-text"])
@pytest.mark.parametrize("q", [0, 1, 2, 4, 5, 6, 9, 10, 11])
def test_gpu_reference_streams(O, gpu_dec, kind, q):
    _need_ref(O)
    n = (4 if q >= 10 else 24) * MiB + 12345
    x = O.corpus(kind, n)
    if x.size < MiB:
        pytest.skip("the image holds no %s data" % kind)
    c = O.ref_brotlimt_compress(x, q, min(os.cpu_count() or 1, 64))
    _check(gpu_dec, c, x)


@pytest.mark.gpu
@pytest.mark.parametrize("level,kind,n", [(1, "text-zipf", 64 * MiB), (6, "web-text", 256 * MiB + 777), (6, "real-bin", 64 * MiB), (9, "lz-7zip", 48 * MiB), (11, "text-zipf", 24 * MiB), (6, "text-zipf", 0), (6, "text-zipf", 1)])
def test_gpu_own_streams(pkg, O, gpu_dec, level, kind, n):
    x = O.corpus(kind, n) if n else np.empty(0, dtype=np.uint8)
    e = pkg.BrotliEncoder(level=level)
    try:
        c = e.code(x)
    finally:
        e.close()
    _check(gpu_dec, c, x)


@pytest.mark.gpu
def test_gpu_bare_streams(O, gpu_dec):
    _need_ref(O)
    x = O.corpus("silesia-like", 3 * MiB + 1)
    for q, lgwin in ((1, 16), (6, 22), (9, 24), (11, 20)):
        _check(gpu_dec, O.ref_brotli_compress(x, q, lgwin), x, capacity=x.size)


@pytest.mark.gpu
def test_gpu_damaged_streams_are_refused_or_harmless(O, gpu_dec):
    _need_ref(O)
    rng = np.random.default_rng(6)
    x = O.corpus("silesia-like", 2 * MiB)
    refused = 0
    for q in (1, 6, 9, 11):
        c = O.ref_brotlimt_compress(x, q, 4)
        r, _ = _damage(gpu_dec, c, x, rng, 60)
        refused += r
    assert refused > 0
    _check(gpu_dec, O.ref_brotlimt_compress(x, 6, 4), x)          # the context still works
