"""ZSTD decoding on the device: the content must be bit-exact for streams of the REFERENCE's encoder (every level family,
content checksum, streamed frames without a content size, long offsets, concatenated and skippable frames) and of this engine's encoder,
and damaged streams must be refused, never crash.  CPU: the kernel under the SIMT emulator; GPU: the product library, larger inputs."""
import hashlib
import os
import struct

import numpy as np
import pytest

MiB = 1 << 20
GOLD = os.path.join(os.path.dirname(__file__), "golden")
# the reference's own golden vector (tests/regression.test:31-89): test.txt.zstd decodes to 1 000 000 bytes with this SHA-256
TEST_TXT_SHA256 = "aeda0f81c8376d1678af53927a08cf641cafab8b68aef509c881eb0be0bc3c97"
KINDS = ["silesia-like", "text-zipf", "lz-7zip", "random", "zeros", "runs"]


def _corpus(O, kind, n):
    if n == 0:
        return np.empty(0, dtype=np.uint8)
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint8)
    if kind == "runs":                       # RLE blocks, RLE literals, long matches with tiny offsets
        rng = np.random.default_rng(7)
        out = np.repeat(rng.integers(0, 256, size=n // 700 + 2, dtype=np.uint8), rng.integers(1, 1400, size=n // 700 + 2))
        return np.ascontiguousarray(np.resize(out, n))
    return O.corpus(kind, n)


def _check(dec, comp, want, capacity=None):
    out = dec.code(comp, capacity=capacity if capacity is not None else len(want) + 64)
    assert out.size == len(want)
    assert out.tobytes() == bytes(want)


@pytest.fixture(scope="module")
def emu_dec(pkg, emu_lib_path):
    d = pkg.ZstdDecoder(lib_path=emu_lib_path)
    yield d
    d.close()


@pytest.fixture(scope="module")
def gpu_dec(pkg, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    d = pkg.ZstdDecoder(device=0)
    yield d
    d.close()


def _golden(dec):
    comp = open(os.path.join(GOLD, "test.txt.zstd"), "rb").read()
    out = dec.code(comp, capacity=2_000_000)
    assert out.size == 1_000_000 and hashlib.sha256(out.tobytes()).hexdigest() == TEST_TXT_SHA256
    assert out[:5].tobytes() == b"TEST\n" and out[-5:].tobytes() == b"\nEND."


def test_emu_golden_fixture_of_the_reference(emu_dec):
    _golden(emu_dec)


@pytest.mark.gpu
def test_gpu_golden_fixture_of_the_reference(gpu_dec):
    _golden(gpu_dec)


@pytest.mark.parametrize("kind", KINDS)
def test_emu_reference_streams(O, emu_dec, kind):
    for n, level in ((0, 3), (1, 3), (5000, 1), (131072, 3), (131073, 6), (300_000, 19), (200_000, 22)):
        x = _corpus(O, kind, n)
        _check(emu_dec, O.ref_zstd_compress(x.tobytes(), level), x.tobytes())


def test_emu_reference_frame_options(O, emu_dec):
    x = _corpus(O, "silesia-like", 400_000)
    for kw in (dict(checksum=True), dict(streamed=True), dict(checksum=True, streamed=True), dict(ldm=True, checksum=True)):
        for level in (1, 5, 17):
            _check(emu_dec, O.ref_zstd_compress_opts(x.tobytes(), level, **kw), x.tobytes())
    # empty and tiny frames with checksum / without size
    for n in (0, 1, 31, 32, 33, 63):
        y = x[:n].tobytes()
        _check(emu_dec, O.ref_zstd_compress_opts(y, 3, checksum=True, streamed=True), y)


def test_emu_concatenated_and_skippable_frames(O, emu_dec):
    x = _corpus(O, "text-zipf", 600_000).tobytes()
    comp = O.ref_zstd_compress(x, 3, piece=100_000).tobytes()                 # 6 frames that state their sizes
    frames, n, total = emu_dec.scan(comp)
    assert n == 6 and total == len(x)
    assert [frames[i].dst_off for i in range(n)] == [100_000 * i for i in range(6)]
    _check(emu_dec, comp, x)
    skip = struct.pack("<II", 0x184D2A53, 11) + b"hello world"
    a, b = O.ref_zstd_compress_opts(x[:250_000], 3, streamed=True).tobytes(), O.ref_zstd_compress_opts(x[250_000:], 6, checksum=True).tobytes()
    mixed = skip + a + skip + b + skip                                         # a frame of unknown size closes a batch
    frames, n, total = emu_dec.scan(mixed)
    assert n == 2 and total is None
    _check(emu_dec, mixed, x, capacity=len(x))
    with pytest.raises(Exception):
        emu_dec.code(mixed, capacity=len(x) - 1)                               # destination too small


@pytest.mark.parametrize("level", [1, 3, 19])
def test_emu_own_encoder_roundtrip(pkg, O, emu_lib_path, emu_dec, level, monkeypatch):
    monkeypatch.setenv("GC_FRAME_BLOCKS", "2")                                 # frames of 256 KiB so that a small input carries several
    x = _corpus(O, "silesia-like", 600_000 if level < 16 else 300_000)      # (the price-based parse is slow under the emulator)
    enc = pkg.ZstdEncoder(lib_path=emu_lib_path, level=level)
    try:
        comp = enc.code(x)
        enc.set_option(enc.OPT_ZSTD_SEEK_TABLE, 1)
        comp2 = enc.code(x)
    finally:
        enc.close()
    frames, n, total = emu_dec.scan(comp)
    assert total == x.size and n >= 2
    _check(emu_dec, comp, x.tobytes())
    _check(emu_dec, comp2, x.tobytes())                                        # the seek table is a skippable frame


def test_emu_damaged_streams_are_refused(pkg, O, emu_dec):
    x = _corpus(O, "silesia-like", 200_000).tobytes()
    comp = bytearray(O.ref_zstd_compress_opts(x, 3, checksum=True).tobytes())
    rng = np.random.default_rng(3)
    refused = 0
    for _ in range(24):
        bad = bytearray(comp)
        pos = int(rng.integers(0, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            out = emu_dec.code(bytes(bad), capacity=len(x) + 64)
            assert out.tobytes() == x           # a flip the format does not notice must not change the content (checksum present)
        except pkg.GpuCodecError:
            refused += 1
    assert refused >= 21
    for cut in (0, 3, 4, 9, len(comp) // 2, len(comp) - 1):                    # truncated
        if cut == 0:
            continue
        with pytest.raises(pkg.GpuCodecError):
            emu_dec.code(bytes(comp[:cut]), capacity=len(x) + 64)
    with pytest.raises(pkg.GpuCodecError):
        emu_dec.code(b"\x00" * 16)                                             # not a zstd stream


def _both_paths(pkg, O, lib_kw, monkeypatch, sizes, flips):
    """The two execution stages (one workgroup per frame, blocks in order / all blocks at once through byte pointers and pointer jumping) must
    give the same bytes and the same verdicts; by default frames of more than one block take the wide one (wide_rounds() says which ran)."""
    x = _corpus(O, "silesia-like", sizes[0]).tobytes()
    z = np.zeros(sizes[1], dtype=np.uint8); z[::4097] = 7; z = z.tobytes()                     # long chains: every match copies the one before
    streams = [(O.ref_zstd_compress(x, 3).tobytes(), x), (O.ref_zstd_compress_opts(x, 19, checksum=True, streamed=True).tobytes(), x),
               (O.ref_zstd_compress(x, 5, piece=len(x) // 5 + 1).tobytes(), x), (O.ref_zstd_compress_opts(z, 3, checksum=True).tobytes(), z)]
    bad = bytearray(streams[1][0]); verdicts = []
    for wide in ("0", "1"):
        monkeypatch.setenv("GC_ZD_WIDE", wide)
        dec = pkg.ZstdDecoder(**lib_kw)
        try:
            for comp, want in streams:
                _check(dec, comp, want)
                assert (dec.wide_rounds() > 0) == (wide == "1")
            verdicts.append([])
            for pos in range(len(bad) // 2, len(bad) // 2 + flips):                                # (the frame carries a checksum: refused or the same content)
                bad[pos] ^= 0x10
                try:
                    assert dec.code(bytes(bad), capacity=len(x) + 64).tobytes() == x; verdicts[-1].append(0)
                except pkg.GpuCodecError:
                    verdicts[-1].append(1)
                bad[pos] ^= 0x10
            with pytest.raises(pkg.GpuCodecError):
                dec.code(streams[1][0], capacity=len(x) - 1)                                    # (a frame that does not state its size)
        finally:
            dec.close()
    assert verdicts[0] == verdicts[1] and sum(verdicts[0]) >= flips * 3 // 4
    monkeypatch.delenv("GC_ZD_WIDE")
    dec = pkg.ZstdDecoder(**lib_kw)
    try:
        _check(dec, streams[0][0], x); assert dec.wide_rounds() > 0                             # frames of several blocks: wide by default
        _check(dec, streams[2][0], x); assert (dec.wide_rounds() > 0) == (len(x) // 5 > 131072)
        ones = O.ref_zstd_compress(x[:1_000_000], 3, piece=100_000).tobytes()
        _check(dec, ones, x[:1_000_000]); assert dec.wide_rounds() == 0                         # frames of one block each: one workgroup per frame
        monkeypatch.setenv("GC_ZD_WIDE_NOMEM", "1")                                             # no room for the pointers: the frame kernel steps in
        _check(dec, streams[1][0], x); assert dec.wide_rounds() == 0
    finally:
        dec.close()


def _several_blocks_per_wave(pkg, O, lib_kw, monkeypatch, sizes):
    """The sequences kernel that takes six blocks per wave (four lanes each, quad-permute moves between them; the default from 1024 blocks on) must
    agree with the one-block-per-wave kernel: every kind of stream, long matches included (their extra bits were what a mis-folded DPP move lost on
    the MI355X: tools/gpu_seqv_variants.py)."""
    monkeypatch.setenv("GC_ZD_SEQV", "1")
    dec = pkg.ZstdDecoder(**lib_kw)
    try:
        for kind in KINDS:
            for n, level in sizes:
                x = _corpus(O, kind, n).tobytes()
                _check(dec, O.ref_zstd_compress(x, level).tobytes(), x)
        x = _corpus(O, "silesia-like", sizes[-1][0]).tobytes()
        _check(dec, O.ref_zstd_compress_opts(x, 5, checksum=True, streamed=True).tobytes(), x)
        _check(dec, O.ref_zstd_compress(x, 3, piece=70_000).tobytes(), x)                        # frames of one block: several frames per wave
        bad = bytearray(O.ref_zstd_compress_opts(x, 3, checksum=True).tobytes())
        for pos in range(len(bad) // 3, len(bad) // 3 + 6):
            bad[pos] ^= 0x04
            try:
                assert dec.code(bytes(bad), capacity=len(x) + 64).tobytes() == x
            except pkg.GpuCodecError:
                pass
            bad[pos] ^= 0x04
    finally:
        dec.close()


def test_emu_sequences_kernel_several_blocks_per_wave(pkg, O, emu_lib_path, monkeypatch):
    _several_blocks_per_wave(pkg, O, dict(lib_path=emu_lib_path), monkeypatch, ((5000, 1), (300_000, 3), (200_000, 19)))


def test_emu_small_batches(pkg, O, emu_lib_path, monkeypatch):
    """The launches of a call cover at most GC_ZD_BATCH_BYTES of content at a time (workspaces grow with the batch); the test hook makes the batches
    small: 7 frames in batches of one or two, a frame larger than the cap alone in its batch, both execution stages."""
    x = _corpus(O, "text-zipf", 900_000).tobytes()
    comp = O.ref_zstd_compress(x[:600_000], 3, piece=100_000).tobytes() + O.ref_zstd_compress_opts(x[600_000:], 3, checksum=True).tobytes()
    monkeypatch.setenv("GC_ZD_BATCH_KIB", "150")
    dec = pkg.ZstdDecoder(lib_path=emu_lib_path)
    try:
        _check(dec, comp, x)
        assert dec.wide_rounds() > 0                       # (the last frame has three blocks)
    finally:
        dec.close()


def test_emu_both_execution_paths(pkg, O, emu_lib_path, monkeypatch):
    _both_paths(pkg, O, dict(lib_path=emu_lib_path), monkeypatch, (280_000, 150_000), 4)


# ------------------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_sequences_kernel_several_blocks_per_wave(pkg, O, gpu_dec, gpu_hooks_kw, monkeypatch):
    _several_blocks_per_wave(pkg, O, gpu_hooks_kw, monkeypatch, ((5000, 1), (3 * MiB + 17, 3), (4 * MiB, 19)))
    monkeypatch.delenv("GC_ZD_SEQV")
    x = _corpus(O, "silesia-like", 160 * MiB + 5).tobytes()                                      # 1281 blocks: the default takes this kernel
    _check(gpu_dec, O.ref_zstd_compress(x, 1).tobytes(), x)


@pytest.mark.gpu
def test_gpu_both_execution_paths(pkg, O, gpu_dec, gpu_hooks_kw, monkeypatch):
    _both_paths(pkg, O, gpu_hooks_kw, monkeypatch, (48 * MiB + 321, 32 * MiB), 24)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_gpu_reference_streams(O, gpu_dec, kind):
    for n, level in ((0, 3), (1, 1), (131072, 3), (3 * MiB + 17, 1), (8 * MiB, 3), (5 * MiB, 19)):
        x = _corpus(O, kind, n)
        _check(gpu_dec, O.ref_zstd_compress(x.tobytes(), level), x.tobytes())


@pytest.mark.gpu
def test_gpu_reference_frame_options(O, gpu_dec):
    x = _corpus(O, "silesia-like", 20 * MiB + 12345).tobytes()
    for kw in (dict(checksum=True), dict(streamed=True), dict(checksum=True, streamed=True), dict(ldm=True, checksum=True)):
        _check(gpu_dec, O.ref_zstd_compress_opts(x, 3, **kw), x)
    comp = O.ref_zstd_compress(x, 3, piece=MiB).tobytes()                      # 21 frames
    frames, n, total = gpu_dec.scan(comp)
    assert n == 21 and total == len(x)
    _check(gpu_dec, comp, x)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [1, 3, 12, 19])
def test_gpu_own_encoder_roundtrip(pkg, O, gpu_dec, level):
    x = _corpus(O, "silesia-like", 100 * MiB + 4321)
    enc = pkg.ZstdEncoder(device=0, level=level)
    try:
        comp = enc.code(x)
    finally:
        enc.close()
    assert O.ref_zstd_decompress(comp, x.size).tobytes() == x.tobytes()        # the reference decoder agrees on the stream first
    frames, n, total = gpu_dec.scan(comp)
    assert total == x.size
    _check(gpu_dec, comp, x.tobytes())


@pytest.mark.gpu
def test_gpu_damaged_streams_are_refused(pkg, O, gpu_dec):
    x = _corpus(O, "text-zipf", 3 * MiB).tobytes()
    comp = bytearray(O.ref_zstd_compress_opts(x, 3, checksum=True).tobytes())
    rng = np.random.default_rng(5)
    refused = 0
    for _ in range(200):
        bad = bytearray(comp)
        pos = int(rng.integers(0, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            out = gpu_dec.code(bytes(bad), capacity=len(x) + 64)
            assert out.tobytes() == x
        except pkg.GpuCodecError:
            refused += 1
    assert refused >= 190
    _check(gpu_dec, bytes(comp), x)                                            # the context still works afterwards


def _serial_retry(pkg, O, lib_kw, monkeypatch, n):
    """A batch whose overlapped execution kernel reports a frame as damaged (it gives up when the entropy kernels' blocks do not arrive: serialised
    launches under a profiler, a device shared with another process) is decoded once more with the kernels one after the other before the stream is
    called damaged.  The hook makes the first pass over every batch fail: sound streams of every shape must still come out whole (frames of several
    blocks, frames of one block, small batches -- the retry starts again at the batch's first output byte), a damaged one must still be refused."""
    monkeypatch.setenv("GC_ZD_FAIL_FIRST", "1")
    x = _corpus(O, "silesia-like", n).tobytes()
    for batch_kib in (None, "256"):
        if batch_kib: monkeypatch.setenv("GC_ZD_BATCH_KIB", batch_kib)
        dec = pkg.ZstdDecoder(**lib_kw)
        try:
            _check(dec, O.ref_zstd_compress(x, 3).tobytes(), x)
            _check(dec, O.ref_zstd_compress(x, 3, piece=100_000).tobytes(), x)
            _check(dec, O.ref_zstd_compress_opts(x, 5, checksum=True, streamed=True).tobytes(), x)
            bad = bytearray(O.ref_zstd_compress_opts(x, 3, checksum=True).tobytes()); bad[len(bad) // 2] ^= 0x10
            with pytest.raises(pkg.GpuCodecError):
                dec.code(bytes(bad), capacity=len(x) + 64)
        finally:
            dec.close()


def test_emu_serial_retry_of_a_batch(pkg, O, emu_lib_path, monkeypatch):
    _serial_retry(pkg, O, dict(lib_path=emu_lib_path), monkeypatch, 700_000)


@pytest.mark.gpu
def test_gpu_serial_retry_of_a_batch(pkg, O, gpu_dec, gpu_hooks_kw, monkeypatch):
    _serial_retry(pkg, O, gpu_hooks_kw, monkeypatch, 5 * MiB + 11)


def test_emu_decoder_selfcheck(emu_dec):
    """every context decodes a built-in frame of the reference's encoder through the six-blocks-per-wave sequences kernel before it trusts it"""
    assert emu_dec.selfcheck() == 1


@pytest.mark.gpu
def test_gpu_decoder_selfcheck(pkg, graft):
    """... and on the device the check must come back right: -1 would mean the lane-to-lane moves of that kernel were miscompiled again (profiles/r02_dpp_combine.md)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    d = pkg.ZstdDecoder(device=0)
    try:
        assert d.selfcheck() == 1
    finally:
        d.close()
