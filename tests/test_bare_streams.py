"""Options for the bare-file handlers: a zstd stream with a seek table (skippable frame listing every frame's sizes, zstd
seekable format) and a plain brotli stream without brotli-mt framing (what the reference writes with threads == 0,
C/zstdmt/brotli-mt_compress.c:462-466).  Both must decode under the reference's plain decoders."""
import struct

import numpy as np
import pytest

BLK = 128 * 1024


def _check_seek_table(O, x, c, frame_bytes):
    n = x.size
    assert np.array_equal(O.ref_zstd_decompress(c, n), x)                 # every zstd decoder skips the skippable frame
    assert np.array_equal(O.port_zstd_decompress(c, n), x)
    b = c.tobytes()
    assert struct.unpack("<I", b[-4:])[0] == 0x8F92EAB1 and b[-5] == 0
    nf = struct.unpack("<I", b[-9:-5])[0]
    assert nf == max(1, -(-n // frame_bytes))
    t0 = len(b) - 9 - 8 * nf - 8
    magic, size = struct.unpack("<II", b[t0:t0 + 8])
    assert magic == 0x184D2A5E and size == 8 * nf + 9
    off, dec = 0, 0
    for f in range(nf):
        cs, ds = struct.unpack("<II", b[t0 + 8 + 8 * f: t0 + 16 + 8 * f])
        assert b[off:off + 4] == b"\x28\xB5\x2F\xFD"                       # every entry points at a frame ...
        piece = np.frombuffer(b[off:off + cs], dtype=np.uint8)
        assert np.array_equal(O.ref_zstd_decompress(piece, ds), x[dec:dec + ds])    # ... that decodes on its own to its part of the input
        off += cs; dec += ds
    assert off == t0 and dec == n


@pytest.mark.parametrize("level,n,frame", [(1, 3 * BLK + 77, 2 * BLK), (3, 5 * BLK + 5, 2 * BLK)])       # (the windowed finder at every level: frames of GC_FRAME_BLOCKS blocks)
def test_zstd_seek_table_emulator(pkg, O, emu_lib_path, monkeypatch, level, n, frame):
    monkeypatch.setenv("GC_FRAME_BLOCKS", "2")
    x = O.corpus("text-zipf", n)
    e = pkg.ZstdEncoder(level=level, lib_path=emu_lib_path)
    plain = e.code(x)
    e.set_option(e.OPT_ZSTD_SEEK_TABLE, 1)
    c = e.code(x)
    e.set_option(e.OPT_ZSTD_SEEK_TABLE, 0)
    assert np.array_equal(e.code(x), plain)
    e.close()
    assert np.array_equal(c[:plain.size], plain)                          # the table is appended, nothing else changes
    _check_seek_table(O, x, c, frame)


@pytest.mark.parametrize("level,n", [(1, 0), (1, 5), (1, 9 * BLK + 123), (3, 4 * BLK + 1)])
def test_brotli_plain_stream_emulator(pkg, O, emu_lib_path, level, n):
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref not built")
    x = O.corpus("silesia-like", n)
    e = pkg.BrotliEncoder(level=level, lib_path=emu_lib_path)
    e.set_option(e.OPT_BROTLI_PLAIN, 1)
    c = e.code(x)
    e.close()
    assert c[:4].tobytes() != b"\x50\x2A\x4D\x18"                         # no brotli-mt frame header
    assert np.array_equal(O.ref_brotli_decompress(c, n), x)               # BrotliDecoderDecompress on ONE stream


@pytest.mark.gpu
def test_gpu_seek_table_and_plain_brotli(pkg, O, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    x = O.corpus("text-zipf", 40_000_000)
    e = pkg.ZstdEncoder(level=3); e.set_option(e.OPT_ZSTD_SEEK_TABLE, 1); c = e.code(x); e.close()
    _check_seek_table(O, x, c, 64 * BLK)
    y = O.corpus("web-text", 40_000_000)
    e = pkg.BrotliEncoder(level=6); e.set_option(e.OPT_BROTLI_PLAIN, 1); c = e.code(y); e.close()
    assert np.array_equal(O.ref_brotli_decompress(c, y.size), y)


def test_brotli_plain_stream_in_pieces_emulator(pkg, O, emu_lib_path, monkeypatch):
    """The host scheduler codes the pieces of ONE plain stream: header in the first piece only, closing meta-block in the last only."""
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref not built")
    monkeypatch.setenv("HIPEMU_DEVICES", "2")
    n = 2 * 8 * BLK + 4321                                                 # three pieces at quality 1 (chunk = 1 MiB)
    x = O.corpus("text-zipf", n)
    m = pkg.MultiEncoder("brotli", 1, lib_path=emu_lib_path)
    c = m.code(x, flags=1)                                                  # GC_BROTLI_PLAIN
    framed = m.code(x)
    m.close()
    assert np.array_equal(O.ref_brotli_decompress(c, n), x)
    assert np.array_equal(O.ref_brotlimt_decompress(framed, n, 2), x)      # (the context's framing comes back after a plain call)
    e = pkg.BrotliEncoder(level=1, lib_path=emu_lib_path); e.set_option(e.OPT_BROTLI_PLAIN, 1); whole = e.code(x); e.close()
    assert abs(int(c.size) - int(whole.size)) <= 64 * 3
