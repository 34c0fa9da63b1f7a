"""The C host scheduler gc_multi (csrc/gc_multi.hip) through the C ABI: one host buffer range-split into pieces over several GPU
contexts (here: the emulated machine reporting two devices, two contexts each), compressed pieces concatenated in order.
zstd / brotli: the bytes equal one whole-buffer call (pieces are multiples of the independence grain); FLZMA2: pieces are coded
with NO_END_MARK and exactly one end marker closes the stream.  Every stream decodes under the reference decoders.
GPU (-m gpu): the same on the real device(s), at sizes that take several 64 MiB pieces."""
import ctypes as C
import os

import numpy as np
import pytest

BLK = 128 * 1024


@pytest.fixture()
def two_devices(monkeypatch):
    monkeypatch.setenv("HIPEMU_DEVICES", "2")


def test_exports_declared_in_header_exist(pkg, emu_lib_path):
    lib = C.CDLL(emu_lib_path)
    for sym in pkg.EXPORTS:
        assert hasattr(lib, sym), sym


def test_grain_and_piece_sizes(pkg, emu_lib_path):
    lib = pkg.load_library(emu_lib_path)
    assert lib.gc_codec_grain(pkg.CODEC_ZSTD, 1) == 64 * BLK and lib.gc_codec_grain(pkg.CODEC_ZSTD, 3) == 64 * BLK      # (the windowed finder at every level)
    assert lib.gc_codec_grain(pkg.CODEC_FLZMA2, 1) == 64 * BLK and lib.gc_codec_grain(pkg.CODEC_FLZMA2, 5) == 128 * BLK     # (FLZMA2: the windowed finder at every level; level 5: overlapping frames in groups of 16 MiB)
    for q in range(0, 12):
        g = lib.gc_codec_grain(pkg.CODEC_BROTLI, q)
        assert g == max(q, 1) * 8 * BLK                      # the brotli-mt chunk (C/zstdmt/brotli-mt_compress.c:115-118)
        p = lib.gc_multi_piece_bytes(pkg.CODEC_BROTLI, q)
        assert p % g == 0 and 0 < p <= 64 << 20
    assert lib.gc_multi_piece_bytes(pkg.CODEC_ZSTD, 3) == 64 << 20 and lib.gc_multi_piece_bytes(pkg.CODEC_FLZMA2, 5) == 256 << 20


@pytest.mark.parametrize("codec,level,piece,n", [("zstd", 1, BLK, 5 * BLK + 777), ("zstd", 3, 2 * BLK, 5 * BLK + 5),
                                                 ("brotli", 1, 0, 2 * 8 * BLK + 999), ("flzma2", 1, 2 * BLK, 5 * BLK + 31)])
def test_two_emulated_devices(pkg, O, emu_lib_path, two_devices, monkeypatch, codec, level, piece, n):
    monkeypatch.setenv("GC_FRAME_BLOCKS", "2")               # test hook: frames of two blocks, so that several frames fit an emulator-sized input
    x = O.corpus("text-zipf", n)
    m = pkg.MultiEncoder(codec, level, lib_path=emu_lib_path)
    assert m.workers() == 4
    y = m.code(x, piece_bytes=piece)
    m.close()
    if codec == "zstd":
        e = pkg.ZstdEncoder(level=level, lib_path=emu_lib_path); whole = e.code(x); e.close()
        assert np.array_equal(y, whole)                      # pieces = whole frames: identical to the whole-buffer call
        assert np.array_equal(O.ref_zstd_decompress(y, n), x)
    elif codec == "brotli":
        e = pkg.BrotliEncoder(level=level, lib_path=emu_lib_path); whole = e.code(x); e.close()
        assert abs(int(y.size) - int(whole.size)) <= 64 * 3  # pieces = whole brotli-mt chunks (block-local finder at quality <= 2: see above)
        assert np.array_equal(O.ref_brotlimt_decompress(y, n, 2), x)
    else:
        e = pkg.Flzma2Encoder(level=level, lib_path=emu_lib_path); prop = e.coder_props()[0]; e.close()
        assert y[-1] == 0 and np.array_equal(O.ref_lzma2_decode(y, n, prop), x)
        assert np.array_equal(O.port_lzma2_decode(y, n, prop), x)


def test_flzma2_no_end_mark_passes_through(pkg, O, emu_lib_path, two_devices):
    x = O.corpus("silesia-like", 3 * BLK)
    m = pkg.MultiEncoder("flzma2", 1, lib_path=emu_lib_path)
    a = m.code(x, piece_bytes=BLK)
    b = m.code(x, flags=1, piece_bytes=BLK)                  # GC_FLZMA2_NO_END_MARK: the caller appends the marker
    m.close()
    assert np.array_equal(a[:-1], b) and a[-1] == 0


def test_destination_too_small_is_reported(pkg, O, emu_lib_path, two_devices):
    x = O.corpus("text-zipf", 4 * BLK)
    lib = pkg.load_library(emu_lib_path)
    m = C.c_void_p()
    assert lib.gc_multi_create(C.byref(m), None, 0, 2) == 0
    out = np.empty(1000, dtype=np.uint8); n = C.c_size_t(0)
    rc = lib.gc_multi_compress_host(m, pkg.CODEC_ZSTD, x.ctypes.data, x.size, out.ctypes.data, out.size, 1, 0, BLK, C.byref(n))
    assert rc == -4 and b"too small" in lib.gc_multi_last_error(m)
    lib.gc_multi_destroy(m)


@pytest.mark.gpu
@pytest.mark.parametrize("codec,level,kind,n", [("zstd", 3, "text-zipf", 200_000_000), ("flzma2", 5, "silesia-like", 300_000_000),
                                                ("brotli", 6, "web-text", 150_000_000)])
def test_gpu_multi_pieces(pkg, O, graft, codec, level, kind, n):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    x = O.corpus(kind, n)
    m = pkg.MultiEncoder(codec, level)
    y = m.code(x)
    m.close()
    thr = min(os.cpu_count() or 1, 64)
    if codec == "zstd":
        e = pkg.ZstdEncoder(level=level); whole = e.code(x); e.close()
        assert np.array_equal(y, whole)                      # 64 MiB pieces = whole 8 MiB frames
        assert np.array_equal(O.ref_zstd_decompress(y, n), x)
    elif codec == "brotli":
        e = pkg.BrotliEncoder(level=level); whole = e.code(x); e.close()
        assert np.array_equal(y, whole)
        assert np.array_equal(O.ref_brotlimt_decompress(y, n, thr), x)
    else:
        e = pkg.Flzma2Encoder(level=level); prop = e.coder_props()[0]; e.close()
        assert np.array_equal(O.ref_lzma2_decode(y, n, prop), x)


@pytest.mark.gpu
def test_gpu_multi_two_real_devices(pkg, O, graft):
    """The host scheduler over TWO real GPUs (skipped on a one-GPU box: the first multi-GPU box that runs the suite exercises it): pieces dealt to
    four contexts on two devices, the stream equal to the one-context stream and decodable by the reference."""
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    graft.build_hip()
    x = O.corpus("text-zipf", 300_000_000)
    m = pkg.MultiEncoder("zstd", 3)
    assert m.workers() >= 4
    y = m.code(x)
    m.close()
    e = pkg.ZstdEncoder(level=3); whole = e.code(x); e.close()
    assert np.array_equal(y, whole)
    assert np.array_equal(O.ref_zstd_decompress(y, x.size), x)
