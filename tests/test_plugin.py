"""The 7-Zip plugin surface, driven the way the 7-Zip host drives a codec module (tests/host/plugin_host.cpp).
CPU: the plugin layer linked against the emulator build of the kernels.  GPU (-m gpu): the product module
7-zip-zstd_amd/plugin/lib7zgpucodec.so over libgpucodec.so."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "_build")
BLK = 128 * 1024


def _host(module, *args, env=None):
    return subprocess.run([os.path.join(EMU, "plugin_host"), module] + [str(a) for a in args], capture_output=True, text=True,
                          env=dict(os.environ, **env) if env else None)


@pytest.fixture(scope="module")
def emu_module(emu_lib_path):
    return os.path.join(EMU, "lib7zgpucodec_emu.so")


def test_exports_and_method_listing(emu_module):
    r = _host(emu_module, "list")
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].split()[1:3] == ["4F71101", "ZSTD"]            # id and name of ZstdRegister.cpp:13-17
    assert "enc=1 dec=1" in lines[0] and "clsid=23170F69-40C1-2791" in lines[0]      # ZSTD: encoder and decoder
    assert "enc=1 dec=0" in lines[1] and "enc=1 dec=1" in lines[2]                   # FLZMA2: the host's LZMA2 decoder; BROTLI: encoder and decoder (round 6)
    assert lines[1].split()[1:3] == ["21", "FLZMA2"]               # FastLzma2Register.cpp:13-18
    assert lines[2].split()[1:3] == ["4F71102", "BROTLI"]          # BrotliRegister.cpp:13-17
    import ctypes
    lib = ctypes.CDLL(emu_module)
    for sym in ["GetNumberOfMethods", "GetMethodProperty", "CreateEncoder", "CreateDecoder", "CreateObject", "GetModuleProp"]:
        assert hasattr(lib, sym)


@pytest.mark.parametrize("n,how", [(0, "index"), (1, "index"), (2 * BLK + 777, "index"), (BLK, "by-clsid")])
def test_code_through_com_surface(O, emu_module, emu_enc, tmp_path, n, how):
    x = O.corpus("text-zipf", n)
    src, dst, props = tmp_path / "in.bin", tmp_path / "out.zst", tmp_path / "props.bin"
    x.tofile(src)
    args = ["encode", "ZSTD", 3, src, dst, props] + (["by-clsid"] if how == "by-clsid" else [])
    r = _host(emu_module, *args)
    assert r.returncode == 0, r.stderr + r.stdout
    c = np.fromfile(dst, dtype=np.uint8)
    assert np.array_equal(c, emu_enc.code(x))                      # same bytes as the C ABI called directly
    assert np.array_equal(O.port_zstd_decompress(c, n), x)
    if O.ref("zstd") is not None:
        assert np.array_equal(O.ref_zstd_decompress(c, n), x)
    assert props.read_bytes() == bytes([1, 5, 3, 0, 0])           # CProps of ZstdEncoder.h:17-32; decoder accepts 1/3/5 bytes


@pytest.mark.parametrize("n", [0, 5, BLK + 4321])
def test_flzma2_code_through_com_surface(O, emu_module, pkg, emu_lib_path, tmp_path, n):
    x = O.corpus("silesia-like", n)
    src, dst, props = tmp_path / "in.bin", tmp_path / "out.lzma2", tmp_path / "props.bin"
    x.tofile(src)
    r = _host(emu_module, "encode", "FLZMA2", 5, src, dst, props)
    assert r.returncode == 0, r.stderr + r.stdout
    c = np.fromfile(dst, dtype=np.uint8)
    prop = props.read_bytes()
    assert len(prop) == 1                                           # 1-byte dictionary-size code, Lzma2Encoder.cpp:353-364
    assert np.array_equal(O.port_lzma2_decode(c, n, prop[0]), x)
    if O.ref("flzma2") is not None:
        assert np.array_equal(O.ref_lzma2_decode(c, n, prop[0]), x)


@pytest.mark.parametrize("n", [0, 9, BLK + 4321])
def test_brotli_code_through_com_surface(O, emu_module, tmp_path, n):
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref not built")
    x = O.corpus("web-text", n)
    src, dst, props = tmp_path / "in.bin", tmp_path / "out.br", tmp_path / "props.bin"
    x.tofile(src)
    r = _host(emu_module, "encode", "BROTLI", 6, src, dst, props)
    assert r.returncode == 0, r.stderr + r.stdout
    c = np.fromfile(dst, dtype=np.uint8)
    assert props.read_bytes() == bytes([1, 2, 6])                   # BrotliEncoder.h:18-32; BrotliDecoder.cpp:86-96 wants exactly 3
    assert np.array_equal(O.ref_brotlimt_decompress(c, n), x)


@pytest.mark.gpu
def test_product_plugin_flzma2_on_gpu(O, graft, tmp_path):
    graft.build_hip()
    module = graft.build_plugin()
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu"), "_build/plugin_host"], check=True, capture_output=True)
    n = 70 * 1024 * 1024 + 4321                                     # > one 64 MiB Code() piece: two dictionary-reset runs + end marker
    x = O.corpus("silesia-like", n)
    src, dst, props = tmp_path / "in.bin", tmp_path / "out.lzma2", tmp_path / "props.bin"
    x.tofile(src)
    r = _host(module, "encode", "FLZMA2", 5, src, dst, props)
    assert r.returncode == 0, r.stderr + r.stdout
    c = np.fromfile(dst, dtype=np.uint8)
    prop = props.read_bytes()[0]
    dec = O.ref_lzma2_decode(c, n, prop) if O.ref("flzma2") is not None else O.port_lzma2_decode(c, n, prop)
    assert np.array_equal(dec, x)


@pytest.mark.gpu
def test_product_plugin_on_gpu(O, graft, tmp_path):
    graft.build_hip()
    module = graft.build_plugin()
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu"), "_build/plugin_host"], check=True, capture_output=True)
    n = 20 * 1024 * 1024 + 4321
    x = O.corpus("silesia-like", n)
    src, dst = tmp_path / "in.bin", tmp_path / "out.zst"
    x.tofile(src)
    r = _host(module, "encode", "ZSTD", 3, src, dst, "-")
    assert r.returncode == 0, r.stderr + r.stdout
    c = np.fromfile(dst, dtype=np.uint8)
    dec = O.ref_zstd_decompress(c, n) if O.ref("zstd") is not None else O.port_zstd_decompress(c, n)
    assert np.array_equal(dec, x)


@pytest.mark.parametrize("n", [0, 7, 3 * BLK + 99])
def test_brotli_plain_stream_through_com_surface(O, emu_module, tmp_path, n):
    """SetNumberOfThreads(0) on the BROTLI coder = a bare .br stream (no brotli-mt frames), as the reference's BrotliHandler asks of its encoder."""
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref not built")
    x = O.corpus("text-zipf", n)
    src, dst = tmp_path / "in.bin", tmp_path / "out.br"
    x.tofile(src)
    r = _host(emu_module, "encode", "BROTLI", 1, src, dst, "-", "threads0")
    assert r.returncode == 0, r.stderr + r.stdout
    c = np.fromfile(dst, dtype=np.uint8)
    assert np.array_equal(O.ref_brotli_decompress(c, n), x)


def _zstd_streams(O, x):
    """(name, compressed bytes) the decoder object has to handle: one frame, six frames + a skippable one, a frame without a content size"""
    import struct
    b = x.tobytes()
    skip = struct.pack("<II", 0x184D2A50, 5) + b"12345"
    yield "one-frame", O.ref_zstd_compress(b, 3).tobytes()
    yield "frames+skippable", skip + O.ref_zstd_compress(b, 5, piece=max(1, len(b) // 6)).tobytes() + skip
    yield "streamed+checksum", O.ref_zstd_compress_opts(b, 1, checksum=True, streamed=True).tobytes()


@pytest.mark.parametrize("n", [0, 1, 5 * BLK + 333])
def test_zstd_decode_through_com_surface(O, emu_module, tmp_path, n):
    """CreateDecoder / CreateObject(decoder class id) -> SetDecoderProperties2 -> Code(): the content of reference-encoder streams."""
    if O.ref("zstd") is None:
        pytest.skip("oracle/_ref not built")
    x = O.corpus("silesia-like", n)
    props = tmp_path / "props.bin"
    props.write_bytes(bytes([1, 5, 3, 0, 0]))
    for i, (name, comp) in enumerate(_zstd_streams(O, x)):
        src, dst = tmp_path / ("c%d.zst" % i), tmp_path / ("d%d.bin" % i)
        src.write_bytes(comp)
        r = _host(emu_module, "decode", "ZSTD", props if i != 1 else "-", src, dst, *(["by-clsid"] if i == 2 else []))
        assert r.returncode == 0, name + ": " + r.stderr + r.stdout
        assert dst.read_bytes() == x.tobytes(), name
    bad = tmp_path / "bad.zst"
    comp = bytearray(O.ref_zstd_compress_opts(x.tobytes(), 3, checksum=True).tobytes())
    comp[len(comp) // 2] ^= 0x10
    bad.write_bytes(bytes(comp))
    r = _host(emu_module, "decode", "ZSTD", "-", bad, tmp_path / "bad.out")
    assert r.returncode == 15 and "80004005" in r.stderr                    # E_FAIL, like ZstdDecoder.cpp:113-131
    trunc = tmp_path / "trunc.zst"
    trunc.write_bytes(bytes(comp[: len(comp) - 3]))
    r = _host(emu_module, "decode", "ZSTD", "-", trunc, tmp_path / "trunc.out")
    assert r.returncode == 15


def test_zstd_decoder_refuses_a_forged_content_size(emu_module, tmp_path):
    """A 17-byte stream whose frame header states 3.75 GiB of content over one 5-byte raw block: the decoder object must answer "damaged" (E_FAIL) from the
    scan's block count -- a frame regenerates at most 128 KiB per block (zstd_decompress_block.c: blockSizeMax) -- and must not size (and pin) its output
    buffer from the header first."""
    forged = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x80, 0x50]) + (0xF0000000).to_bytes(4, "little") + bytes([0x29, 0x00, 0x00]) + b"hello"
    src = tmp_path / "forged.zst"; src.write_bytes(forged)
    r = _host(emu_module, "decode", "ZSTD", "-", src, tmp_path / "forged.out")
    assert r.returncode == 15 and "80004005" in r.stderr, r.stderr + r.stdout
    honest = bytes([0x28, 0xB5, 0x2F, 0xFD, 0x80, 0x50]) + (5).to_bytes(4, "little") + bytes([0x29, 0x00, 0x00]) + b"hello"
    src.write_bytes(honest)
    r = _host(emu_module, "decode", "ZSTD", "-", src, tmp_path / "honest.out")
    assert r.returncode == 0 and (tmp_path / "honest.out").read_bytes() == b"hello", r.stderr + r.stdout


def test_zstd_encode_then_decode_through_com_surface(O, emu_module, tmp_path):
    x = O.corpus("text-zipf", 3 * BLK + 17)
    src, mid, props, dst = tmp_path / "in.bin", tmp_path / "mid.zst", tmp_path / "props.bin", tmp_path / "out.bin"
    x.tofile(src)
    r = _host(emu_module, "encode", "ZSTD", 3, src, mid, props)
    assert r.returncode == 0, r.stderr + r.stdout
    r = _host(emu_module, "decode", "ZSTDGPU", props, mid, dst)
    assert r.returncode == 0, r.stderr + r.stdout
    assert dst.read_bytes() == x.tobytes()


def _brotli_dictionary_env(O, tmp_path):
    """the stand-in host has no brotli of its own: the plugin reads the RFC's dictionary from the file this variable names"""
    f = tmp_path / "rfc7932_dictionary.bin"
    O.ref_brotli_dictionary().tofile(f)
    return {"GPUCODEC_BROTLI_DICTIONARY": str(f)}


@pytest.mark.parametrize("n", [0, 1, 9 * BLK + 333])
def test_brotli_decode_through_com_surface(O, emu_module, tmp_path, n):
    """CreateDecoder / CreateObject(decoder class id) -> SetDecoderProperties2 (3 bytes, BrotliDecoder.cpp:86-96) -> Code(): brotli-mt streams of the reference encoder
    (several frames; with references to the static dictionary), a bare stream with the size the folder states, damaged and cut streams."""
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref not built")
    env = _brotli_dictionary_env(O, tmp_path)
    x = O.corpus("real-src", n) if n > 1 else O.corpus("text-zipf", n)
    if x.size < n:
        x = O.corpus("text-zipf", n)
    props = tmp_path / "props.bin"
    props.write_bytes(bytes([1, 2, 6]))
    streams = [("q6 x3", O.ref_brotlimt_compress(x, 6, 3)), ("q1", O.ref_brotlimt_compress(x, 1, 2)), ("q11 by class id", O.ref_brotlimt_compress(x, 11, 1))]
    for i, (name, comp) in enumerate(streams):
        src, dst = tmp_path / ("c%d.br" % i), tmp_path / ("d%d.bin" % i)
        src.write_bytes(comp.tobytes())
        r = _host(emu_module, "decode", "BROTLI", props if i != 1 else "-", src, dst, *(["by-clsid"] if i == 2 else []), env=env)
        assert r.returncode == 0, name + ": " + r.stderr + r.stdout
        assert dst.read_bytes() == x.tobytes(), name
    if n > 1:
        bare = tmp_path / "bare.br"; bare.write_bytes(O.ref_brotli_compress(x, 5, 22).tobytes())
        r = _host(emu_module, "decode", "BROTLIGPU", props, bare, tmp_path / "bare.out", "size=%d" % n, env=env)
        assert r.returncode == 0 and (tmp_path / "bare.out").read_bytes() == x.tobytes(), r.stderr + r.stdout
        r = _host(emu_module, "decode", "BROTLI", props, bare, tmp_path / "bare2.out", env=env)          # no size stated: this decoder cannot size the one chunk
        assert r.returncode == 15 and "80004001" in r.stderr                                             # E_NOTIMPL
        # without the dictionary a stream that refers to it is "unsupported", not "damaged"
        r = _host(emu_module, "decode", "BROTLI", props, tmp_path / "c0.br", tmp_path / "nodict.out")
        assert r.returncode == 15 and "80004001" in r.stderr, r.stderr + r.stdout
        comp = bytearray(streams[0][1].tobytes())
        cut = tmp_path / "cut.br"; cut.write_bytes(bytes(comp[: len(comp) - 3]))
        r = _host(emu_module, "decode", "BROTLI", "-", cut, tmp_path / "cut.out", env=env)
        assert r.returncode == 15 and "80004005" in r.stderr                                             # E_FAIL: the stream ends inside a frame
        comp[4] ^= 1                                                                                     # the frame header's "8"
        bad = tmp_path / "bad.br"; bad.write_bytes(bytes(comp))
        r = _host(emu_module, "decode", "BROTLI", "-", bad, tmp_path / "bad.out", env=env)
        assert r.returncode == 15 and "80004005" in r.stderr


def test_brotli_encode_then_decode_through_com_surface(O, emu_module, tmp_path):
    x = O.corpus("silesia-like", 9 * BLK + 17)
    src, mid, props, dst = tmp_path / "in.bin", tmp_path / "mid.br", tmp_path / "props.bin", tmp_path / "out.bin"
    x.tofile(src)
    r = _host(emu_module, "encode", "BROTLI", 1, src, mid, props)
    assert r.returncode == 0, r.stderr + r.stdout
    r = _host(emu_module, "decode", "BROTLIGPU", props, mid, dst)         # this engine's streams never refer to the dictionary: none handed over
    assert r.returncode == 0, r.stderr + r.stdout
    assert dst.read_bytes() == x.tobytes()


@pytest.mark.gpu
def test_product_plugin_brotli_decoder_on_gpu(O, graft, tmp_path):
    graft.build_hip()
    module = graft.build_plugin()
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu"), "_build/plugin_host"], check=True, capture_output=True)
    n = 300 * 1024 * 1024 + 4321                                   # the compressed stream exceeds one 64 MiB read piece of the decoder
    x = O.corpus("web-text", n)
    src, mid, props, dst = tmp_path / "in.bin", tmp_path / "mid.br", tmp_path / "props.bin", tmp_path / "out.bin"
    x.tofile(src)
    r = _host(module, "encode", "BROTLI", 6, src, mid, props)
    assert r.returncode == 0, r.stderr + r.stdout
    r = _host(module, "decode", "BROTLI", props, mid, dst)
    assert r.returncode == 0, r.stderr + r.stdout
    assert np.array_equal(np.fromfile(dst, dtype=np.uint8), x)
    if O.ref("brotli") is not None:
        env = _brotli_dictionary_env(O, tmp_path)
        y = O.corpus("real-src", 40 * 1024 * 1024)
        for q in (1, 6, 11):
            s2, d2 = tmp_path / ("c%d.br" % q), tmp_path / ("d%d.bin" % q)
            s2.write_bytes(O.ref_brotlimt_compress(y[: (8 if q == 11 else 40) * 1024 * 1024], q, min(os.cpu_count() or 1, 64)).tobytes())
            r = _host(module, "decode", "BROTLI", "-", s2, d2, env=env)
            assert r.returncode == 0, str(q) + ": " + r.stderr + r.stdout
            assert d2.read_bytes() == y[: (8 if q == 11 else 40) * 1024 * 1024].tobytes(), q


@pytest.mark.gpu
def test_product_plugin_zstd_decoder_on_gpu(O, graft, tmp_path):
    graft.build_hip()
    module = graft.build_plugin()
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "emu"), "_build/plugin_host"], check=True, capture_output=True)
    n = 150 * 1024 * 1024 + 4321                                   # the compressed stream exceeds one 64 MiB read piece of the decoder
    x = O.corpus("silesia-like", n)
    src, mid, props, dst = tmp_path / "in.bin", tmp_path / "mid.zst", tmp_path / "props.bin", tmp_path / "out.bin"
    x.tofile(src)
    r = _host(module, "encode", "ZSTD", 3, src, mid, props)
    assert r.returncode == 0, r.stderr + r.stdout
    r = _host(module, "decode", "ZSTD", props, mid, dst)
    assert r.returncode == 0, r.stderr + r.stdout
    assert np.array_equal(np.fromfile(dst, dtype=np.uint8), x)
    for i, (name, comp) in enumerate(_zstd_streams(O, x[: 40 * 1024 * 1024])):
        s2, d2 = tmp_path / ("c%d.zst" % i), tmp_path / ("d%d.bin" % i)
        s2.write_bytes(comp)
        r = _host(module, "decode", "ZSTD", "-", s2, d2)
        assert r.returncode == 0, name + ": " + r.stderr + r.stdout
        assert d2.read_bytes() == x[: 40 * 1024 * 1024].tobytes(), name


@pytest.mark.parametrize("codec,level,kind,n", [("ZSTD", 3, "text-zipf", 5 * 2 * BLK + BLK + 777), ("FLZMA2", 5, "silesia-like", 3 * 2 * BLK + 11), ("BROTLI", 1, "web-text", 2 * 8 * BLK + 99),
                                                ("ZSTD", 3, "text-zipf", 0), ("ZSTD", 3, "text-zipf", 2 * BLK)])
def test_read_ahead_while_the_devices_open(O, emu_module, tmp_path, codec, level, kind, n):
    """Code() reads pieces ahead into ordinary memory while the scheduler is still being created in the background (0.2 s of HIP start-up on a real
    machine; here a test hook delays the emulator's), compresses them first and goes on with the pinned buffers: the stream must be the one a run
    without the delay writes (zstd / brotli: the same bytes), whatever falls into the read-ahead -- nothing, a part, or the whole input.  (The tail behind
    the last whole piece has more than one block here: a tail of less than a block is coded by the block-local kernel when it is a call of its own and by the
    windowed finder when it closes a longer call -- two valid streams that differ in a few bytes.)"""
    x = O.corpus(kind, n)
    src = tmp_path / "in.bin"; x.tofile(src)
    small = {"GC_FRAME_BLOCKS": "2", "GC_PLUGIN_PIECE_KIB": str(8 * 128 if codec == "BROTLI" else 2 * 128)}       # pieces of 256 KiB (brotli quality 1: its 1 MiB chunk)
    outs = []
    for tag, env in (("plain", small), ("delayed", dict(small, GC_PLUGIN_WARM_DELAY_MS="400"))):
        dst, props = tmp_path / ("out_" + tag), tmp_path / ("props_" + tag)
        r = _host(emu_module, "encode", codec, level, src, dst, props, env=env)
        assert r.returncode == 0, r.stderr + r.stdout
        outs.append((np.fromfile(dst, dtype=np.uint8), props.read_bytes()))
    (a, pa), (b, pb) = outs
    assert pa == pb
    if codec == "ZSTD":
        assert np.array_equal(a, b)
        assert np.array_equal(O.port_zstd_decompress(b, n), x)
    elif codec == "BROTLI":
        if O.ref("brotli") is None:
            pytest.skip("oracle/_ref not built")
        assert np.array_equal(a, b)
        assert np.array_equal(O.ref_brotlimt_decompress(b, n, 2), x)
    else:
        assert np.array_equal(O.port_lzma2_decode(a, n, pa[0]), x) and np.array_equal(O.port_lzma2_decode(b, n, pb[0]), x)
