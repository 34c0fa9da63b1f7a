"""the mainline LZMA match finders HC4 and BT4 on the device (csrc/gc_lzfind.hip) against the reference's own
C/LzFind.c -- integer parity, value for value: for every position the (length, distance - 1) values GetMatches writes.  The checker is the
reference compiled into oracle/_ref/liblzfind_ref.so (where present) and the committed lists it produced (tests/golden/*_matches.npz)."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAMS = [(1 << 16, 32, 64), (1 << 12, 8, 32), (1 << 20, 1, 273), (70000, 16, 5), (300, 24, 48)]     # history (incl. window wrap-around), cut, nice length


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


@pytest.mark.parametrize("bt", [False, True])
def test_emu_matches_the_golden_lists(pkg, emu_lib_path, bt):
    g = np.load(os.path.join(GOLD, "bt4_matches.npz" if bt else "hc4_matches.npz"))
    for name in ("text", "lz", "sil"):
        hist, cut, nice = (int(v) for v in g[name + "_params"])
        counts, pairs = pkg.lzfind_matches(g[name + "_input"], hist, bt, cut, nice, lib_path=emu_lib_path)
        assert np.array_equal(counts, g[name + "_counts"].astype(np.uint32)) and np.array_equal(pairs, g[name + "_pairs"]), name


@pytest.mark.parametrize("bt", [False, True])
@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like", "zeros", "random"])
def test_emu_matches_the_reference(O, pkg, emu_lib_path, kind, bt):
    if O.ref("lzfind") is None:
        pytest.skip("the compiled reference is absent")
    x = O.corpus(kind, 40000)
    for hist, cut, nice in PARAMS:
        assert _same(O.ref_lzfind_matches(x, hist, bt, 4, cut, nice), pkg.lzfind_matches(x, hist, bt, cut, nice, lib_path=emu_lib_path)), (kind, hist, cut, nice)
    for n in range(0, 10):                                         # inputs too short for a 4-byte hash
        assert _same(O.ref_lzfind_matches(x[:n], 1 << 16, bt, 4, 32, 64), pkg.lzfind_matches(x[:n], 1 << 16, bt, 32, 64, lib_path=emu_lib_path)), n


def test_emu_list_longer_than_the_stride_is_reported(pkg, emu_lib_path):
    x = np.frombuffer(b"abcdefgh" * 64 + b"abcdeXgh" * 8 + b"abcdefgh" * 8, dtype=np.uint8).copy()
    counts = np.zeros(x.size, dtype=np.uint32); pairs = np.zeros(x.size * 4, dtype=np.uint32)
    with pytest.raises(pkg.GpuCodecError):
        pkg.lzfind_get_matches_device(x.ctypes.data, x.size, counts.ctypes.data, pairs.ctypes.data, 4, 1 << 16, True, 32, 64, lib_path=emu_lib_path)
    assert counts.max() <= 4


@pytest.mark.gpu
@pytest.mark.parametrize("bt", [False, True])
def test_gpu_matches_the_golden_lists(pkg, gpu_enc, bt):
    g = np.load(os.path.join(GOLD, "bt4_matches.npz" if bt else "hc4_matches.npz"))
    for name in ("text", "lz", "sil"):
        hist, cut, nice = (int(v) for v in g[name + "_params"])
        counts, pairs = pkg.lzfind_matches(g[name + "_input"], hist, bt, cut, nice, device=0)
        assert np.array_equal(counts, g[name + "_counts"].astype(np.uint32)) and np.array_equal(pairs, g[name + "_pairs"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("bt", [False, True])
@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like", "zeros", "random"])
def test_gpu_matches_the_reference(O, pkg, gpu_enc, kind, bt):
    if O.ref("lzfind") is None:
        pytest.skip("the compiled reference is absent")
    x = O.corpus(kind, 3_000_000 if kind != "zeros" else 300_000)
    for hist, cut, nice in PARAMS:
        assert _same(O.ref_lzfind_matches(x, hist, bt, 4, cut, nice), pkg.lzfind_matches(x, hist, bt, cut, nice, device=0)), (kind, hist, cut, nice)


@pytest.mark.gpu
def test_gpu_bytes_equal_emulator_bytes(O, pkg, gpu_enc, emu_lib_path):
    x = O.corpus("lz-7zip", 50000)
    for bt in (False, True):
        assert _same(pkg.lzfind_matches(x, 1 << 14, bt, 16, 48, device=0), pkg.lzfind_matches(x, 1 << 14, bt, 16, 48, lib_path=emu_lib_path))
