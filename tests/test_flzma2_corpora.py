"""FLZMA2 path, second file (the CPU suite spreads test FILES over worker processes: the corpus sweeps of tests/test_flzma2.py live here so that no
single file is the critical path): every corpus at levels 1 / 5 / 9 under the emulator, decoded by the reference's LZMA2 decoder and its restatement;
size against the reference encoder."""
import numpy as np
import pytest

BLK = 128 * 1024


def _roundtrip(O, enc, x):
    c = enc.code(x)
    prop = enc.coder_props()[0]
    assert np.array_equal(O.port_lzma2_decode(c, x.size, prop), x)
    if O.ref("flzma2") is not None:
        assert np.array_equal(O.ref_lzma2_decode(c, x.size, prop), x)
    return c


@pytest.fixture(scope="module")
def emu_fl2(pkg, emu_lib_path):
    encs = {lv: pkg.Flzma2Encoder(lib_path=emu_lib_path, level=lv) for lv in (1, 5, 9)}
    yield encs
    for e in encs.values():
        e.close()


@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like", "web-text", "random", "zeros"])
def test_emu_corpora_all_levels(O, emu_fl2, kind):
    x = O.corpus(kind, BLK + 70_000)
    sizes = [len(_roundtrip(O, emu_fl2[lv], x)) for lv in (1, 5, 9)]
    if kind == "random":
        assert sizes[1] <= x.size + 3 * (x.size // 4096 + 2) + 1       # stored 4 KiB chunks: 3-byte headers only
    if kind in ("text-zipf", "web-text"):
        assert sizes[2] <= sizes[0]                                    # larger chunks = fewer state resets


def test_emu_ratio_band_vs_reference(O, emu_fl2):
    """Size against the reference encoder at level 5 (recorded, and bounded so that regressions show)."""
    if O.ref("flzma2") is None:
        pytest.skip("oracle/_ref not built")
    x = O.corpus("text-zipf", 4 * BLK)
    ours = len(emu_fl2[5].code(x))
    ref, _ = O.ref_fl2_compress(x, 5)
    assert ours <= 1.06 * len(ref), (ours, len(ref))      # 1.033 with the far + short pass and the price-based parse


