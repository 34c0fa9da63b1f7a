"""GPU ratio bars (-m gpu): compressed size of the HIP path against the reference ENCODER (oracle/_ref, the reference's own C
sources compiled here) at the same level on the same bytes, with the north-star band of 2 %, plus decode under the reference
decoder -- for the levels and configs that round 1 only measured in profiles/: zstd 5 / 7 / 9 / 12 (the lazy range),
zstd 19 (btultra2, config C4), FLZMA2 level 5 on the Silesia stand-in and three more corpora (config C3), brotli quality 6 on
binary-like corpora, and the per-GPU shares of configs C4 (125 MB at level 19) and C5 (1.25 GB of web-text at quality 6)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MiB = 1024 * 1024
THR = min(os.cpu_count() or 1, 64)


@pytest.fixture(scope="module")
def gpu(pkg, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    return pkg


def _need(O, name):
    if O.ref(name) is None:
        pytest.skip("oracle/_ref did not travel")


# Bars that are NOT met yet are recorded as expected failures with the measured figure (MI355X, runs r03_levels / r03_q1 of round 3 unless a round is
# named), so that the suite stays green and the gap stays visible; strict=False: the day a bar is met the test simply passes.
NOT_YET = {("zstd", 16, "text-zipf"): "met since round 5 (1.003 x btopt in round 6): the entry stays as a guard",
           ("zstd", 19, "text-zipf"): "1.030 x btultra2 (round 6, run s9: the catch-up, 16 MiB finder frames every 8 MiB; 1.046 in round 5, 1.066 in round 4).  What is left is not the candidates (more key classes, 64 links: nothing, DESIGN section 4) but its accurate, adaptive prices (1.024 at 4 MiB, where the window plays no part)",
           ("zstd", 19, "lz-7zip"): "met in round 6 (0.997 x btultra2: the catch-up + 16 MiB finder frames; 1.023 in round 5): the entry stays as a guard",
           ("zstd", 22, "text-zipf"): "1.042 x the reference's level 22 (round 6: 16 MiB finder frames + the catch-up; 1.056 in round 5; btultra2, window 128 MiB)",
           ("zstd", 22, "lz-7zip"): "met in round 6 (1.010: 16 MiB finder frames, 24-bit positions; 1.033 in round 5): the entry stays as a guard",
           ("flzma2", 7, "text-zipf"): "met in round 6 (1.010; 1.025 in round 5): a guard", ("flzma2", 7, "lz-7zip"): "met in round 6 (1.013 with 16 MiB finder frames; 1.041 in round 5): a guard",
           ("flzma2", 7, "silesia-like"): "met since round 5 (1.015 in round 6): a guard",
           ("flzma2", 9, "text-zipf"): "met in round 6 (1.012; 1.024 in round 5): a guard", ("flzma2", 9, "lz-7zip"): "met in round 6 (1.010; 1.034 in round 5): a guard",
           ("flzma2", 9, "silesia-like"): "met since round 5 (1.015 in round 6): a guard",
           ("brotli", 9, "lz-7zip"): "met in round 6 (0.992: the 9 MiB chunk is ONE finder frame since positions have 24 bits; 1.027 before): a guard",
           ("brotli", 11, "text-zipf"): "1.058 x the reference's quality 11 in round 5 (zopfli-style parse, context clustering, block splitting; B1 chooses between one literal tree and a static map of thirteen, one block type per category)",
           ("brotli", 11, "lz-7zip"): "1.071 x the reference's quality 11 in round 5", ("brotli", 11, "web-text"): "1.067 x the reference's quality 11 in round 5"}


def _xfail_if_known(codec, level, kind):
    why = NOT_YET.get((codec, level, kind))
    if why:
        pytest.xfail("known gap: " + why)


@pytest.mark.parametrize("level", [5, 7, 9, 10, 12])
@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip"])
def test_zstd_lazy_levels_within_2_percent(O, gpu, level, kind):
    _need(O, "zstd")
    x = O.corpus(kind, 32 * MiB)
    e = gpu.ZstdEncoder(level=level); c = e.code(x); e.close()
    assert np.array_equal(O.ref_zstd_decompress(c, x.size), x)
    ref = O.ref_zstd_compress(x, level)
    if len(c) > 1.02 * len(ref):
        _xfail_if_known("zstd", level, kind)
    assert len(c) <= 1.02 * len(ref), (level, kind, len(c), len(ref), round(len(c) / len(ref), 4))


@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip"])
def test_zstd_level19_within_2_percent(O, gpu, kind):
    """config C4's codec: btultra2 (clevels.h:47)"""
    _need(O, "zstd")
    x = O.corpus(kind, 32 * MiB)
    e = gpu.ZstdEncoder(level=19); c = e.code(x); e.close()
    assert np.array_equal(O.ref_zstd_decompress(c, x.size), x)
    ref = O.ref_zstd_compress(x, 19, workers=THR)
    if len(c) > 1.02 * len(ref):
        _xfail_if_known("zstd", 19, kind)
    assert len(c) <= 1.02 * len(ref), (kind, len(c), len(ref), round(len(c) / len(ref), 4))


@pytest.mark.parametrize("kind,n", [("silesia-like", 211_900_000), ("text-zipf", 64 * MiB), ("lz-7zip", 64 * MiB), ("web-text", 64 * MiB)])
def test_flzma2_level5_within_2_percent(O, gpu, kind, n):
    """config C3 and the second half of the headline metric: level 5 against FL2_compressMt on the same bytes"""
    _need(O, "flzma2")
    x = O.corpus(kind, n)
    e = gpu.Flzma2Encoder(level=5); c = e.code(x); prop = e.coder_props()[0]; e.close()
    assert np.array_equal(O.ref_lzma2_decode(c, x.size, prop), x)
    ref, _ = O.ref_fl2_compress(x, 5, threads=THR)
    assert len(c) <= 1.02 * len(ref), (kind, len(c), len(ref), round(len(c) / len(ref), 4))


@pytest.mark.parametrize("kind", ["silesia-like", "lz-7zip", "web-text"])
def test_brotli_q6_within_2_percent(O, gpu, kind):
    _need(O, "brotli")
    x = O.corpus(kind, 48 * MiB)
    e = gpu.BrotliEncoder(level=6); c = e.code(x); e.close()
    assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, THR), x)
    ref = O.ref_brotlimt_compress(x, 6, THR)
    assert len(c) <= 1.02 * len(ref), (kind, len(c), len(ref), round(len(c) / len(ref), 4))


# ---- every level the encoders accept has a size bar (round 3): the ends and the strategy changes of each level table
@pytest.mark.parametrize("level", [1, 2, 16, 22])
@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip"])
def test_zstd_other_levels_within_2_percent(O, gpu, level, kind):
    """levels 1-2 = the reference's `fast` (config C1 runs level 1), 16 = btopt, 22 = its last level (clevels.h:25-47)"""
    _need(O, "zstd")
    x = O.corpus(kind, 32 * MiB)
    e = gpu.ZstdEncoder(level=level); c = e.code(x); e.close()
    assert np.array_equal(O.ref_zstd_decompress(c, x.size), x)
    ref = O.ref_zstd_compress(x, level, workers=THR if level >= 16 else 0)
    if len(c) > 1.02 * len(ref):
        _xfail_if_known("zstd", level, kind)
    assert len(c) <= 1.02 * len(ref), (level, kind, len(c), len(ref), round(len(c) / len(ref), 4))


@pytest.mark.parametrize("level", [1, 2, 3, 7, 9])
@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like"])
def test_flzma2_other_levels_within_2_percent(O, gpu, level, kind):
    """the reference's level table for 7-Zip (fl2_compress.c:52-63): FL2_fast at 1-2, FL2_opt at 3-4, FL2_ultra with dictionaries of 16-128 MiB from 5"""
    _need(O, "flzma2")
    x = O.corpus(kind, 32 * MiB)
    e = gpu.Flzma2Encoder(level=level); c = e.code(x); prop = e.coder_props()[0]; e.close()
    assert np.array_equal(O.ref_lzma2_decode(c, x.size, prop), x)
    ref, _ = O.ref_fl2_compress(x, level, threads=THR)
    if len(c) > 1.02 * len(ref):
        _xfail_if_known("flzma2", level, kind)
    assert len(c) <= 1.02 * len(ref), (level, kind, len(c), len(ref), round(len(c) / len(ref), 4))


@pytest.mark.parametrize("level", [1, 2, 4, 9, 11])
@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "web-text"])
def test_brotli_other_qualities_within_2_percent(O, gpu, level, kind):
    _need(O, "brotli")
    x = O.corpus(kind, 32 * MiB)
    e = gpu.BrotliEncoder(level=level); c = e.code(x); e.close()
    assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, THR), x)
    ref = O.ref_brotlimt_compress(x, level, THR)
    if len(c) > 1.02 * len(ref):
        _xfail_if_known("brotli", level, kind)
    assert len(c) <= 1.02 * len(ref), (level, kind, len(c), len(ref), round(len(c) / len(ref), 4))


def test_config_c4_share_round_trip(O, gpu):
    """zstd level 19 on one GPU's share of enwik9 over 8 GPUs (125 MB)"""
    _need(O, "zstd")
    x = O.corpus("text-zipf", 125_000_000)
    e = gpu.ZstdEncoder(level=19); c = e.code(x); e.close()
    assert np.array_equal(O.ref_zstd_decompress(c, x.size), x)


def test_config_c5_share_round_trip(O, gpu):
    """brotli quality 6 on one GPU's share of 10 GB of web-text over 8 GPUs (1.25 GB), through the host scheduler"""
    _need(O, "brotli")
    x = O.corpus("web-text", 1_250_000_000)
    m = gpu.MultiEncoder("brotli", 6); c = m.code(x); m.close()
    assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, THR), x)
