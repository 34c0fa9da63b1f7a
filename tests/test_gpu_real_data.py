"""GPU size bars on REAL data (-m gpu; round 3).  The stand-in corpora are generators; these are real bytes that the image itself holds, the same
on the build container and on the GPU box (7-zip-zstd_amd/corpus: real-src = C / C++ headers + Python sources, real-bin = the shared objects of
/opt/rocm/lib incl. their gfx code objects, real-py = the standard library's .py + .pyc): the three BASELINE codecs at their BASELINE levels against
the reference ENCODER (oracle/_ref) on the same bytes, band 2 %, each stream decoded by the reference decoder."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MiB = 1024 * 1024
THR = min(os.cpu_count() or 1, 64)
CASES = [("real-src", 64 * MiB), ("real-bin", 211_900_000), ("real-py", 64 * MiB)]       # (real-py: all there is, 18.9 MB)
# bars that are not met, with the measured figure (filled from tools/gpu_ratio.py on the MI355X; strict=False)
NOT_YET = {("flzma2", "real-src"): "0.998 x the reference in round 4 (1.018 in round 3): inside the band, the entry stays as a guard",
           ("flzma2", "real-bin"): "met in round 5: 1.018 x the reference on all 211.9 MB (model segments over cheap neighbouring blocks, level 5 in overlapping finder frames of 16 MiB groups -- the reference's dictionary at this level; 1.026 in round 4, 1.054 in round 3): the entry stays as a guard",
           ("flzma2", "real-py"): "1.008 x the reference in round 4 (1.017 in round 3): inside the band, the entry stays as a guard",
           ("brotli", "real-src"): "1.0205 x the reference (round 6, 64 MiB, run final4: the ring-aware parse W6r; 1.041 before it, 1.058 in round 5, 1.084 in round 3).  What is left: 32 meta-blocks' worth of prefix-code descriptions where the reference writes one meta-block with block types (1 % of the stream), block splitting, the static dictionary",
           ("brotli", "real-bin"): "met in round 6: 1.012 x the reference (W6r: the parse tries its own last four distances first, as the reference's hasher does; 1.073 before it, 1.085 in round 5, 1.106 in round 3); the entry stays as a guard",
           ("brotli", "real-py"): "1.015 x the reference in round 4 (1.024 in round 3): inside the band, the entry stays as a guard"}


@pytest.fixture(scope="module")
def gpu(pkg, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    return pkg


def _corpus(O, kind, n):
    x = O.corpus(kind, n)
    if x.size < (1 << 20):
        pytest.skip("the image holds no %s data" % kind)
    return x


def _bar(codec, kind, ours, ref):
    if ours > 1.02 * ref and (codec, kind) in NOT_YET:
        pytest.xfail("known gap: " + NOT_YET[(codec, kind)])
    assert ours <= 1.02 * ref, (codec, kind, ours, ref, round(ours / ref, 4))


@pytest.mark.parametrize("kind,n", CASES)
def test_zstd_level3_real_data(O, gpu, kind, n):
    if O.ref("zstd") is None:
        pytest.skip("oracle/_ref did not travel")
    x = _corpus(O, kind, min(n, 128 * MiB))
    e = gpu.ZstdEncoder(level=3); c = e.code(x); e.close()
    assert np.array_equal(O.ref_zstd_decompress(c, x.size), x)
    _bar("zstd", kind, len(c), len(O.ref_zstd_compress(x, 3)))


# The levels between the BASELINE ones on real bytes (round 6; the review of round 5 found zstd 9 on shared objects at 1.060 with no test looking): the lazy range (9, 12), C4's
# level (19) and FLZMA2's first ultra level, 32 MiB each, level L against the reference's level L.  Figures: MI355X, run s2 of round 6 (tools/gpu_sizes.py).
LEVEL_CASES = [("zstd", 5, "real-src"), ("zstd", 5, "real-bin"), ("zstd", 6, "real-src"), ("zstd", 9, "real-src"), ("zstd", 9, "real-bin"), ("zstd", 12, "real-src"), ("zstd", 12, "real-bin"), ("zstd", 19, "real-src"), ("zstd", 19, "real-bin"), ("flzma2", 7, "real-bin"),
               ("brotli", 5, "real-bin"), ("brotli", 7, "real-bin"), ("brotli", 5, "real-src"), ("brotli", 7, "real-src"), ("brotli", 9, "real-bin"), ("brotli", 9, "real-src")]      # (brotli 5 / 7: the ring-aware parse W6r, run final4)
NOT_YET_LEVELS = {("zstd", 19, "real-src"): "1.077 x btultra2 on real sources (round 6: 16 MiB finder frames; 1.095 in round 5, 1.152 in round 4): one merged record per position against the binary tree's list of matches, static prices, no block splitter",
                  ("zstd", 19, "real-bin"): "1.044 x btultra2 on shared objects (round 6; 1.043 in round 5)",
                  ("flzma2", 7, "real-bin"): "1.041 x the reference's level 7 on shared objects (round 6, first measurement; real sources 1.019)",
                  ("brotli", 5, "real-src"): "1.038 x the reference's quality 5 on 32 MiB of real sources (round 6, run final4; shared objects 1.001)",
                  ("brotli", 7, "real-src"): "met in round 6: 1.004 x the reference's quality 7 on 32 MiB of real sources (quality 7 on the price-based parse with W7L; 1.060 with the greedy parse); a guard",
                  ("brotli", 9, "real-bin"): "met in round 6: 0.980 x the reference's quality 9 on shared objects (qualities 8-11 run W7L, the lane-per-window parse with the last distances at every node; 1.023 with W7); a guard",
                  ("brotli", 9, "real-src"): "1.038 x the reference's quality 9 on 32 MiB of real sources (round 6, W7L; 1.092 with W7)"}


@pytest.mark.parametrize("codec,level,kind", LEVEL_CASES)
def test_levels_between_the_baseline_ones_on_real_data(O, gpu, codec, level, kind):
    if O.ref(codec) is None:
        pytest.skip("oracle/_ref did not travel")
    x = _corpus(O, kind, 32 * MiB)
    if codec == "zstd":
        e = gpu.ZstdEncoder(level=level); c = e.code(x); e.close()
        assert np.array_equal(O.ref_zstd_decompress(c, x.size), x)
        ref = len(O.ref_zstd_compress(x, level))
    elif codec == "brotli":
        e = gpu.BrotliEncoder(level=level); c = e.code(x); e.close()
        assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, THR), x)
        ref = len(O.ref_brotlimt_compress(x, level, THR))
    else:
        e = gpu.Flzma2Encoder(level=level); c = e.code(x); prop = e.coder_props()[0]; e.close()
        assert np.array_equal(O.ref_lzma2_decode(c, x.size, prop), x)
        ref = len(O.ref_fl2_compress(x, level, threads=THR)[0])
    if len(c) > 1.02 * ref and (codec, level, kind) in NOT_YET_LEVELS:
        pytest.xfail("known gap: " + NOT_YET_LEVELS[(codec, level, kind)])
    assert len(c) <= 1.02 * ref, (codec, level, kind, len(c), ref, round(len(c) / ref, 4))


@pytest.mark.parametrize("kind,n", CASES)
def test_flzma2_level5_real_data(O, gpu, kind, n):
    if O.ref("flzma2") is None:
        pytest.skip("oracle/_ref did not travel")
    x = _corpus(O, kind, n)
    e = gpu.Flzma2Encoder(level=5); c = e.code(x); prop = e.coder_props()[0]; e.close()
    assert np.array_equal(O.ref_lzma2_decode(c, x.size, prop), x)
    ref, _ = O.ref_fl2_compress(x, 5, threads=THR)
    _bar("flzma2", kind, len(c), len(ref))


@pytest.mark.parametrize("kind,n", CASES)
def test_brotli_q6_real_data(O, gpu, kind, n):
    if O.ref("brotli") is None:
        pytest.skip("oracle/_ref did not travel")
    x = _corpus(O, kind, min(n, 64 * MiB))
    e = gpu.BrotliEncoder(level=6); c = e.code(x); e.close()
    assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, THR), x)
    _bar("brotli", kind, len(c), len(O.ref_brotlimt_compress(x, 6, THR)))
