"""SURVEY.md 8 f4 as designed: the CRC-32 of the raw input and a 7-Zip pre-filter applied on the device INSIDE the compress call (gc_codec_compress_host_pre,
include/gpucodec.h gc_pre) -- one transfer for the reader's CRC (C/7zCrc.c), the filter coder (C/Bra86.c, C/Bra.c, C/Delta.c through CFilterCoder) and the
coder.  The stream must be that of the reference's converter output: decoded by the reference's decoder and converted back by the reference's converter it
gives the input; the CRC is the container's CrcCalc.  CPU: emulator build; -m gpu: the product library."""
import zlib

import numpy as np
import pytest

from test_bra import KID, _code_like, _x86_like


def _check(O, enc, x, flt, name, pc=0, delta=1):
    c, crc, done = enc.code_pre(x, filter=flt, pc=pc, delta=delta)
    assert crc == (zlib.crc32(x.tobytes()) & 0xFFFFFFFF), name
    y = O.ref_zstd_decompress(c, x.size)                             # the filtered bytes, as the reference's decoder sees them
    if flt == enc.FILTER_X86:
        want, rdone, _ = O.ref_bra_x86_convert(x, pc, True, 0)
    elif flt == enc.FILTER_DELTA:
        want, rdone = O.ref_delta_convert(x, delta, True); rdone = x.size
    elif flt:
        want, rdone = O.ref_bra_convert(flt, x, pc, True)
    else:
        want, rdone = x, 0
    assert np.array_equal(y, want), name
    if flt:
        assert done == rdone, (name, done, rdone)


def _run(O, enc):
    if O.ref("zstd") is None or O.ref("bra") is None:
        pytest.skip("oracle/_ref not built")
    enc.set_level(3)
    _check(O, enc, _x86_like(300_007, 3), enc.FILTER_X86, "x86")
    _check(O, enc, _code_like("ARM64", 262_144 + 333, 5)[: 262_144 + 333].copy(), KID["ARM64"], "arm64", pc=0x1000)
    _check(O, enc, O.corpus("silesia-like", 200_001), enc.FILTER_DELTA, "delta", delta=4)
    _check(O, enc, O.corpus("text-zipf", 150_000), 0, "crc only")
    c, crc, done = enc.code_pre(np.empty(0, dtype=np.uint8), filter=enc.FILTER_X86)
    assert crc == 0 and done == 0 and O.ref_zstd_decompress(c, 0).size == 0


def test_pre_filter_and_crc_inside_the_compress_call_emulator(O, emu_enc):
    _run(O, emu_enc)


@pytest.mark.gpu
def test_gpu_pre_filter_and_crc_inside_the_compress_call(O, gpu_enc):
    _run(O, gpu_enc)
    x = _x86_like(40_000_000, 11)                                   # whole 8 MiB frames: the windowed finder behind the converter
    _check(O, gpu_enc, x, gpu_enc.FILTER_X86, "x86 40 MB")
