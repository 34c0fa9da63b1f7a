"""CRC-32 on the device against the reference's own (C/7zCrc.c CrcCalc, compiled into oracle/_ref/libbra_ref.so) and against
zlib's (the same CRC-32/ISO-HDLC; the check that still runs where oracle/_ref is absent): ragged sizes around the 4 KiB slice and 1 MiB chunk
boundaries."""
import zlib

import numpy as np
import pytest

MiB = 1 << 20


@pytest.mark.parametrize("n", [0, 1, 4095, 4096, 4097, MiB - 1, MiB, MiB + 1, 2 * MiB + 12345])
def _want(O, x):
    z = zlib.crc32(x.tobytes()) & 0xFFFFFFFF
    lib = O.ref("bra")
    if lib is not None and hasattr(lib, "ref_crc32"):
        assert O.ref_crc32(x) == z                 # the reference's table-driven CRC and zlib's agree on every input used here
    return z


def test_reference_crc_is_the_checker(O):
    """the checker really is C/7zCrc.c when oracle/_ref is present (here and on the GPU box)"""
    lib = O.ref("bra")
    if lib is None:
        pytest.skip("oracle/_ref not built")
    assert hasattr(lib, "ref_crc32")
    assert O.ref_crc32(np.frombuffer(b"123456789", dtype=np.uint8)) == 0xCBF43926


@pytest.mark.parametrize("n", [0, 1, 4095, 4096, 4097, MiB - 1, MiB, MiB + 1, 2 * MiB + 12345])
def test_crc32_emulator(pkg, O, emu_lib_path, n):
    x = O.corpus("silesia-like", n) if n else np.empty(0, dtype=np.uint8)
    assert pkg.crc32_device(x.ctypes.data if n else 0, n, emu_lib_path) == _want(O, x)


@pytest.mark.gpu
def test_gpu_crc32_full_size(pkg, O, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    for n in (0, 5, 100_000_000, 3 * MiB):
        x = O.corpus("text-zipf", n) if n else np.empty(0, dtype=np.uint8)
        d = torch.from_numpy(x).cuda() if n else torch.empty(1, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        assert pkg.crc32_device(d.data_ptr(), n) == _want(O, x)
