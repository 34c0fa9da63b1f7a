"""CRC-32 on the device against zlib's (the same CRC-32/ISO-HDLC as C/7zCrc.c): ragged sizes around the 4 KiB slice and
1 MiB chunk boundaries, and the linearity the kernel rests on."""
import zlib

import numpy as np
import pytest

MiB = 1 << 20


@pytest.mark.parametrize("n", [0, 1, 4095, 4096, 4097, MiB - 1, MiB, MiB + 1, 2 * MiB + 12345])
def test_crc32_emulator(pkg, O, emu_lib_path, n):
    x = O.corpus("silesia-like", n) if n else np.empty(0, dtype=np.uint8)
    assert pkg.crc32_device(x.ctypes.data if n else 0, n, emu_lib_path) == (zlib.crc32(x.tobytes()) & 0xFFFFFFFF)


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
        assert pkg.crc32_device(d.data_ptr(), n) == (zlib.crc32(x.tobytes()) & 0xFFFFFFFF)
