"""The C-ABI library builds for gfx950, loads, and exports every symbol include/gpucodec.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "gpucodec.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gc_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_all_declared_symbols(graft, pkg):
    lib_path = graft.build_hip()
    assert os.path.exists(lib_path)
    lib = ctypes.CDLL(lib_path)
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "libgpucodec.so does not export %s" % n
    assert sorted(pkg.EXPORTS) == names


def test_bound_and_no_fallback_without_gpu(pkg, graft):
    import torch
    lib = pkg.load_library(graft.build_hip())
    assert lib.gc_zstd_compress_bound(0) >= 9
    n = 100_000_000
    assert lib.gc_zstd_compress_bound(n) >= n + (n // (128 * 1024) + 1) * 12
    if not torch.cuda.is_available():
        # the product must fail loudly, never fall back to a CPU codec
        with pytest.raises(pkg.GpuCodecError):
            pkg.ZstdEncoder(device=0)


def test_emulator_library_has_same_surface(emu_lib_path):
    lib = ctypes.CDLL(emu_lib_path)
    for n in _declared_functions():
        assert hasattr(lib, n)
