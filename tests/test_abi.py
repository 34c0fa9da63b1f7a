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


def test_quad_permute_moves_of_the_sequences_kernel_stay_unfolded(graft, tmp_path):
    """gc_zstd_dec_seqv_kernel passes values between the four lanes of a block with quad-permute DPP moves.  Folded into their users by the compiler's
    DPP combiner (v_subrev_u32_dpp ... quad_perm:[1,1,1,1]) they read the neighbour lane as 0 on the MI355X (profiles/r02_dpp_combine.md), so the source
    keeps every move an instruction of its own; this compiles the file for gfx950 (no GPU needed) and looks at the kernel's instructions."""
    import re, shutil, subprocess
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    if not hipcc:
        pytest.skip("no hipcc")
    src = os.path.join(graft.CSRC, "gc_zstd_dec.hip")
    out = str(tmp_path / "gc_zstd_dec.s")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + graft.CSRC, "-I" + os.path.join(graft.ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    m = re.search(r"^gc_zstd_dec_seqv_kernel:.*?s_endpgm", text, re.S | re.M)
    assert m, "kernel not found in the assembly"
    dpp = re.findall(r"\b(v_\w+_dpp)\b", m.group(0))
    assert dpp and set(dpp) == {"v_mov_b32_dpp"}, sorted(set(dpp))
    assert len(dpp) == 9                                           # 2 x 3 bit counts + 3 values per sequence step


def test_shipped_library_reads_no_environment(pkg, graft):
    """The GC_* hooks are compiled into the test build only: the shipped library imports no getenv and says so."""
    import subprocess
    graft.build_hip()
    syms = subprocess.run(["nm", "-D", "--undefined-only", graft.LIB], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in syms
    hooks = subprocess.run(["nm", "-D", "--undefined-only", graft.LIB_HOOKS], capture_output=True, text=True, check=True).stdout
    assert "getenv" in hooks
    import ctypes
    assert ctypes.CDLL(graft.LIB).gc_test_hooks_enabled() == 0
    assert ctypes.CDLL(graft.LIB_HOOKS).gc_test_hooks_enabled() == 1
