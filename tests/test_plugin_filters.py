"""The plugin's pre-filter objects (BCJGPU, ARM64GPU, ..., DELTAGPU: SURVEY.md 8 f4 behind the 7-Zip codec interface), driven the way 7-Zip's CFilterCoder
drives a filter (tests/host/plugin_host.cpp `filter`: Init, then Filter() on a buffer that is refilled behind the bytes the filter left over).  CPU: the
plugin layer over the emulator build.  The converted stream must equal what the REFERENCE's converter (C/Bra.c, C/Bra86.c, C/Delta.c compiled into
oracle/_ref/libbra_ref.so) makes of the whole input in one call -- the chunking of the filter coder must not show -- and decoding must give the input back."""
import os
import subprocess

import numpy as np
import pytest

from test_bra import KID, _code_like, _x86_like

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "_build")
NAMES = {"ARM64": "ARM64GPU", "ARM": "ARMGPU", "ARMT": "ARMTGPU", "PPC": "PPCGPU", "SPARC": "SPARCGPU", "IA64": "IA64GPU", "RISCV": "RISCVGPU"}


def _filter(module, name, how, prop, src, dst, buf):
    r = subprocess.run([os.path.join(EMU, "plugin_host"), module, "filter", name, how, str(prop), str(src), str(dst), str(buf)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    return np.fromfile(dst, dtype=np.uint8)


@pytest.fixture(scope="module")
def emu_module(emu_lib_path):
    return os.path.join(EMU, "lib7zgpucodec_emu.so")


def test_filters_are_listed_as_filters(emu_module):
    r = subprocess.run([os.path.join(EMU, "plugin_host"), emu_module, "list"], capture_output=True, text=True)
    assert r.returncode == 0
    rows = {l.split()[2]: l.split() for l in r.stdout.strip().splitlines()}
    for name, mid in (("BCJGPU", "3030103"), ("PPCGPU", "3030205"), ("IA64GPU", "3030401"), ("ARMGPU", "3030501"), ("ARMTGPU", "3030701"), ("SPARCGPU", "3030805"),
                      ("ARM64GPU", "A"), ("RISCVGPU", "B"), ("DELTAGPU", "3")):            # ids of BcjRegister.cpp:12-15, BranchRegister.cpp:33-57, DeltaFilter.cpp:121-124
        assert rows[name][1] == mid and "enc=1" in rows[name] and "dec=1" in rows[name]


@pytest.mark.parametrize("kind", ["ARM64", "ARM", "ARMT", "PPC", "SPARC", "IA64", "RISCV"])
def test_branch_filters_equal_the_reference(O, emu_module, tmp_path, kind):
    if O.ref("bra") is None:
        pytest.skip("oracle/_ref not built")
    n = 300_007
    x = _code_like(kind, n, 5)[:n].copy()
    src = tmp_path / "in.bin"; x.tofile(src)
    pc = 0x1000 if kind in ("ARM64", "RISCV") else 0                 # only these two take a branch offset (BranchRegister.cpp:56-57)
    enc = _filter(emu_module, NAMES[kind], "enc", pc if pc else "-", src, tmp_path / "enc.bin", 70_001)
    want, done = O.ref_bra_convert(KID[kind], x, pc, True)
    assert np.array_equal(enc, want), kind
    props = tmp_path / "enc.bin.props"
    if pc:
        assert props.read_bytes() == int(pc).to_bytes(4, "little")   # BranchMisc.cpp:66-73
    dec = _filter(emu_module, NAMES[kind], "dec", props if pc else "-", tmp_path / "enc.bin", tmp_path / "dec.bin", 33_333)
    assert np.array_equal(dec, x), kind


def test_bcj_filter_equals_the_reference(O, emu_module, tmp_path):
    if O.ref("bra") is None:
        pytest.skip("oracle/_ref not built")
    x = _x86_like(400_003, 9)
    src = tmp_path / "in.bin"; x.tofile(src)
    enc = _filter(emu_module, "BCJGPU", "enc", "-", src, tmp_path / "enc.bin", 65_537)
    want, done, st = O.ref_bra_x86_convert(x, 0, True, 0)
    assert np.array_equal(enc, want)
    dec = _filter(emu_module, "BCJGPU", "dec", "-", tmp_path / "enc.bin", tmp_path / "dec.bin", 100_000)
    assert np.array_equal(dec, x)


@pytest.mark.parametrize("delta", [1, 4, 256])
def test_delta_filter_equals_the_reference(O, emu_module, tmp_path, delta):
    if O.ref("bra") is None:
        pytest.skip("oracle/_ref not built")
    x = O.corpus("silesia-like", 200_001)
    src = tmp_path / "in.bin"; x.tofile(src)
    enc = _filter(emu_module, "DELTAGPU", "enc", delta, src, tmp_path / "enc.bin", 50_001)
    want, _ = O.ref_delta_convert(x, delta, True)
    assert np.array_equal(enc, want)
    assert (tmp_path / "enc.bin.props").read_bytes() == bytes([delta - 1])           # DeltaFilter.cpp:82-86
    dec = _filter(emu_module, "DELTAGPU", "dec", tmp_path / "enc.bin.props", tmp_path / "enc.bin", tmp_path / "dec.bin", 77_777)
    assert np.array_equal(dec, x)


@pytest.mark.parametrize("hook,code", [("GC_PLUGIN_FILTER_NO_DEVICE=1", 15), ("GC_PLUGIN_FILTER_FAIL_AT_PC=100000", 16)])
def test_filter_failure_is_reported_not_passed_through(emu_module, tmp_path, hook, code):
    """no device: Init() fails (the call CFilterCoder checks); a failure in mid-stream: Filter() answers with a size no buffer has -- never with 0, which the
    filter coder reads as "pass the rest through unfiltered" (FilterCoder.cpp:172-174)"""
    x = _x86_like(300_000, 7)
    src = tmp_path / "in.bin"; x.tofile(src)
    k, v = hook.split("=")
    r = subprocess.run([os.path.join(EMU, "plugin_host"), emu_module, "filter", "BCJGPU", "enc", "-", str(src), str(tmp_path / "out.bin"), "65537"],
                       capture_output=True, text=True, env=dict(os.environ, **{k: v}))
    assert r.returncode == code, (r.returncode, r.stderr)
