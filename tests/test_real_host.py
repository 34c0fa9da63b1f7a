"""Config C1 under the REAL host: the reference's own `7z` console program (built by oracle/build_ref_7z.sh from
/root/reference into oracle/_ref/host7z/, test infrastructure) loads the plugin from its Codecs/ directory the way the product
does (CPP/7zip/UI/Common/LoadCodecs.cpp:531-650), lists its methods (`7z i`), and runs `7z a -m0=<method> -mx<level>` through
the plugin's ICompressCoder::Code / SetCoderProperties / WriteCoderProperties (CPP/7zip/Archive/7z/7zEncode.cpp:160-304);
`7z t` / `7z x` then decode the archive with the host's own built-in decoders (decoder lookup is by method id,
CPP/7zip/Common/CreateCoder.cpp:206-232).

The host has ZSTD / FLZMA2 / BROTLI built in and resolves a method NAME to its built-in encoder first, so the plugin is driven
through its alias names ZSTDGPU / FLZMA2GPU / BROTLIGPU (same method ids).

CPU: the plugin layer over the emulator build of the kernels.  GPU (-m gpu): the product module lib7zgpucodec.so."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "_build")
HOST = os.path.join(ROOT, "oracle", "_ref", "host7z")


def _have_host():
    return os.path.exists(os.path.join(HOST, "7z")) and os.path.exists(os.path.join(HOST, "7z.so"))


@pytest.fixture(scope="module")
def host_dir(tmp_path_factory):
    """A private install directory: 7z + 7z.so + Codecs/ (the host looks for Codecs/ next to its own binary)."""
    if not _have_host():
        if os.path.isdir("/root/reference/CPP"):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_ref/host7z/7z"], check=True, capture_output=True)
        if not _have_host():
            pytest.skip("reference host not built (oracle/_ref/host7z)")
    d = tmp_path_factory.mktemp("host7z")
    for f in ("7z", "7z.so"):
        shutil.copy2(os.path.join(HOST, f), d / f)
    (d / "Codecs").mkdir()
    return d


@pytest.fixture(scope="module")
def host_dir_nozstd(tmp_path_factory, host_dir):
    """The same host with a format/codec bundle that has no ZSTD codec of its own (oracle/build_ref_7z.sh links it without ZstdRegister.o),
    like mainline 7-Zip: method id 4F71101 then resolves to the plugin for encoding AND decoding."""
    if not os.path.exists(os.path.join(HOST, "7z_nozstd.so")):
        if os.path.isdir("/root/reference/CPP"):
            subprocess.run(["bash", os.path.join(ROOT, "oracle", "build_ref_7z.sh"), HOST], check=True, capture_output=True, env=dict(os.environ, REF_ROOT="/root/reference"))
        if not os.path.exists(os.path.join(HOST, "7z_nozstd.so")):
            pytest.skip("reference host without zstd not built (oracle/_ref/host7z/7z_nozstd.so)")
    d = tmp_path_factory.mktemp("host7z_nozstd")
    shutil.copy2(os.path.join(HOST, "7z"), d / "7z")
    shutil.copy2(os.path.join(HOST, "7z_nozstd.so"), d / "7z.so")
    (d / "Codecs").mkdir()
    return d


@pytest.fixture(scope="module")
def host_dir_nobrotli(tmp_path_factory, host_dir):
    """The host with a bundle that has no BROTLI codec of its own (linked without BrotliRegister.o; the brotli library itself stays inside): method id 4F71102 resolves to the
    plugin for encoding AND decoding, and the plugin finds the RFC's static dictionary in the host (BrotliGetDictionary of its 7z.so)."""
    if not os.path.exists(os.path.join(HOST, "7z_nobrotli.so")):
        if os.path.isdir("/root/reference/CPP"):
            subprocess.run(["bash", os.path.join(ROOT, "oracle", "build_ref_7z.sh"), HOST], check=True, capture_output=True, env=dict(os.environ, REF_ROOT="/root/reference"))
        if not os.path.exists(os.path.join(HOST, "7z_nobrotli.so")):
            pytest.skip("reference host without brotli not built (oracle/_ref/host7z/7z_nobrotli.so)")
    d = tmp_path_factory.mktemp("host7z_nobrotli")
    shutil.copy2(os.path.join(HOST, "7z"), d / "7z")
    shutil.copy2(os.path.join(HOST, "7z_nobrotli.so"), d / "7z.so")
    (d / "Codecs").mkdir()
    return d


def _install(host_dir, module, libdir):
    for f in os.listdir(host_dir / "Codecs"):
        os.remove(host_dir / "Codecs" / f)
    shutil.copy2(module, host_dir / "Codecs" / os.path.basename(module))
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = libdir + os.pathsep + env.get("LD_LIBRARY_PATH", "")     # libgpucodec*.so stays outside Codecs/ (the host would try to load it as a codec module)
    return env


def _run(host_dir, env, *args):
    return subprocess.run([str(host_dir / "7z")] + [str(a) for a in args], capture_output=True, text=True, env=env, cwd=host_dir)


def _check_listing(out, module_name):
    assert module_name in out, out
    lines = [l.split() for l in out.splitlines()]
    ours = [l for l in lines if len(l) == 4 and l[0] == "1" and l[1] in ("E", "ED", "EDF")]   # "<lib index> E|ED|EDF <id> <name>" from library 1
    got = {(l[2], l[3]) for l in ours}
    filters = {"BCJGPU", "PPCGPU", "IA64GPU", "ARMGPU", "ARMTGPU", "SPARCGPU", "ARM64GPU", "RISCVGPU", "DELTAGPU"}        # the pre-filters on the device (round 3)
    assert {l[3] for l in ours if l[1] == "ED"} == {"ZSTD", "ZSTDGPU", "BROTLI", "BROTLIGPU"}     # encoder + decoder: ZSTD and (round 6) BROTLI
    assert {l[3] for l in ours if l[1] == "EDF"} == filters                          # ... and the filters, which the host recognises as such
    for want in [("4F71101", "ZSTD"), ("21", "FLZMA2"), ("4F71102", "BROTLI"), ("4F71101", "ZSTDGPU"), ("21", "FLZMA2GPU"), ("4F71102", "BROTLIGPU")]:
        assert want in got, (want, out)


def _roundtrip(host_dir, env, O, method, level, expect_method, n, kind="text-zipf"):
    x = O.corpus(kind, n)
    src = host_dir / ("f_%s_%d.bin" % (method, n))
    x.tofile(src)
    arc = host_dir / ("a_%s_%d.7z" % (method, n))
    if arc.exists():
        arc.unlink()
    r = _run(host_dir, env, "a", "-m0=%s" % method, "-mx%d" % level, arc.name, src.name)
    assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr
    r = _run(host_dir, env, "t", arc.name)
    assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr
    r = _run(host_dir, env, "l", "-slt", arc.name)
    assert ("Method = " + expect_method) in r.stdout, r.stdout                        # the archive records the method id (0x21 is listed as LZMA2)
    out = host_dir / "x"
    if out.exists():
        shutil.rmtree(out)
    r = _run(host_dir, env, "x", "-o" + str(out), arc.name)
    assert r.returncode == 0, r.stdout + r.stderr
    y = np.fromfile(out / src.name, dtype=np.uint8)
    assert np.array_equal(x, y)
    return os.path.getsize(arc)


def test_real_host_lists_the_emulator_module(host_dir, emu_lib_path):
    env = _install(host_dir, os.path.join(EMU, "lib7zgpucodec_emu.so"), EMU)
    r = _run(host_dir, env, "i")
    assert r.returncode == 0, r.stderr
    _check_listing(r.stdout, "lib7zgpucodec_emu.so")


@pytest.mark.parametrize("method,level,expect,n", [("ZSTDGPU", 1, "ZSTD", 1048576),        # BASELINE config C1: zstd level 1 on a 1 MiB buffer
                                                   ("ZSTDGPU", 3, "ZSTD", 300000),
                                                   ("FLZMA2GPU", 1, "LZMA2", 200000),
                                                   ("BROTLIGPU", 1, "BROTLI", 200000)])
def test_real_host_archives_through_the_emulator_module(host_dir, emu_lib_path, O, method, level, expect, n):
    env = _install(host_dir, os.path.join(EMU, "lib7zgpucodec_emu.so"), EMU)
    size = _roundtrip(host_dir, env, O, method, level, expect, n)
    assert size < n


def _decoder_under_real_host(host_dir, host_dir_nozstd, env_full, env_nozstd, O, n, level):
    """(1) the host without its own ZSTD codec archives with -m0=ZSTD (the plugin's encoder) and tests / extracts with the plugin's DECODER;
    (2) an archive written by the reference's own CPU encoder is extracted by the plugin's decoder; (3) and the other way round."""
    r = _run(host_dir_nozstd, env_nozstd, "i")
    assert r.returncode == 0 and " ED " in r.stdout and "4F71101 ZSTD" in r.stdout, r.stdout
    _roundtrip(host_dir_nozstd, env_nozstd, O, "ZSTD", level, "ZSTD", n)
    x = O.corpus("silesia-like", n)
    src = host_dir / "cpu_src.bin"
    x.tofile(src)
    arc = host_dir / "cpu.7z"
    if arc.exists():
        arc.unlink()
    r = _run(host_dir, env_full, "a", "-m0=zstd", "-mx%d" % level, arc.name, src.name)                 # the reference's built-in encoder
    assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr
    shutil.copy2(arc, host_dir_nozstd / "cpu.7z")
    out = host_dir_nozstd / "x_cpu"
    if out.exists():
        shutil.rmtree(out)
    r = _run(host_dir_nozstd, env_nozstd, "x", "-o" + str(out), "cpu.7z")                                 # ... decoded by the plugin
    assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(out / src.name, dtype=np.uint8), x)
    r = _run(host_dir, env_full, "t", str(host_dir_nozstd / ("a_ZSTD_%d.7z" % n)))                        # the plugin's archive under the reference's decoder
    assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr


def _brotli_decoder_under_real_host(host_dir, host_dir_nobrotli, env_full, env_nob, O, n, level, kind):
    """(1) the host without its own BROTLI codec archives with -m0=BROTLI (the plugin's encoder) and tests / extracts with the plugin's DECODER; (2) an archive written by the
    reference's own CPU encoder (brotli-mt frames, references to the static dictionary) is extracted by the plugin's decoder, which takes the dictionary from the host's
    BrotliGetDictionary; (3) the plugin's archive under the reference's decoder."""
    r = _run(host_dir_nobrotli, env_nob, "i")
    assert r.returncode == 0 and "4F71102 BROTLI" in r.stdout, r.stdout
    assert [l for l in r.stdout.splitlines() if l.split()[-2:] == ["4F71102", "BROTLI"] and " ED " in l], r.stdout
    _roundtrip(host_dir_nobrotli, env_nob, O, "BROTLI", level, "BROTLI", n)
    x = O.corpus(kind, n)
    src = host_dir / "cpu_src_br.bin"
    x.tofile(src)
    for lv in sorted({level, 6, 11} if n <= 2_000_000 else {level, 6}):
        arc = host_dir / ("cpu_br%d.7z" % lv)
        if arc.exists():
            arc.unlink()
        r = _run(host_dir, env_full, "a", "-m0=brotli", "-mx%d" % lv, arc.name, src.name)                # the reference's built-in encoder
        assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr
        shutil.copy2(arc, host_dir_nobrotli / arc.name)
        out = host_dir_nobrotli / "x_cpu_br"
        if out.exists():
            shutil.rmtree(out)
        r = _run(host_dir_nobrotli, env_nob, "x", "-o" + str(out), arc.name)                              # ... decoded by the plugin
        assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(out / src.name, dtype=np.uint8), x)
    r = _run(host_dir, env_full, "t", str(host_dir_nobrotli / ("a_BROTLI_%d.7z" % n)))                   # the plugin's archive under the reference's decoder
    assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr


def test_real_host_decodes_brotli_through_the_emulator_module(host_dir, host_dir_nobrotli, emu_lib_path, O):
    module = os.path.join(EMU, "lib7zgpucodec_emu.so")
    kind = "real-src" if O.corpus("real-src", 1 << 20).size >= (1 << 20) else "text-zipf"
    _brotli_decoder_under_real_host(host_dir, host_dir_nobrotli, _install(host_dir, module, EMU), _install(host_dir_nobrotli, module, EMU), O, 300_000, 1, kind)


@pytest.mark.gpu
def test_gpu_real_host_decodes_brotli_through_the_product_module(host_dir, host_dir_nobrotli, graft, O):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    module = graft.build_plugin()
    libdir = os.path.join(ROOT, "7-zip-zstd_amd", "csrc")
    kind = "real-src" if O.corpus("real-src", 1 << 20).size >= (1 << 20) else "text-zipf"
    _brotli_decoder_under_real_host(host_dir, host_dir_nobrotli, _install(host_dir, module, libdir), _install(host_dir_nobrotli, module, libdir), O, 100_000_000, 6, kind)


def test_real_host_decodes_through_the_emulator_module(host_dir, host_dir_nozstd, emu_lib_path, O):
    module = os.path.join(EMU, "lib7zgpucodec_emu.so")
    _decoder_under_real_host(host_dir, host_dir_nozstd, _install(host_dir, module, EMU), _install(host_dir_nozstd, module, EMU), O, 400_000, 3)


@pytest.mark.gpu
def test_gpu_real_host_decodes_through_the_product_module(host_dir, host_dir_nozstd, graft, O):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    module = graft.build_plugin()
    libdir = os.path.join(ROOT, "7-zip-zstd_amd", "csrc")
    _decoder_under_real_host(host_dir, host_dir_nozstd, _install(host_dir, module, libdir), _install(host_dir_nozstd, module, libdir), O, 150_000_000, 3)


@pytest.mark.gpu
def test_gpu_real_host_archives_through_the_product_module(host_dir, graft, O):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    module = graft.build_plugin()
    env = _install(host_dir, module, os.path.join(ROOT, "7-zip-zstd_amd", "csrc"))
    r = _run(host_dir, env, "i")
    assert r.returncode == 0, r.stderr
    _check_listing(r.stdout, "lib7zgpucodec.so")
    _roundtrip(host_dir, env, O, "ZSTDGPU", 1, "ZSTD", 1048576)                       # config C1
    _roundtrip(host_dir, env, O, "ZSTDGPU", 3, "ZSTD", 150_000_000)                   # three pieces of 64 MiB over the host scheduler
    _roundtrip(host_dir, env, O, "FLZMA2GPU", 5, "LZMA2", 80_000_000, "silesia-like")        # (one piece: FLZMA2 pieces are 256 MiB)
    _roundtrip(host_dir, env, O, "BROTLIGPU", 6, "BROTLI", 80_000_000, "web-text")


# The reference's own regression archives (tests/regr-arc, fixtures of tests/regression.test:66-88,153-173; committed as data under tests/golden/): written by the
# reference's ZSTD encoder at level 17 / max, solid and non-solid.  The host WITHOUT a ZSTD codec of its own resolves method id 4F71101 to this module, so `7z t` / `7z x`
# run the plugin's decoder; the pinned results are the CRCs (the host checks them itself: "Everything is Ok") and the SHA-256 of the decoded data.
REGR_ARCS = [("test.txt.zstd.7z", {"test.txt": (1000000, "C601982A", "aeda0f81c8376d1678af53927a08cf641cafab8b68aef509c881eb0be0bc3c97")}),
             ("test-sol.zstd.7z", {"test.txt": (1000000, "C601982A", "aeda0f81c8376d1678af53927a08cf641cafab8b68aef509c881eb0be0bc3c97"), "tesx.txt": (100000, "7ECE9EBC", None)}),
             ("test-sol.zstd.max.7z", {"test.txt": (1000000, "C601982A", "aeda0f81c8376d1678af53927a08cf641cafab8b68aef509c881eb0be0bc3c97"), "tesx.txt": (100000, "7ECE9EBC", None)})]


def _regression_archives(host_dir_nozstd, env):
    import hashlib
    import zlib
    r = _run(host_dir_nozstd, env, "i")
    assert r.returncode == 0 and "4F71101 ZSTD" in r.stdout, r.stdout                 # the only ZSTD decoder this host has is the module's
    for name, files in REGR_ARCS:
        arc = os.path.join(ROOT, "tests", "golden", name)
        r = _run(host_dir_nozstd, env, "t", arc)
        assert r.returncode == 0 and "Everything is Ok" in r.stdout, name + "\n" + r.stdout + r.stderr
        r = _run(host_dir_nozstd, env, "l", "-slt", arc)
        for fn, (size, crc, _) in files.items():
            assert ("Path = " + fn) in r.stdout and ("CRC = " + crc) in r.stdout and ("Size = %d" % size) in r.stdout, r.stdout
        assert "Method = ZSTD" in r.stdout, r.stdout
        out = host_dir_nozstd / ("x_regr_" + name)
        if out.exists():
            shutil.rmtree(out)
        r = _run(host_dir_nozstd, env, "x", "-o" + str(out), arc)
        assert r.returncode == 0 and "Everything is Ok" in r.stdout, name + "\n" + r.stdout + r.stderr
        for fn, (size, crc, sha) in files.items():
            data = (out / fn).read_bytes()
            assert len(data) == size and ("%08X" % (zlib.crc32(data) & 0xFFFFFFFF)) == crc, (name, fn)
            if sha is not None:
                assert hashlib.sha256(data).hexdigest() == sha, (name, fn)


def test_real_host_extracts_the_reference_regression_archives_through_the_emulator_module(host_dir_nozstd, emu_lib_path):
    _regression_archives(host_dir_nozstd, _install(host_dir_nozstd, os.path.join(EMU, "lib7zgpucodec_emu.so"), EMU))


@pytest.mark.gpu
def test_gpu_real_host_extracts_the_reference_regression_archives(host_dir_nozstd, graft):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    module = graft.build_plugin()
    _regression_archives(host_dir_nozstd, _install(host_dir_nozstd, module, os.path.join(ROOT, "7-zip-zstd_amd", "csrc")))


def _filter_chain(host_dir, env, O, n_bcj, n_delta):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_bra import _x86_like
    for tag, methods, x in (("bcj", ["-m0=BCJGPU", "-m1=ZSTDGPU", "-mx3"], _x86_like(n_bcj, 3)),
                            ("delta", ["-m0=DELTAGPU:4", "-m1=ZSTDGPU", "-mx3"], O.corpus("silesia-like", n_delta))):
        src = host_dir / ("flt_%s.bin" % tag); x.tofile(src)
        arc = host_dir / ("flt_%s.7z" % tag)
        if arc.exists():
            arc.unlink()
        r = _run(host_dir, env, "a", *methods, arc.name, src.name)
        assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr
        r = _run(host_dir, env, "l", "-slt", arc.name)
        assert ("BCJ" if tag == "bcj" else "Delta:4") in r.stdout and "ZSTD" in r.stdout, r.stdout
        r = _run(host_dir, env, "t", arc.name)                                        # built-in decoders first (CreateCoder.cpp:206-232)
        assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout + r.stderr
        out = host_dir / ("x_flt_" + tag)
        if out.exists():
            shutil.rmtree(out)
        r = _run(host_dir, env, "x", "-o" + str(out), arc.name)
        assert r.returncode == 0, r.stdout + r.stderr
        assert np.array_equal(np.fromfile(out / src.name, dtype=np.uint8), x)


@pytest.mark.gpu
def test_gpu_real_host_filter_chain_through_the_product_module(host_dir, graft, O):
    """the same chain on the device: 7-Zip's filter coder hands its buffer to gc_filter_host call by call"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    graft.build_hip()
    module = graft.build_plugin()
    _filter_chain(host_dir, _install(host_dir, module, os.path.join(ROOT, "7-zip-zstd_amd", "csrc")), O, 40_000_000, 20_000_000)


def test_real_host_filter_chain_through_the_emulator_module(host_dir, emu_lib_path, O):
    """`7z a -m0=BCJGPU -m1=ZSTDGPU`: the reference's own host runs this module's x86 branch converter in front of this module's ZSTD encoder (7-Zip's filter
    coder drives Filter() on its buffer), records the reference's method ids, and tests / extracts the archive with its BUILT-IN BCJ and ZSTD decoders;
    the same with the Delta filter and its property."""
    _filter_chain(host_dir, _install(host_dir, os.path.join(EMU, "lib7zgpucodec_emu.so"), EMU), O, 300_000, 200_000)


@pytest.mark.parametrize("hook", ["GC_PLUGIN_FILTER_NO_DEVICE=1", "GC_PLUGIN_FILTER_FAIL_AT_PC=65536"])
def test_real_host_filter_failure_is_an_error_not_an_unfiltered_archive(host_dir, emu_lib_path, hook):
    """A filter that cannot run -- no device when the coder is set up, or a device failure in mid-stream -- must end `7z a` with an error: a return of 0
    from Filter() would make CFilterCoder write the bytes through unfiltered into a folder that declares the filter (FilterCoder.cpp:172-174)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_bra import _x86_like
    env = _install(host_dir, os.path.join(EMU, "lib7zgpucodec_emu.so"), EMU)
    k, v = hook.split("="); env[k] = v
    x = _x86_like(400_000, 5)
    src = host_dir / "flt_fail.bin"; x.tofile(src)
    arc = host_dir / "flt_fail.7z"
    if arc.exists():
        arc.unlink()
    r = _run(host_dir, env, "a", "-m0=BCJGPU", "-m1=ZSTDGPU", "-mx3", arc.name, src.name)
    assert r.returncode != 0 and "Everything is Ok" not in r.stdout, r.stdout + r.stderr
    if arc.exists():                                                   # whatever was left behind must not pass as an archive of the file
        t = _run(host_dir, dict(env, **{k: "0"}), "t", arc.name)
        assert t.returncode != 0 or "flt_fail.bin" not in _run(host_dir, env, "l", arc.name).stdout
