"""CPU bring-up of the real kernel sources under the SIMT emulator (tests/emu): every stream must decode
bit-exactly under the oracle decoder and the reference decoder.  Sizes are small: the emulator runs one
workgroup at a time at roughly 1 s per 128 KiB block."""
import numpy as np
import pytest

BLK = 128 * 1024


def _roundtrip(O, enc, x):
    c = enc.code(x)
    y = O.port_zstd_decompress(c, x.size)
    assert np.array_equal(x, y)
    if O.ref("zstd") is not None:
        assert np.array_equal(O.ref_zstd_decompress(c, x.size), x)
    return c


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 8, 9, 63, 64, 255, 256, 257, 1023, 1024, 1025, 4097])
def test_tiny_inputs(O, emu_enc, n):
    _roundtrip(O, emu_enc, O.corpus("text-zipf", n))


@pytest.mark.parametrize("kind", ["text-zipf", "lz-7zip", "silesia-like", "web-text"])
def test_corpora_multi_block(O, emu_enc, kind):
    x = O.corpus(kind, 2 * BLK + 4321)
    c = _roundtrip(O, emu_enc, x)
    if O.ref("zstd") is not None:
        ref = O.ref_zstd_compress(x, 3, piece=BLK)
        assert len(c) <= 1.05 * len(ref), (len(c), len(ref))


@pytest.mark.parametrize("n", [BLK - 1, BLK, BLK + 1])
def test_block_boundaries(O, emu_enc, n):
    _roundtrip(O, emu_enc, O.corpus("text-zipf", n))


def test_incompressible_goes_raw(O, emu_enc):
    x = O.corpus("random", BLK + 777)
    c = _roundtrip(O, emu_enc, x)
    assert len(c) <= emu_enc.compress_bound(x.size)
    assert len(c) >= x.size


def test_all_same_and_long_matches(O, emu_enc):
    _roundtrip(O, emu_enc, O.corpus("zeros", BLK + 100))
    # long period-7 pattern: matches far longer than 65535 after merging, litLength 0 chains
    x = np.tile(np.arange(7, dtype=np.uint8), (BLK + 50) // 7 + 1)[:BLK + 50].copy()
    _roundtrip(O, emu_enc, x)
    # one literal run longer than 65535 followed by a repeat of it
    r = O.corpus("random", 70_000)
    _roundtrip(O, emu_enc, np.concatenate([r, r[:50_000]]))


def test_binary_alphabet_over_128_symbols(O, emu_enc):
    # skewed distribution over all 256 byte values: exercises the FSE-compressed Huffman weight header
    rng = np.random.default_rng(7)
    p = 1.0 / np.arange(1, 257) ** 1.2; p /= p.sum()
    x = rng.choice(256, size=BLK, p=p).astype(np.uint8)
    _roundtrip(O, emu_enc, x)


def test_deterministic(O, emu_enc):
    x = O.corpus("silesia-like", BLK)
    assert np.array_equal(emu_enc.code(x), emu_enc.code(x))


def test_watchdog_word_turns_into_an_error_code(pkg, emu_lib_path, O, monkeypatch):
    """the inter-workgroup waits of the fused verify + parse kernel are bounded; a wait that runs out sets a word that the finish call turns into
    GC_ERR_HIP (test hook GC_WATCHDOG_TRIP: as if that had happened) -- an error code, not a hang and not a stream"""
    x = O.corpus("text-zipf", 300_000)
    e = pkg.ZstdEncoder(level=3, lib_path=emu_lib_path)
    assert len(e.code(x)) > 0
    monkeypatch.setenv("GC_WATCHDOG_TRIP", "1")
    with pytest.raises(pkg.GpuCodecError):
        e.code(x)
    monkeypatch.delenv("GC_WATCHDOG_TRIP")
    assert len(e.code(x)) > 0
    e.close()


_GARBAGE_SCRIPT = r"""
import sys, hashlib
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/oracle')
import numpy as np
import __graft_entry__ as g, oracle as O
pkg = g.load_package()
x = np.concatenate([O.corpus('silesia-like', 131072 + 4321), O.corpus('text-zipf', 30000)])
for name, cls, level in (('zstd', pkg.ZstdEncoder, 3), ('zstd', pkg.ZstdEncoder, 19), ('flzma2', pkg.Flzma2Encoder, 5), ('brotli', pkg.BrotliEncoder, 6)):
    e = cls(level=level, lib_path=sys.argv[2]); c = e.code(x); e.close()
    print(name, level, len(c), hashlib.sha1(c.tobytes()).hexdigest())
"""


def test_streams_do_not_depend_on_what_memory_held(emu_lib_path, graft):
    """Round 4 (profiles/r04_defects.md): on the device a fresh allocation and a workgroup's LDS hold whatever was there before; the emulator's malloc hands out zero
    pages and its `__shared__` statics keep the previous workgroup's values.  With every allocation and all of LDS filled with garbage (GC_EMU_POISON, GC_EMU_POISON_LDS:
    tests/emu/hip_runtime_stub.h, hipemu.cpp) the encoders must produce the same bytes."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def run(extra):
        env = dict(os.environ); env.update(extra)
        r = subprocess.run([sys.executable, "-c", _GARBAGE_SCRIPT, root, emu_lib_path], capture_output=True, text=True, env=env, timeout=1500)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout
    plain = run({})
    assert plain.count("\n") == 4
    assert run({"GC_EMU_POISON": "-7", "GC_EMU_POISON_LDS": "3", "GC_POISON_WORKSPACE": "90"}) == plain
