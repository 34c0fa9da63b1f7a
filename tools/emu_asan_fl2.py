"""CPU: the FLZMA2 encoder kernels and the zstd ones of level 19 (both with the lane-per-window price parse) under the SIMT emulator built with AddressSanitizer (make -C tests/emu asan): every
access to the input, the workspaces and the LDS arrays is checked; the streams must decode.
usage: ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python tools/emu_asan_fl2.py [seed] [seconds]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
lib = os.path.join(ROOT, 'tests', 'emu', '_asan', 'libgpucodec_asan.so')
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 120
kinds = ['silesia-like', 'text-zipf', 'lz-7zip', 'real-bin', 'real-src']
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    n = int(2 ** rng.uniform(6, 19.5))
    src = O.corpus(kinds[it % len(kinds)], max(n, 1 << 20))
    off = int(rng.integers(0, src.size - n + 1))
    x = np.ascontiguousarray(src[off:off + n])                   # an exact-size buffer of its own: reads outside it are caught
    x = x.copy()
    for level in (5, 3):
        e = pkg.Flzma2Encoder(level=level, lib_path=lib); c = e.code(x); prop = e.coder_props()[0]; e.close()
        assert np.array_equal(O.ref_lzma2_decode(c, x.size, prop), x), ("decode", n, level)
    e = pkg.ZstdEncoder(level=19, lib_path=lib); c = e.code(x); e.close()
    assert np.array_equal(O.ref_zstd_decompress(c, x.size), x), ("zstd decode", n)
    it += 1
print("iterations", it)
