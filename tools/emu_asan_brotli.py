"""CPU: the BROTLI encoder kernels (qualities 0 / 1 / 6 / 9, brotli-mt framed and plain streams in pieces) under the SIMT emulator built with AddressSanitizer
(make -C tests/emu asan): every access to the input (exact-size buffers), the workspaces and the LDS arrays is checked -- B1's last-distance
substitution reads the source at a copy's position minus the distances of the commands in front; the streams must decode under the reference decoder.
usage: ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python tools/emu_asan_brotli.py [seed] [seconds]"""
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
kinds = ['real-bin', 'silesia-like', 'text-zipf', 'lz-7zip', 'real-src', 'web-text']
t0 = time.time(); it = 0
while time.time() - t0 < secs:
    n = int(2 ** rng.uniform(1, 20.3))
    src = O.corpus(kinds[it % len(kinds)], max(n, 1 << 21))
    off = int(rng.integers(0, src.size - n + 1))
    x = np.ascontiguousarray(src[off:off + n]).copy()           # an exact-size buffer of its own: reads outside it are caught
    for q in (6, 9, 5, 7, 1, 0)[: (6 if it % 3 == 0 else 2)]:
        os.environ["GC_BR_REPSUB"] = str(1 + it % 2)
        e = pkg.BrotliEncoder(level=q, lib_path=lib); c = e.code(x); e.close()
        assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 2), x), ("decode", n, q)
    it += 1
print("iterations", it, "seconds", round(time.time() - t0))
