"""CPU: the decoder kernels under the SIMT emulator built with AddressSanitizer (make -C tests/emu asan), fed reference-encoder streams with random
bytes overwritten or cut short: every access of the kernels to the (host-allocated) workspaces, streams and LDS arrays is checked.
usage: ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so) python tools/emu_asan_fuzz.py [seed] [seconds]"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
lib = os.path.join(ROOT, 'tests', 'emu', '_asan', 'libgpucodec_asan.so')
dec = pkg.ZstdDecoder(lib_path=lib)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t0 = time.time(); it = ref = 0
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 120
kinds = ['silesia-like','text-zipf','lz-7zip']
while time.time() - t0 < secs:
    n = int(2 ** rng.uniform(4, 18.5))
    x = O.corpus(kinds[it % 3], n)[:n]
    opts = dict(checksum=bool(rng.integers(0,2)), streamed=bool(rng.integers(0,2)))
    c = O.ref_zstd_compress_opts(x.tobytes(), int(rng.choice([1,3,7,17,19])), **opts)
    out = dec.code(c, capacity=n + 64)
    assert np.array_equal(out, x), "clean stream mismatch"
    for k in range(6):
        bad = c.copy()
        m = int(rng.integers(1, 4))
        for _ in range(m):
            bad[int(rng.integers(0, bad.size))] = rng.integers(0, 256)
        if k == 5: bad = bad[: int(rng.integers(1, bad.size))]
        try:
            dec.code(bad, capacity=n + 64)
        except pkg.GpuCodecError:
            ref += 1
    it += 1
print("iterations", it, "damaged streams refused", ref, "of", it * 6)
