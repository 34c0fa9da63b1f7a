"""GPU / emulator: FLZMA2 level 5 of 1 MiB slices of the Silesia stand-in around block 38 (where the device and the emulator disagreed), with state dumps.
usage: python tools/gpu_diag3.py <outdir> [emu]"""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
out = sys.argv[1]; emu = len(sys.argv) > 2
os.makedirs(out, exist_ok=True)
x = O.corpus('silesia-like', 32 << 20)
lib = os.path.join(ROOT, 'tests', 'emu', '_build', 'libgpucodec_emu.so') if emu else g.LIB_HOOKS
for k, (a, n) in enumerate([(36 * 131072, 1 << 20), (38 * 131072, 1 << 19), (40 * 131072, 1 << 18), (38 * 131072, 1 << 17)]):
    d = os.path.join(out, 's%d' % k); os.makedirs(d, exist_ok=True)
    os.environ['GC_DUMP_STATE'] = d
    y = np.ascontiguousarray(x[a:a + n])
    e = pkg.Flzma2Encoder(level=5, lib_path=lib) if emu else pkg.Flzma2Encoder(level=5, device=0, lib_path=lib)
    c = e.code(y); e.close()
    print(k, a, n, len(c), hashlib.sha1(c.tobytes()).hexdigest()[:12], flush=True)
