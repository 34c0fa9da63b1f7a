"""Lab (CPU, emulator): brotli size against the reference with B1's last-distance substitution (hook GC_BR_REPSUB = passes).
usage: python tools/emu_brotli_repsub.py [MiB] [quality] [corpus ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as graft
import oracle as O
pkg = graft.load_package()
emu = os.path.join(ROOT, "tests", "emu", "_build", "libgpucodec_emu.so")
mib = float(sys.argv[1]) if len(sys.argv) > 1 else 2
q = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for kind in sys.argv[3:] or ["real-bin", "real-src", "real-py", "text-zipf"]:
    x = O.corpus(kind, int(mib * 1024 * 1024))
    ref = len(O.ref_brotlimt_compress(x, q, 8))
    row = []
    for passes in (0, 1, 2):
        os.environ["GC_BR_REPSUB"] = str(passes)
        e = pkg.BrotliEncoder(level=q, lib_path=emu); c = e.code(x); e.close()
        assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 8), x), (kind, passes)
        row.append("passes=%d %.4f" % (passes, len(c) / ref))
    print(kind, x.size, "ref", ref, " | ".join(row), flush=True)
