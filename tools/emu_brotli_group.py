"""Lab (CPU, emulator): brotli size against the reference at the same quality for GC_BR_GROUP = 1 / 2 / 4 / 8 blocks per meta-block.
usage: python tools/emu_brotli_group.py [MiB] [quality] [corpus ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as graft
import oracle as O
pkg = graft.load_package()
emu = os.path.join(os.path.dirname(__file__), "..", "tests", "emu", "_build", "libgpucodec_emu.so")
mib = float(sys.argv[1]) if len(sys.argv) > 1 else 4
q = int(sys.argv[2]) if len(sys.argv) > 2 else 6
kinds = sys.argv[3:] or ["real-src", "real-bin", "real-py", "text-zipf"]
groups = [int(g) for g in os.environ.get("BRG", "1,4").split(",")]
for kind in kinds:
    x = O.corpus(kind, int(mib * 1024 * 1024))
    ref = len(O.ref_brotlimt_compress(x, q, 8))
    row = []
    for g in groups:
        os.environ["GC_BR_GROUP"] = str(g)
        e = pkg.BrotliEncoder(level=q, lib_path=emu); t0 = time.time(); c = e.code(x); dt = time.time() - t0; e.close()
        assert np.array_equal(O.ref_brotlimt_decompress(c, x.size, 8), x), (kind, g)
        row.append("G=%d %.4f (%.0fs)" % (g, len(c) / ref, dt))
    print(kind, x.size, "ref", ref, " | ".join(row), flush=True)
