"""GPU bisect probe: variants of the library (tools/_variants/lib_*.so) on small streams with the several-blocks sequences kernel."""
import sys, os, glob, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import __graft_entry__ as g
    pkg = g.load_package()
    import oracle as O
    os.environ["GC_ZD_SEQV"] = "1"; os.environ["GC_ZD_WIDE"] = "1"
    x = O.corpus("text-zipf", 1 << 20)
    res = []
    for name, n, lvl in (("100K L3", 100_000, 3), ("1M L3", 1 << 20, 3), ("zeros", 0, 3)):
        d = x[:n] if n else np.zeros(300_000, dtype=np.uint8)
        comp = O.ref_zstd_compress(d.tobytes(), lvl)
        dec = pkg.ZstdDecoder(lib_path=sys.argv[1], device=0)
        try:
            out = dec.code(bytes(comp), capacity=d.size + 64); res.append("ok" if out.tobytes() == d.tobytes() else "MISMATCH")
        except Exception as e:
            res.append("ERR")
        dec.close()
    print(os.path.basename(sys.argv[1]), res, flush=True)
else:
    for lib in sorted(glob.glob(os.path.join(ROOT, "tools", "_variants", "lib_*.so"))):
        env = dict(os.environ, GC_ZD_PROF="1", GC_ZD_SEQV_DBG="1")
        r = subprocess.run([sys.executable, __file__, lib], capture_output=True, text=True, env=env, timeout=120)
        lines = [l for l in (r.stdout + r.stderr).splitlines() if "lib_" in l or "dbg 14" in l or "dbg 15" in l or "Error" in l or "error" in l]
        print("\n".join(lines[:12]), flush=True)
