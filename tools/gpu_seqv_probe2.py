import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
os.environ["GC_ZD_SEQV"] = "1"; os.environ["GC_ZD_WIDE"] = "1"; os.environ["GC_ZD_PROF"] = "1"; os.environ["GC_ZD_SEQV_DBG"] = "1"
x = O.corpus("text-zipf", 100_000)
comp = O.ref_zstd_compress(x.tobytes(), 3)
kw = {"lib_path": sys.argv[1]} if len(sys.argv) > 1 else {"device": 0}
dec = pkg.ZstdDecoder(**kw)
try:
    out = dec.code(bytes(comp), capacity=x.size + 64); print("ok", out.tobytes() == x.tobytes())
except Exception as e:
    print("ERROR", str(e)[:100])
