"""GPU probe: small streams through the decoder with the several-blocks-per-wave sequences kernel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
os.environ["GC_ZD_SEQV"] = "1"
x = O.corpus("text-zipf", 4 << 20)
for name, n, lvl, own in (("own 100K", 100_000, 3, True), ("ref 100K L1", 100_000, 1, False), ("ref 100K L3", 100_000, 3, False), ("ref 700K L3", 700_000, 3, False), ("own 4M", 4 << 20, 3, True), ("zeros", 0, 3, False)):
    d = x[:n] if n else np.zeros(300_000, dtype=np.uint8)
    if own:
        enc = pkg.ZstdEncoder(device=0, level=lvl); comp = enc.code(d); enc.close()
    else:
        comp = O.ref_zstd_compress(d.tobytes(), lvl)
    for wide in ("0", "1"):
        os.environ["GC_ZD_WIDE"] = wide
        dec = pkg.ZstdDecoder(device=0)
        try:
            out = dec.code(bytes(comp), capacity=d.size + 64)
            print(name, "wide", wide, "ok" if out.tobytes() == d.tobytes() else "MISMATCH at %d" % int(np.argmax(out[:d.size] != d[:out.size]) if out.size == d.size else -1), flush=True)
        except Exception as e:
            print(name, "wide", wide, "ERROR", str(e)[:100], flush=True)
        dec.close()
