"""GPU: links followed per round in the wide execution path (hook GC_ZD_HOPS) against time and rounds."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1000000000
x = O.corpus("text-zipf", n)
enc = pkg.ZstdEncoder(device=0, level=3); own = enc.code(x); enc.close()
y = O.corpus("silesia-like", 128 << 20)
cases = [("text own L3 1 GB", x, own), ("silesia ref L3 one frame 128 MiB", y, O.ref_zstd_compress(y.tobytes(), 3))]
os.environ["GC_ZD_WIDE"] = "1"
for name, want, comp in cases:
    comp = np.ascontiguousarray(np.frombuffer(bytes(comp), dtype=np.uint8)).copy()
    for hops in (1, 2, 3, 4, 6, 8, 12, 16, 32):
        os.environ["GC_ZD_HOPS"] = str(hops)
        dec = pkg.ZstdDecoder(device=0)
        frames, nf, total = dec.scan(comp)
        d_src = torch.from_numpy(comp).cuda(); d_dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        best = 1e9
        for it in range(3):
            dec.code_device(d_src.data_ptr(), comp.size, d_dst.data_ptr(), total, frames, nf)
            if dec.last_timing_ms() < best: best = dec.last_timing_ms(); k = dec.kernel_timing_ms()
        ok = bool((d_dst[:total].cpu().numpy() == want[:total]).all())
        print("%-34s hops %2d: %7.2f ms  exec %6.2f ms rounds %2d ok=%s" % (name, hops, best, k["execution"], dec.wide_rounds(), ok), flush=True)
        dec.close(); del d_src, d_dst
