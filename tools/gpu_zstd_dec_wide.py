"""GPU: the two execution paths of the zstd decoder side by side (frame-per-workgroup kernel vs the wide pointer-jumping path) on streams of
many frames, few frames and one frame.  usage: gpu_zstd_dec_wide.py [bytes]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 256 << 20
cases = []
x = O.corpus("text-zipf", n)
enc = pkg.ZstdEncoder(device=0, level=3); cases.append(("text own L3 (8 MiB frames)", x, enc.code(x))); enc.close()
m = min(n, 128 << 20)
cases.append(("text ref L3 one frame", x[:m], O.ref_zstd_compress(x[:m].tobytes(), 3)))
cases.append(("text ref L3 1 MiB frames", x[:m], O.ref_zstd_compress(x[:m].tobytes(), 3, piece=1 << 20)))
y = O.corpus("silesia-like", m)
cases.append(("silesia ref L3 one frame", y, O.ref_zstd_compress(y.tobytes(), 3)))
cases.append(("silesia ref L19 one frame", y[: 32 << 20], O.ref_zstd_compress(y[: 32 << 20].tobytes(), 19)))
z = np.zeros(m, dtype=np.uint8); z[::4097] = 7
cases.append(("sparse runs ref L3 one frame", z, O.ref_zstd_compress(z.tobytes(), 3)))
for name, want, comp in cases:
    comp = np.ascontiguousarray(np.frombuffer(bytes(comp), dtype=np.uint8))
    line = "%-30s" % name
    for wide in (0, 1):
        os.environ["GC_ZD_WIDE"] = str(wide)
        dec = pkg.ZstdDecoder(device=0)
        frames, nf, total = dec.scan(comp)
        d_src = torch.from_numpy(comp).cuda(); d_dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        best = 1e9
        for it in range(3):
            dec.code_device(d_src.data_ptr(), comp.size, d_dst.data_ptr(), total, frames, nf)
            if dec.last_timing_ms() < best: best = dec.last_timing_ms(); k = dec.kernel_timing_ms()
        ok = bool((d_dst[:total].cpu().numpy() == want[:total]).all())
        line += " | %s frames=%4d %8.2f ms %6.2f GB/s exec %7.2f ms rounds %2d ok=%s" % ("wide" if wide else "frame", nf, best, total / best / 1e6, k["execution"], dec.wide_rounds(), ok)
        dec.close(); del d_src, d_dst
    print(line, flush=True)
