"""GPU: the two sequences kernels of the zstd decoder (one block per wave / several blocks per wave, hook GC_ZD_SEQV) side by side."""
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
enc = pkg.ZstdEncoder(device=0, level=1); own1 = enc.code(x[: 256 << 20]); enc.close()
y = O.corpus("silesia-like", 128 << 20)
cases = [("text own L3 1 GB", x, own), ("text own L1 256 MiB", x[: 256 << 20], own1), ("silesia ref L3 one frame 128 MiB", y, O.ref_zstd_compress(y.tobytes(), 3)),
         ("silesia ref L19 one frame 32 MiB", y[: 32 << 20], O.ref_zstd_compress(y[: 32 << 20].tobytes(), 19)),
         ("text ref L3 1 MiB frames 128 MiB", x[: 128 << 20], O.ref_zstd_compress(x[: 128 << 20].tobytes(), 3, piece=1 << 20))]
for name, want, comp in cases:
    comp = np.ascontiguousarray(np.frombuffer(bytes(comp), dtype=np.uint8)).copy()
    for v in (0, 1):
        os.environ["GC_ZD_SEQV"] = str(v)
        dec = pkg.ZstdDecoder(device=0)
        frames, nf, total = dec.scan(comp)
        d_src = torch.from_numpy(comp).cuda(); d_dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        best = 1e9
        for it in range(3):
            dec.code_device(d_src.data_ptr(), comp.size, d_dst.data_ptr(), total, frames, nf)
            if dec.last_timing_ms() < best: best = dec.last_timing_ms(); k = dec.kernel_timing_ms()
        ok = bool((d_dst[:total].cpu().numpy() == want[:total]).all())
        print("%-34s seqv %d: %7.2f ms = %6.2f GB/s  literals %6.2f sequences %6.2f execution %6.2f ms rounds %d ok=%s" % (name, v, best, total / best / 1e6, k["literals"], k["sequences"], k["execution"], dec.wide_rounds(), ok), flush=True)
        dec.close(); del d_src, d_dst
