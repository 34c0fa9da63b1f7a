"""GPU: decode rate of the zstd decoder kernel on streams of this engine's encoder and of the reference's (inputs and outputs in HBM)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 256 << 20
dec = pkg.ZstdDecoder(device=0)
for kind in ("silesia-like", "text-zipf"):
    x = O.corpus(kind, n)
    for src, lvl in (("own", 3), ("own", 1), ("ref-1MiB-frames", 3), ("ref-one-frame", 3)):
        if src == "own":
            enc = pkg.ZstdEncoder(device=0, level=lvl); comp = enc.code(x); enc.close()
        elif src == "ref-1MiB-frames":
            comp = O.ref_zstd_compress(x[: 64 << 20].tobytes(), lvl, piece=1 << 20)
        else:
            comp = O.ref_zstd_compress(x[: 16 << 20].tobytes(), lvl)
        frames, nf, total = dec.scan(comp)
        d_src = torch.from_numpy(np.ascontiguousarray(comp)).cuda(); d_dst = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
        best = 1e9
        for it in range(3):
            size = dec.code_device(d_src.data_ptr(), comp.size, d_dst.data_ptr(), total, frames, nf)
            best = min(best, dec.last_timing_ms())
        ok = bool((d_dst[:total].cpu().numpy() == x[:total]).all())
        t0 = time.time(); O.ref_zstd_decompress(comp, total); cpu = time.time() - t0
        print("%-13s %-16s L%d frames=%4d content=%6.1f MiB kernel %8.2f ms = %6.2f GB/s (content)  ok=%s   reference decoder 1 core: %.2f GB/s" %
              (kind, src, lvl, nf, total / 2**20, best, total / best / 1e6, ok, total / cpu / 1e9), flush=True)
