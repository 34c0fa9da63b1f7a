"""GPU: the metric's own stream (zstd level 3 of the 1 GB enwik9 stand-in) through the decoder kernels, three times (for rocprofv3 passes)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
x = O.corpus("text-zipf", n)
enc = pkg.ZstdEncoder(device=0, level=3); comp = enc.code(x); enc.close()
dec = pkg.ZstdDecoder(device=0)
frames, nf, total = dec.scan(comp)
d_src = torch.from_numpy(np.ascontiguousarray(comp)).cuda(); d_dst = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
for _ in range(3):
    dec.code_device(d_src.data_ptr(), comp.size, d_dst.data_ptr(), total, frames, nf)
    print("decode kernels %.2f ms (%d frames, %d -> %d bytes)" % (dec.last_timing_ms(), nf, comp.size, total), flush=True)
print("bit-exact", bool(torch.equal(d_dst[:total], torch.from_numpy(x).cuda())))
