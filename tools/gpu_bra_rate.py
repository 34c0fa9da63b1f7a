"""GPU: rate of the branch converter kernel (1 GB of instruction-like data in HBM, out of place) against the reference's C/Bra.c on one host core."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
from test_bra import _code_like, KID
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 30
for kind in ("ARM64", "ARM", "ARMT", "PPC", "SPARC", "IA64", "RISCV"):
    x = _code_like(kind, n, 3)
    d_in = torch.from_numpy(x).cuda(); d_out = torch.empty_like(d_in)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        pkg.bra_convert_device(kind, d_in.data_ptr(), d_out.data_ptr(), n, 0x400000, True)      # synchronous on the default stream
        best = min(best, (time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); want, _ = O.ref_bra_convert(KID[kind], x, 0x400000, True); cpu = time.perf_counter() - t0
    ok = bool(np.array_equal(d_out.cpu().numpy(), want))
    print("%-6s %d B: call %.3f ms = %.0f GB/s of input (%.0f GB/s read+write = %.1f %% of 8 TB/s); reference on one core %.2f GB/s; bit-exact %s"
          % (kind, n, best, n / best / 1e6, 2 * n / best / 1e6, 2 * n / best / 1e6 / 80.0, n / cpu / 1e9, ok), flush=True)
from test_bra import _x86_like
x = _x86_like(n, 3)
d_in = torch.from_numpy(x).cuda(); d_out = torch.empty_like(d_in)
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    done, st = pkg.bra_x86_convert_device(d_in.data_ptr(), d_out.data_ptr(), n, 0x400000, True, 0)
    best = min(best, (time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter(); want, wdone, wst = O.ref_bra_x86_convert(x, 0x400000, True, 0); cpu = time.perf_counter() - t0
ok = bool(np.array_equal(d_out.cpu().numpy(), want)) and (done, st) == (wdone, wst)
print("X86    %d B: call %.3f ms = %.0f GB/s of input (copy + scan: 3 B of traffic per byte = %.1f %% of 8 TB/s); reference on one core %.2f GB/s; bit-exact %s"
      % (n, best, n / best / 1e6, 3 * n / best / 1e6 / 80.0, n / cpu / 1e9, ok), flush=True)
rng = np.random.default_rng(2); x = rng.integers(0, 256, size=n, dtype=np.uint8)
d_in = torch.from_numpy(x).cuda(); d_out = torch.empty_like(d_in); torch.cuda.synchronize()
for delta in (1, 4, 100):
    for enc in (True, False):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); pkg.delta_convert_device(d_in.data_ptr(), d_out.data_ptr(), n, delta, enc); best = min(best, (time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); want, _ = O.ref_delta_convert(x, delta, enc); cpu = time.perf_counter() - t0
        print("DELTA %3d %s %d B: call %.3f ms = %.0f GB/s of input; reference on one core %.2f GB/s; bit-exact %s"
              % (delta, "enc" if enc else "dec", n, best, n / best / 1e6, n / cpu / 1e9, bool(np.array_equal(d_out.cpu().numpy(), want))), flush=True)
