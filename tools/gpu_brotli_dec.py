#!/usr/bin/env python3
"""GPU box: rate of the BROTLI decoder (gc_brotli_dec.hip) with the compressed stream resident in HBM, content compared on the device.
(a) config C5: 1 GB of web-text through THIS engine's encoder at quality 6 (1 590 brotli-mt chunks), (b) streams of the reference encoder (oracle/_ref, all host threads) at
qualities 1 / 6 / 9 on 256 MB and 11 on 32 MB, with the reference decoder timed on the same streams (all threads and one).
usage: python tools/gpu_brotli_dec.py [--bytes N] [--ref-bytes N]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O

ap = argparse.ArgumentParser(); ap.add_argument("--bytes", type=int, default=1_000_000_000); ap.add_argument("--ref-bytes", type=int, default=256 << 20); ap.add_argument("--corpus", default="web-text"); ap.add_argument("--only-own", action="store_true"); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--lib", default=None)
a = ap.parse_args()
THR = min(os.cpu_count() or 1, 64)
dec = pkg.BrotliDecoder(device=0, lib_path=a.lib)
have_ref = O.ref("brotli") is not None
if have_ref:
    dec.set_dictionary(O.ref_brotli_dictionary())

def run(label, comp, x, reps=None):
    reps = reps or a.reps
    chunks, n, cap, used = dec.scan(comp)
    d_c = torch.from_numpy(np.ascontiguousarray(comp)).cuda(); d_x = torch.from_numpy(x).cuda(); d_y = torch.empty(x.size + 64, dtype=torch.uint8, device="cuda")
    got = dec.code_device(d_c.data_ptr(), int(comp.size), d_y.data_ptr(), x.size, chunks, n)      # warm-up (workspace)
    ok = got == x.size and bool(torch.equal(d_y[:x.size], d_x))
    ms = []
    for _ in range(reps):
        d_y.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter(); dec.code_device(d_c.data_ptr(), int(comp.size), d_y.data_ptr(), x.size, chunks, n); wall = (time.perf_counter() - t0) * 1e3
        ms.append((dec.last_timing_ms(), wall))
    k = sum(m[0] for m in ms) / reps; w = sum(m[1] for m in ms) / reps
    out = {"stream": label, "content_bytes": int(x.size), "compressed_bytes": int(comp.size), "chunks": int(n), "bit_exact": ok, "kernel_ms": round(k, 3), "call_ms": round(w, 3),
           "GBps_content": round(x.size / k / 1e6, 2)}
    if have_ref:
        t0 = time.perf_counter(); y = O.ref_brotlimt_decompress(comp, x.size, THR); t1 = time.perf_counter() - t0
        out["reference_decoder_%d_threads_GBps" % THR] = round(x.size / t1 / 1e9, 2)
        m = min(comp.size, max(1, comp.size // max(1, n)) * 8); _, n8, cap8, used8 = dec.scan(comp[:m])
        if n8:
            t0 = time.perf_counter(); y = O.ref_brotlimt_decompress(comp[:used8], cap8, 1); t1 = time.perf_counter() - t0
            out["reference_decoder_1_thread_GBps"] = round(y.size / t1 / 1e9, 3)
    print(json.dumps(out), flush=True)
    del d_c, d_x, d_y

x = O.corpus(a.corpus, a.bytes)
e = pkg.BrotliEncoder(level=6); c = e.code(x); e.close()
run("this engine, quality 6, %s" % a.corpus, c, x)
for kind in (() if a.only_own else ("real-src", "real-bin")):
    xr = O.corpus(kind, 256 << 20)
    if xr.size >= (1 << 20):
        e = pkg.BrotliEncoder(level=6); c = e.code(xr); e.close()
        run("this engine, quality 6, %s" % kind, c, xr)
if have_ref and not a.only_own:
    for q, nb in ((1, a.ref_bytes), (6, a.ref_bytes), (9, a.ref_bytes), (11, min(a.ref_bytes, 32 << 20))):
        xr = x[:nb]
        c = O.ref_brotlimt_compress(xr, q, THR)
        run("reference encoder, quality %d, %s" % (q, a.corpus), c, xr, reps=3)
dec.close()
