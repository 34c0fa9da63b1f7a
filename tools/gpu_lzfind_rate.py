#!/usr/bin/env python3
"""Rate of the device match finders (csrc/gc_lzfind.hip) against the reference's C/LzFind.c on one host core, same buffer, same parameters,
and a value-for-value comparison of the lists.  usage: python tools/gpu_lzfind_rate.py [--bytes N] [--corpus K] [--cut C] [--nice L]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import __graft_entry__ as g
import oracle as O
ap = argparse.ArgumentParser()
ap.add_argument("--bytes", type=int, default=32 << 20)
ap.add_argument("--corpus", default="text-zipf")
ap.add_argument("--cut", type=int, default=32)
ap.add_argument("--nice", type=int, default=64)
ap.add_argument("--history", type=int, default=1 << 24)
a = ap.parse_args()
g.build_hip(); pkg = g.load_package()
x = O.corpus(a.corpus, a.bytes)
stride = 2 * (min(a.cut, a.nice) + 2)
d = torch.from_numpy(x).cuda()
dc = torch.zeros(x.size, dtype=torch.int32, device="cuda"); dp = torch.zeros(x.size * stride, dtype=torch.int32, device="cuda")
for bt in (False, True):
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t = time.time()
        pkg.lzfind_get_matches_device(d.data_ptr(), x.size, dc.data_ptr(), dp.data_ptr(), stride, a.history, bt, a.cut, a.nice)
        torch.cuda.synchronize(); ts.append(time.time() - t)
    counts = dc.cpu().numpy().view(np.uint32)
    out = {"finder": "BT4" if bt else "HC4", "corpus": a.corpus, "bytes": a.bytes, "cut": a.cut, "nice": a.nice, "history": a.history,
           "gpu_s": round(min(ts), 4), "gpu_MBps": round(a.bytes / min(ts) / 1e6, 1), "values": int(counts.sum())}
    if O.ref("lzfind") is not None:
        m = min(a.bytes, 8 << 20)                                  # the reference on one core, on the first 8 MiB
        t = time.time(); c1, p1 = O.ref_lzfind_matches(x[:m], a.history, bt, 4, a.cut, a.nice); dt = time.time() - t
        c2, p2 = pkg.lzfind_matches(x[:m], a.history, bt, a.cut, a.nice, device=0)
        out.update({"ref_1core_MBps": round(m / dt / 1e6, 1), "ref_sample_bytes": m, "identical_to_reference": bool(np.array_equal(c1, c2) and np.array_equal(p1, p2))})
    print(json.dumps(out), flush=True)
