#!/usr/bin/env python3
"""Diagnosis (round 6; the cause was the converters' result read back from stream-ordered pool memory, see gc_host_stream.h): `7z a -m0=BCJGPU -m1=ZSTDGPU` through the product module failed its CRC once on a fresh box (run final5) and on the first of twelve repeats.
Repeats the chain and its halves (filter alone in front of Copy, encoder alone) from a cold start and, when the host's own decoders report an error, extracts
and says where the bytes differ.   usage: python tools/gpu_diag_bcj.py [repeats]"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
from test_bra import _x86_like
HOST = os.path.join(ROOT, "oracle", "_ref", "host7z")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g.build_hip(); module = g.build_plugin()
d = tempfile.mkdtemp(prefix="diagbcj")
for f in ("7z", "7z.so"): shutil.copy2(os.path.join(HOST, f), os.path.join(d, f))
os.mkdir(os.path.join(d, "Codecs")); shutil.copy2(module, os.path.join(d, "Codecs", os.path.basename(module)))
env = dict(os.environ); env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "7-zip-zstd_amd", "csrc") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
def run(*a): return subprocess.run([os.path.join(d, "7z")] + list(a), capture_output=True, text=True, env=env, cwd=d)
x = _x86_like(40_000_000, 3); x.tofile(os.path.join(d, "in.bin"))
for it in range(reps):
    for tag, methods in (("bcj+zstd", ["-m0=BCJGPU", "-m1=ZSTDGPU", "-mx3"]), ("bcj+copy", ["-m0=BCJGPU", "-m1=Copy"])):
        arc = os.path.join(d, "a.7z")
        if os.path.exists(arc): os.remove(arc)
        t0 = time.time(); r = run("a", *methods, "a.7z", "in.bin"); ta = time.time() - t0
        t = run("t", "a.7z")
        ok = r.returncode == 0 and t.returncode == 0
        print("iteration %d %-9s add %.2fs rc %d test rc %d %s" % (it, tag, ta, r.returncode, t.returncode, "ok" if ok else "FAILED: " + (t.stderr.strip() or r.stderr.strip())[:120]), flush=True)
        if not ok:
            out = os.path.join(d, "x"); shutil.rmtree(out, ignore_errors=True)
            run("x", "-o" + out, "a.7z")
            p = os.path.join(out, "in.bin")
            if os.path.exists(p):
                y = np.fromfile(p, dtype=np.uint8)
                m = min(x.size, y.size); diff = np.nonzero(x[:m] != y[:m])[0]
                print("   sizes %d / %d, differing bytes %d" % (x.size, y.size, diff.size), flush=True)
                if diff.size:
                    gaps = np.diff(diff); runs = np.concatenate([[0], np.nonzero(gaps > 64)[0] + 1])
                    print("   first %d last %d; %d separate places; starts %s" % (diff[0], diff[-1], runs.size, [int(diff[r_]) for r_ in runs[:12]]), flush=True)
                    a = int(diff[0]); print("   original %s\n   got      %s" % (x[a - 4:a + 12].tobytes().hex(), y[a - 4:a + 12].tobytes().hex()), flush=True)
shutil.rmtree(d, ignore_errors=True)
