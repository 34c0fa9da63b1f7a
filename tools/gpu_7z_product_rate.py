"""GPU box: the product-level numbers (SURVEY 8d ii): wall time of the reference's own `7z a` / `7z x` on a 1 GB file with the plugin's GPU
encoder / decoder against its built-in CPU codecs.  File IO (tmpfs), the host's CRC and the 7z container are inside every number."""
import os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
import oracle as O
g.build_hip(); module = g.build_plugin()
HOST = os.path.join(ROOT, "oracle", "_ref", "host7z")
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
base = "/dev/shm/gc7z" if os.path.isdir("/dev/shm") else "/tmp/gc7z"
shutil.rmtree(base, ignore_errors=True); os.makedirs(base)
hostbase = "/tmp/gc7z_host"; shutil.rmtree(hostbase, ignore_errors=True)      # (/dev/shm is mounted noexec: the binaries live in /tmp, the data in shared memory)
def install(name, bundle):
    d = os.path.join(hostbase, name); os.makedirs(os.path.join(d, "Codecs"))
    shutil.copy2(os.path.join(HOST, "7z"), d); shutil.copy2(os.path.join(HOST, bundle), os.path.join(d, "7z.so"))
    shutil.copy2(module, os.path.join(d, "Codecs"))
    return d
full, nozstd = install("full", "7z.so"), install("nozstd", "7z_nozstd.so")
env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "7-zip-zstd_amd", "csrc") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
src = os.path.join(base, "enwik9_standin.bin"); O.corpus("text-zipf", n).tofile(src)
def run(d, *a):
    t = time.perf_counter(); r = subprocess.run([os.path.join(d, "7z")] + list(a), capture_output=True, text=True, env=env, cwd=d); t = time.perf_counter() - t
    assert r.returncode == 0 and "Everything is Ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    return t
rows = []
for label, d, method, extra in (("GPU encoder (plugin, -m0=ZSTDGPU -mx3)", full, "ZSTDGPU", []), ("reference CPU encoder (-m0=zstd -mx3, all threads)", full, "zstd", ["-mmt=on"]),
                                ("reference CPU encoder (-m0=zstd -mx3 -mmt=1)", full, "zstd", ["-mmt=1"])):
    arc = os.path.join(base, "a_%s_%d.7z" % (method, len(rows)))
    run(d, "a", "-m0=" + method, "-mx3", *extra, arc, src)                      # warm (page cache, device context)
    os.remove(arc)
    t = run(d, "a", "-m0=" + method, "-mx3", *extra, arc, src)
    rows.append((label, t, os.path.getsize(arc), arc))
    print("7z a  %-55s %6.2f s = %6.2f GB/s  archive %d B" % (label, t, n / t / 1e9, os.path.getsize(arc)), flush=True)
gpu_arc = rows[0][3]
for label, d in (("GPU decoder (plugin, host without its own ZSTD codec)", nozstd), ("reference CPU decoder (built in)", full)):
    out = os.path.join(base, "x"); shutil.rmtree(out, ignore_errors=True)
    run(d, "x", "-o" + out, gpu_arc); shutil.rmtree(out)
    t = run(d, "x", "-o" + out, gpu_arc)
    ok = os.path.getsize(os.path.join(out, os.path.basename(src))) == n
    print("7z x  %-55s %6.2f s = %6.2f GB/s  ok=%s" % (label, t, n / t / 1e9, ok), flush=True)
    t = run(d, "t", gpu_arc)
    print("7z t  %-55s %6.2f s = %6.2f GB/s" % (label, t, n / t / 1e9), flush=True)
shutil.rmtree(base, ignore_errors=True)
