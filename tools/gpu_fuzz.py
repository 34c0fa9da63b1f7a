#!/usr/bin/env python3
"""Round-trip fuzz of the three codecs on the GPU (or, with --lib, the emulator build): inputs stitched from random, text, runs, PCM-like and
repeated segments of ragged lengths, random levels, sizes from a few bytes to tens of MiB; every stream must decode under the reference
decoder.  ZSTD streams additionally go through the GPU decoder (this engine's stream and a reference-encoder stream of a random level, which must
both give the input back), and a damaged copy of the stream must be refused or decode to the same content -- never hang or crash.  Failing inputs are saved under gpurun_out/fuzz/.   usage: python tools/gpu_fuzz.py --seconds 60 [--max-mib 24] [--lib path]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import __graft_entry__ as g
import oracle as O
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60); ap.add_argument("--max-mib", type=float, default=24); ap.add_argument("--lib", default=""); ap.add_argument("--seed", type=int, default=7)
a = ap.parse_args()
pkg = g.load_package(); kw = {"lib_path": a.lib} if a.lib else {"device": 0}
if a.lib and os.environ.get("GC_FUZZ_DEVICE"): kw["device"] = int(os.environ["GC_FUZZ_DEVICE"])      # a device library other than the shipped one (the hooks build)
rng = np.random.default_rng(a.seed)
text = O.corpus("text-zipf", 4 << 20); lz = O.corpus("lz-7zip", 4 << 20); sil = O.corpus("silesia-like", 211_900_000)[31_785_000:31_785_000 + (4 << 20)]   # PCM-like part
os.makedirs(os.path.join(ROOT, "gpurun_out", "fuzz"), exist_ok=True)
t0 = time.time(); it = bad = 0; encs = {}; total = 0; dec = pkg.ZstdDecoder(**kw); ndec = nrefused = 0
bdec = pkg.BrotliDecoder(**kw); nbdec = nbref = 0
if O.ref("brotli") is not None: bdec.set_dictionary(O.ref_brotli_dictionary())
while time.time() - t0 < a.seconds:
    target = int(min(a.max_mib * (1 << 20), 2 ** rng.uniform(3, 25)))
    parts, n = [], 0
    while n < target:
        L = int(min(target - n, rng.choice([1, 7, 1000, 4095, 4096, 4097, 16385, 65536, 131071, 131072, 131073, 1 << 20, 3 << 20])))
        k = int(rng.integers(0, 7))
        if k == 0: p = rng.integers(0, 256, L, dtype=np.uint8)
        elif k == 1: o = int(rng.integers(0, text.size - L)); p = text[o:o + L]
        elif k == 2: p = np.full(L, int(rng.integers(0, 256)), dtype=np.uint8)
        elif k == 3 and parts: q = parts[int(rng.integers(0, len(parts)))]; p = np.resize(q, L)
        elif k == 4: o = int(rng.integers(0, lz.size - L)); p = lz[o:o + L]
        elif k == 5: o = int(rng.integers(0, sil.size - L)); p = sil[o:o + L]
        else: p = np.tile(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8), L // 1 + 1)[:L]
        parts.append(np.ascontiguousarray(p)); n += L
    x = np.ascontiguousarray(np.concatenate(parts)) if parts else np.empty(0, dtype=np.uint8); n = x.size
    codec = ["zstd", "flzma2", "brotli"][it % 3]
    level = int(rng.choice({"zstd": [1, 2, 3, 5, 6, 7, 9, 10, 12, 16, 18, 19, 22], "flzma2": [1, 2, 3, 4, 5, 7, 9], "brotli": [1, 2, 3, 4, 5, 6, 8, 9, 10, 11]}[codec]))
    key = (codec, level)
    if key not in encs:
        encs[key] = {"zstd": pkg.ZstdEncoder, "flzma2": pkg.Flzma2Encoder, "brotli": pkg.BrotliEncoder}[codec](level=level, **kw)
    e = encs[key]
    try:
        y = e.code(x)
        z = O.ref_zstd_decompress(y, n) if codec == "zstd" else O.ref_lzma2_decode(y, n, e.coder_props()[0]) if codec == "flzma2" else O.ref_brotlimt_decompress(y, n, 8)
        ok = np.array_equal(np.asarray(z), x)
        if ok and codec == "zstd":
            ok = np.array_equal(dec.code(y, capacity=n + 64), x)
            if ok and n <= (8 << 20):
                opts = dict(checksum=bool(rng.integers(0, 2)), streamed=bool(rng.integers(0, 2)), ldm=bool(rng.integers(0, 4) == 0))
                r = O.ref_zstd_compress_opts(x.tobytes(), int(rng.choice([1, 2, 3, 5, 8, 13, 17, 19, 22])) if n <= (1 << 20) else int(rng.choice([1, 3, 5])), **opts)
                ok = np.array_equal(dec.code(r, capacity=n + 64), x)
                ndec += 1
                if ok and r.size > 12:
                    badr = r.copy(); badr[int(rng.integers(4, r.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
                    try:
                        w = dec.code(badr, capacity=n + 64)
                        ok = (not opts["checksum"]) or np.array_equal(w, x)     # without a checksum a flip may go unnoticed by the format
                    except pkg.GpuCodecError:
                        nrefused += 1
        if ok and codec == "brotli":
            # ... and BROTLI streams through the GPU decoder: this engine's, a reference-encoder stream of a random quality (with its references to the static dictionary), damaged copies
            ok = np.array_equal(bdec.code(y), x)
            if ok and n <= (8 << 20):
                r = O.ref_brotlimt_compress(x, int(rng.choice([0, 1, 2, 4, 5, 6, 9, 10, 11])) if n <= (1 << 20) else int(rng.choice([1, 5, 6])), int(rng.integers(1, 5)))
                ok = np.array_equal(bdec.code(r), x)
                nbdec += 1
                for src_stream in (r, y):
                    if ok and src_stream.size > 20:
                        badr = src_stream.copy(); k = int(rng.integers(0, 3))
                        if k == 0: badr[int(rng.integers(16, badr.size))] ^= np.uint8(1 << int(rng.integers(0, 8)))
                        elif k == 1: i = int(rng.integers(16, badr.size)); badr[i:i + 8] = rng.integers(0, 256, size=badr[i:i + 8].size, dtype=np.uint8)
                        else: badr = badr[:int(rng.integers(17, badr.size))]
                        try:
                            bdec.code(badr)                                   # brotli carries no checksum: other content is not a failure, a crash or a hang is
                        except pkg.GpuCodecError:
                            nbref += 1
                        ok = np.array_equal(bdec.code(y), x)                   # the context still works
    except Exception as ex:
        ok = False; print("EXC", codec, level, n, repr(ex)[:200], flush=True)
    if not ok:
        bad += 1; print("FAIL", codec, level, n, flush=True); np.save(os.path.join(ROOT, "gpurun_out", "fuzz", "fail_%s_%d_%d.npy" % (codec, level, n)), x)
    it += 1; total += n
print("iterations", it, "bytes", total, "failures", bad, "| reference streams through the GPU decoder", ndec, "damaged copies refused", nrefused, "| brotli: reference streams through the GPU decoder", nbdec, "damaged copies refused", nbref)
