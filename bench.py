#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: compression MB/s + size against the reference for
    (a) zstd level 3 on the enwik9 stand-in (1 000 000 000 B of `text-zipf`)            -> the top-level fields of the JSON line
    (b) Fast-LZMA2 level 5 on the Silesia stand-in (211 900 000 B of `silesia-like`)    -> the object "flzma2_l5_silesia"
at 1 / 2 / 4 / 8 GPUs.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

STRONG scaling: the corpus is fixed; the host range-splits it at the codec's independence grain (8 MiB match-finder frames;
`sharding.shard_ranges`), rank r compresses range r on GPU r, and the compressed ranges concatenated in rank order are ONE valid
stream (zstd frames / LZMA2 chunk runs with a single end marker).  There is no data-path collective: RCCL carries the timing
barrier + max-reduction inside the timed region, and -- outside it -- the gather of the compressed bytes to rank 0, where the
concatenation is decoded by the reference's own decoder (oracle/_ref, test infrastructure) and compared with the corpus.

A "step" is one pass of the whole hot path (match finder -> entropy stage -> framing) over the rank's range, input resident in
HBM before the timed region starts, output left in HBM.  K steps are timed per codec between barrier + synchronize pairs; the
maximum over ranks counts.  `value` = corpus bytes * K / that time.  One JSON line is printed by rank 0 (DESIGN.md section 6
defines every field).  `--codec brotli|zstd|flzma2` with `--bytes/--level/--corpus` runs ONE codec on a chosen workload
(configs C4 / C5 and experiments); the default run is the metric.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

ENWIK9_BYTES = 1_000_000_000
SILESIA_BYTES = 211_900_000
DEFAULTS = {"zstd": (3, "text-zipf", ENWIK9_BYTES), "flzma2": (5, "silesia-like", SILESIA_BYTES), "brotli": (6, "web-text", ENWIK9_BYTES)}


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O          # cpu_baseline + decode-check legs only
    return O


def cpu_baseline(codec, x, level):
    """The reference codec (oracle/_ref = C/zstd, C/fast-lzma2, C/brotli compiled from /root/reference) on this box's host cores.
    Returns (cpu_baseline object, reference size of the WHOLE corpus or None).  Bounded: about 10-20 s of CPU work per codec."""
    O = _oracle()
    if O.ref(codec) is None:
        return None, None
    cores = os.cpu_count() or 1
    if codec == "zstd":
        # all threads on the whole corpus (ZSTDMT), one thread on the whole corpus (nbWorkers=0: the stricter size reference)
        t0 = time.perf_counter(); cm = O.ref_zstd_compress(x, level, workers=cores); tm = time.perf_counter() - t0
        one = x if x.size <= 1_000_000_000 else x[:1_000_000_000]
        t0 = time.perf_counter(); c1 = O.ref_zstd_compress(one, level, workers=0); t1 = time.perf_counter() - t0
        return ({"value": round(x.size / tm / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "reference",
                 "sample": "ZSTD_compress2 level %d on the whole %d-byte corpus, one run, %d threads (ZSTDMT nbWorkers=%d: %d B); single thread "
                           "(nbWorkers=0, %d bytes): %.1f MB/s" % (level, x.size, cores, cores, len(cm), one.size, one.size / t1 / 1e6)},
                len(c1) if one.size == x.size else None)
    if codec == "flzma2":
        thr = min(cores, 64)
        t0 = time.perf_counter(); cm, _ = O.ref_fl2_compress(x, level, threads=thr); tm = time.perf_counter() - t0
        one = x[: min(x.size, 16 * 1024 * 1024)]
        t0 = time.perf_counter(); O.ref_fl2_compress(one, level, threads=1); t1 = time.perf_counter() - t0
        return ({"value": round(x.size / tm / 1e6, 1), "unit": "MB/s", "cores": thr, "kind": "reference",
                 "sample": "FL2_compressCCtx level %d on the whole %d-byte corpus, one run, %d threads; single thread (first %d bytes): %.1f MB/s"
                           % (level, x.size, thr, one.size, one.size / t1 / 1e6)}, len(cm))
    thr = min(cores, 128)           # BROTLIMT_THREAD_MAX
    sample = x[: min(x.size, 256 * 1024 * 1024)]
    one = sample[: 8 * 1024 * 1024]
    t0 = time.perf_counter(); O.ref_brotlimt_compress(one, level, 1); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); cm = O.ref_brotlimt_compress(sample, level, thr); tm = time.perf_counter() - t0
    return ({"value": round(sample.size / tm / 1e6, 1), "unit": "MB/s", "cores": thr, "kind": "reference",
             "sample": "BROTLIMT_compressCCtx quality %d on the first %d bytes of the corpus, one run, %d threads; single thread (8 MiB): %.1f MB/s"
                       % (level, sample.size, thr, one.size / t1 / 1e6)}, (len(cm), sample.size))


def _pmc_lookup(section, n, kname):
    """(hbm bytes per launch, commit stamp, kernel name as profiled) from profiles/pmc_traffic.json for a kernel of `section` measured on a workload
    of n bytes, else (None, None, kname).  The finder's kernels carry the suffix _p8 in the fast geometry, and at zstd level 3 verify + parse are
    one fused kernel (gc_mf_vparse_tile_kernel): whichever of the names the passes saw is the one that ran."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pm.get("_workload_bytes", {}).get(section) != n:
            return None, None, kname
        sec = pm.get(section, {})
        cands = [kname + "_p8", kname]
        if kname == "gc_mf_verify_kernel":
            cands = ["gc_mf_vparse_tile_kernel_p8", "gc_mf_vparse_tile_kernel"] + cands
        for k in cands:
            if k in sec:
                return sec[k]["hbm_bytes_per_launch"], sec.get("_commit"), k
    except Exception:
        pass
    return None, None, kname


def real_data_check(pkg, device):
    """Outside every timed region, N = 1 only: the two codecs of the metric on REAL bytes from the image (7-zip-zstd_amd/corpus real_corpus: C / C++ / Python
    sources, ROCm shared objects) -- the stand-in corpora above are generators.  Sizes against the reference encoders at the same levels on 64 MiB, and
    THROUGHPUT on the real bytes tiled to the metric's sizes (frames are independent, so repetition across frames is of no help to the finder): zstd level 3
    on 1 GB, Fast-LZMA2 level 5 on 211.9 MB, data resident in HBM, three steps after a warm-up, with the kernels' own total."""
    import time
    import numpy as np
    import torch
    O = _oracle()
    out = {"note": "sizes: zstd on 64 MiB per corpus, Fast-LZMA2 on config C3's 211.9 MB of it (or all there is), the reference with up to 64 threads, ours_over_ref against the 2 % band; MBps: the corpus tiled to the metric's size, "
                   "resident in HBM, mean of 3 steps after a warm-up (kernel_ms: the library's own first-kernel-start to last-kernel-end)", "corpora": {}}

    def rate(enc, x, fl2):
        d_src = torch.from_numpy(x).to("cuda:%d" % device)
        cap = enc.compress_bound(x.size)
        d_dst = torch.empty(cap, dtype=torch.uint8, device="cuda:%d" % device)
        enc.code_device(d_src.data_ptr(), x.size, d_dst.data_ptr(), cap); n = enc.finish()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            enc.code_device(d_src.data_ptr(), x.size, d_dst.data_ptr(), cap); n = enc.finish()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3.0
        kms = enc.last_timing_ms()
        del d_src, d_dst
        return {"bytes": int(x.size), "compressed_bytes": int(n), "MBps": round(x.size / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 3), "kernel_ms": round(float(kms["total"]), 3)}

    try:
        thr = min(os.cpu_count() or 1, 64)
        for kind in ("real-src", "real-bin"):
            x = O.corpus(kind, 64 << 20)
            if x.size < (1 << 20):
                continue
            row = {"bytes": int(x.size)}
            full = O.corpus(kind, SILESIA_BYTES)
            tile = lambda n: np.ascontiguousarray(np.resize(full[: full.size - full.size % (8 << 20)] if full.size >= (8 << 20) else full, n))     # whole 8 MiB frames repeated
            if O.ref("zstd") is not None:
                e = pkg.ZstdEncoder(level=3, device=device); c = e.code(x)
                r = O.ref_zstd_compress(x, 3)
                row["zstd_l3"] = {"ours": int(len(c)), "ref": int(len(r)), "ours_over_ref": round(len(c) / len(r), 4), "decodes": bool(np.array_equal(O.ref_zstd_decompress(c, x.size), x))}
                row["zstd_l3"]["throughput"] = rate(e, tile(ENWIK9_BYTES), False); e.close()
            if O.ref("flzma2") is not None:
                # (config C3's size, not 64 MiB: on shared objects the first 64 MiB flatter the engine -- round 4: 1.020 there, 1.026 on all 211.9 MB)
                e = pkg.Flzma2Encoder(level=5, device=device); c = e.code(full); prop = e.coder_props()[0]
                r, _ = O.ref_fl2_compress(full, 5, threads=thr)
                row["flzma2_l5"] = {"bytes": int(full.size), "ours": int(len(c)), "ref": int(len(r)), "ours_over_ref": round(len(c) / len(r), 4), "decodes": bool(np.array_equal(O.ref_lzma2_decode(c, full.size, prop), full))}
                row["flzma2_l5"]["throughput"] = rate(e, tile(SILESIA_BYTES), True); e.close()
            out["corpora"][kind] = row
    except Exception as ex:                      # (a report, never a reason for the bench line to be missing)
        out["error"] = repr(ex)
    return out


def run_codec(codec, level, corpus_name, total, args, env):
    """K timed steps of one codec over this rank's range of the corpus; rank 0 returns the result object."""
    import numpy as np
    import torch
    rank, world, dist, dev, pkg, corpus_mod, S = env["rank"], env["world"], env["dist"], env["dev"], env["pkg"], env["corpus"], env["sharding"]
    fl2, br = codec == "flzma2", codec == "brotli"
    x_all = corpus_mod.corpus(corpus_name, total, seed=20260921)              # the same corpus on every rank
    grain = S.codec_grain(codec, level)
    if not br and total < world * grain:
        grain = S.GRAIN_ZSTD
    s, e = S.shard_ranges(total, world, grain)[rank]
    n = e - s
    enc = (pkg.Flzma2Encoder if fl2 else (pkg.BrotliEncoder if br else pkg.ZstdEncoder))(device=env["local_rank"], level=level)
    d_src = torch.from_numpy(x_all[s:e]).to(dev) if n else torch.empty(1, dtype=torch.uint8, device=dev)
    cap = enc.compress_bound(n) + 16
    d_dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    code = (lambda: enc.code_device(d_src.data_ptr(), n, d_dst.data_ptr(), cap, enc.NO_END_MARK)) if fl2 else \
           (lambda: enc.code_device(d_src.data_ptr(), n, d_dst.data_ptr(), cap))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    csize = 0
    for _ in range(args.warmup if n else 0):
        code(); csize = enc.finish()
    kern_ms = {k: 0.0 for k in enc.KERNELS}
    mf_ms = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps if n else 0):
        code(); csize = enc.finish()
        t = enc.last_timing_ms()         # hipEvent pairs recorded on the library's own streams around each kernel
        for k in kern_ms:
            kern_ms[k] += t[k]
        mf = enc.mf_timing_ms()          # stage durations of the windowed match finder (None: block-local finder ran)
        if mf:
            for k, v in mf.items():
                mf_ms[k] = mf_ms.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    # ---- outside the timed region: max over ranks, gather of the compressed ranges, decode of the concatenation
    sizes = [csize]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        sz = torch.tensor([csize], dtype=torch.int64, device=dev)
        all_sz = [torch.zeros_like(sz) for _ in range(world)]
        dist.all_gather(all_sz, sz)
        sizes = [int(v.item()) for v in all_sz]
        pad = max(sizes) + 1
        mine = torch.zeros(pad, dtype=torch.uint8, device=dev); mine[:csize] = d_dst[:csize]
        parts = [torch.empty(pad, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.all_gather(parts, mine)
        stream = np.concatenate([p[:z].cpu().numpy() for p, z in zip(parts, sizes)]) if rank == 0 else None
    else:
        stream = d_dst[:csize].cpu().numpy()
    if rank != 0:
        enc.close()
        return None
    if fl2:
        stream = np.concatenate([stream, np.zeros(1, dtype=np.uint8)])         # the single LZMA2 end marker
    total_csize = int(stream.size)
    for k in kern_ms:
        kern_ms[k] /= max(args.steps, 1)
    for k in mf_ms:
        mf_ms[k] /= max(args.steps, 1)
    decodes = None
    gpu_decode = None
    if not args.no_decode_check:
        O = _oracle()
        if O.ref(codec) is not None:
            thr = min(os.cpu_count() or 1, 64)
            td = time.perf_counter()
            y = O.ref_zstd_decompress(stream, total) if codec == "zstd" else \
                O.ref_lzma2_decode(stream, total, enc.coder_props()[0]) if fl2 else O.ref_brotlimt_decompress(stream, total, thr)
            td = time.perf_counter() - td
            decodes = bool(np.array_equal(y, x_all))
            del y
            if codec == "zstd" and world == 1:
                # SURVEY.md 8f1, outside the timed region: the same stream through the GPU decoder (compressed bytes and content in HBM), content
                # compared on the device; beside it the reference's decoder on one host core (its 7-Zip decoder is single-threaded per stream)
                dec = pkg.ZstdDecoder(device=env["local_rank"])
                frames, nf, content = dec.scan(stream)
                d_c = torch.from_numpy(np.ascontiguousarray(stream)).to(dev)
                d_y = torch.empty(total + 64, dtype=torch.uint8, device=dev)
                DEC_RUNS = 5                                         # one untimed warm-up (workspaces grow), then the MEAN of DEC_RUNS decodes
                best, kms, rounds = 0.0, {}, 0
                for it in range(DEC_RUNS + 1):
                    got = dec.code_device(d_c.data_ptr(), int(stream.size), d_y.data_ptr(), total, frames, nf)
                    if it == 0:
                        continue
                    best += dec.last_timing_ms() / DEC_RUNS
                    for k, v in dec.kernel_timing_ms().items():
                        kms[k] = kms.get(k, 0.0) + v / DEC_RUNS
                    rounds = dec.wide_rounds()
                same = bool(got == total and torch.equal(d_y[:total], d_src[:total]))
                dom = max(kms, key=lambda k: kms[k])
                nblk = sum(int(frames[i].n_blocks) for i in range(nf))
                kname = {"execution": "gc_zstd_dec_chase_kernel" if rounds else "gc_zstd_dec_exec_kernel", "literals": "gc_zstd_dec_lit_kernel", "index": "gc_zstd_dec_index_kernel",
                         "sequences": "gc_zstd_dec_seqv_kernel" if nblk >= 1024 else "gc_zstd_dec_seq_kernel"}[dom]      # (gc_api.hip: six blocks per wave from 1024 blocks on)
                algo = total + int(stream.size)                      # compressed stream read once + content written once
                dtraffic, dcommit, kname = _pmc_lookup("zstd_dec", total, kname)
                gpu_decode = {"frames": nf, "content_bytes": total, "timing": "mean of %d decodes after one warm-up" % DEC_RUNS, "kernel_ms": round(best, 3), "value": round(total / best / 1e3, 1), "unit": "MB/s of content",
                              "bit_exact": same, "reference_decoder_1_core_MBps": round(total / td / 1e6, 1),
                              "kernels_ms": {k: round(v, 3) for k, v in kms.items()},
                              "execution": ("wide: place + spread + %d pointer-jumping rounds + finish over all blocks at once" % rounds) if rounds else "one workgroup per frame, blocks in order",
                              "roofline": {"bound": "hbm", "kernel": kname,
                                           "achieved": round(algo / (kms[dom] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(algo / (kms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": dtraffic, "traffic_measured_at_commit": dcommit, "algorithmic_bytes_per_launch": algo}}
                dec.close(); del d_c, d_y
            if br and world == 1:
                # the same for BROTLI (gc_brotli_dec.hip: one wave per brotli-mt chunk; this engine's streams never refer to the static dictionary)
                dec = pkg.BrotliDecoder(device=env["local_rank"])
                chunks, nch, _, _ = dec.scan(stream)
                d_c = torch.from_numpy(np.ascontiguousarray(stream)).to(dev)
                d_y = torch.empty(total + 64, dtype=torch.uint8, device=dev)
                DEC_RUNS = 3
                best = 0.0
                for it in range(DEC_RUNS + 1):
                    got = dec.code_device(d_c.data_ptr(), int(stream.size), d_y.data_ptr(), total, chunks, nch)
                    if it:
                        best += dec.last_timing_ms() / DEC_RUNS
                same = bool(got == total and torch.equal(d_y[:total], d_src[:total]))
                algo = total + int(stream.size)
                gpu_decode = {"chunks": nch, "content_bytes": total, "timing": "mean of %d decodes after one warm-up" % DEC_RUNS, "kernel_ms": round(best, 3), "value": round(total / best / 1e3, 1), "unit": "MB/s of content",
                              "bit_exact": same, "reference_decoder_%d_threads_MBps" % thr: round(total / td / 1e6, 1),
                              "parallelism": "one wave per brotli-mt chunk (%d chunks): inside a chunk the stream is serial by format" % nch,
                              "roofline": {"bound": "hbm", "kernel": "gc_brotli_dec_kernel_" + ("a" if nch <= 256 else "b" if nch <= 512 else "c" if nch <= 1024 else "d"),
                                           "achieved": round(algo / (best * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                           "traffic": None, "algorithmic_bytes_per_launch": algo}}
                dec.close(); del d_c, d_y
    value = total * args.steps / elapsed / 1e6
    ratio = total / total_csize
    # dominant kernel of rank 0 = the longest single kernel by live HIP-event timing; algorithmic bytes per launch =
    # (1 + 1/ratio) bytes per input byte x the bytes one launch of it processes (this rank's range)
    per_kernel = {k: v for k, v in kern_ms.items() if k != "total"}
    if mf_ms:                            # "lz" is the sum of the finder kernels: rank them individually
        per_kernel.pop("lz")
        per_kernel.update(mf_ms)
    for grp in ("mf.far", "mf.shortpass"):       # groups of five kernels each (a second / third W1..W5 pass), not single kernels
        per_kernel.pop(grp, None)
    if "mf.dp" in per_kernel and "mf.parse" in per_kernel:
        per_kernel["mf.parse"] = per_kernel["mf.parse"] / 2.0                  # two launches of gc_mf_parse_kernel
    dom = max(per_kernel, key=lambda k: per_kernel[k])
    algo_bytes = n * (1.0 + 1.0 / ratio)
    achieved = algo_bytes / (per_kernel[dom] * 1e-3) / 1e9
    mf_names = {"mf.count": "gc_mf_count_kernel", "mf.scan": "gc_mf_scan_kernel", "mf.scatter": "gc_mf_scatter_kernel",
                "mf.link": "gc_mf_link_kernel", "mf.verify": "gc_mf_verify_kernel", "mf.parse": "gc_mf_parse_kernel",
                "mf.short": "gc_mf_short_kernel", "mf.deepen": "gc_mf_deepen_kernel", "mf.dp": "gc_mf_dpl2_kernel" if fl2 else ("gc_mf_dp3_kernel + gc_mf_dplz_kernel (phase B per block in one of them)" if (not br and level >= 16) else "gc_mf_dp3_kernel")}
    kname = mf_names[dom] if dom in mf_names else \
        "gc_zstd_lz_kernel" if dom == "lz" else ("gc_lzma2_%s_kernel" if fl2 else ("gc_brotli_%s_kernel" if br else "gc_zstd_%s_kernel")) % dom
    # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/gpu_pmc.sh), same per-launch workload only; the section's
    # commit stamp travels with it (a kernel changed after the pass shows as a stamp that is not the library's commit)
    traffic, traffic_commit, kname = _pmc_lookup(codec, n, kname)
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_measured_at_commit": traffic_commit,
                "algorithmic_bytes_per_launch": int(algo_bytes),
                "kernel_ms": {k: round(v, 4) for k, v in list(kern_ms.items()) + list(mf_ms.items())},
                "pipeline_read_frac": round(n / (kern_ms["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                "pipeline_rw_frac": round(algo_bytes / (kern_ms["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    cpu, ref = (None, None) if args.no_cpu_baseline else cpu_baseline(codec, x_all, level)
    ratio_vs_ref = None
    if ref is not None and not isinstance(ref, tuple):
        ratio_vs_ref = {"note": "whole corpus: our %d-rank stream against the reference's %s at the same level" % (world, "single stream (nbWorkers=0)" if codec == "zstd" else "stream"),
                        "ref_bytes": ref, "ours_bytes": total_csize, "ours_over_ref": round(total_csize / ref, 4), "pass_le_1.02": bool(total_csize <= 1.02 * ref)}
    elif isinstance(ref, tuple) and world == 1:      # reference ran on a prefix: the GPU path once more on exactly that prefix (outside the timed region)
        enc.code_device(d_src.data_ptr(), ref[1], d_dst.data_ptr(), cap)
        ours = enc.finish()
        ratio_vs_ref = {"note": "both encoders on the first sample_bytes of the corpus", "sample_bytes": ref[1], "ref_bytes": ref[0], "ours_bytes": ours,
                        "ours_over_ref": round(ours / ref[0], 4), "pass_le_1.02": bool(ours <= 1.02 * ref[0])}
    frames = ("8 MiB match-finder frames" + (", overlapping (stride %d MiB) in groups of %d MiB" % ((2 if (codec == "zstd" and level >= 20) or (codec == "flzma2" and level >= 8) else 4), grain >> 20) if grain > (8 << 20) else "")) if mf_ms else "block-local match finder"
    res = {
        "metric": ("brotli-q%d (brotli-mt framed)" % level if br else "flzma2-L%d" % level if fl2 else "zstd-L%d" % level) + " compression throughput (input MB/s)",
        "value": round(value, 1), "unit": "MB/s", "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "workload": ("Brotli quality %d, synthetic web-text (%s, %d B), %d MiB brotli-mt chunks" % (level, corpus_name, total, max(level, 1))) if br else
                    ("Fast-LZMA2 level %d, Silesia stand-in (%s, %d B), %s, 4 KiB range-coder chunks grouped into LZMA2 chunks, "
                     "model segments per gc_api.hip flzma2_seg_log" % (level, corpus_name, total, frames)) if fl2 else
                    ("zstd level %d, %s stand-in (%s, %d B), 128 KiB blocks in independent %s" % (level, "enwik9" if total == ENWIK9_BYTES else "enwik", corpus_name, total, frames)),
        "bytes_total": total, "bytes_per_gpu": [b - a for a, b in S.shard_ranges(total, world, grain)], "shard_grain": grain,
        "compressed_bytes": total_csize, "compressed_bytes_per_gpu": sizes, "ratio": round(ratio, 4),
        "decodes_under_reference": decodes, "gpu_decode": gpu_decode, "ratio_vs_ref": ratio_vs_ref, "roofline": roofline, "cpu_baseline": cpu,
    }
    enc.close()
    return res


def shard_sweep(codec, level, corpus_name, total, counts, args, env):
    """--shard-of N1,N2,...: what `--gpus N` would deliver, measured on ONE GPU.  For every N the corpus is range-split exactly as `run_codec` splits it
    (sharding.shard_ranges at the codec grain), EVERY rank's range is timed on this GPU (input resident in HBM, K steps after W warm-ups, one after the
    other), and since the ranks of a real run share nothing (no data-path collective) the job takes as long as its slowest rank: predicted aggregate =
    corpus bytes / max over ranks of ms per step; efficiency = that / (N x the N = 1 rate)."""
    import torch
    corpus_mod, S, pkg, dev = env["corpus"], env["sharding"], env["pkg"], env["dev"]
    fl2, br = codec == "flzma2", codec == "brotli"
    x_all = corpus_mod.corpus(corpus_name, total, seed=20260921)
    d_all = torch.from_numpy(x_all).to(dev)
    enc = (pkg.Flzma2Encoder if fl2 else (pkg.BrotliEncoder if br else pkg.ZstdEncoder))(device=env["local_rank"], level=level)
    cap = enc.compress_bound(total) + 16
    d_dst = torch.empty(cap, dtype=torch.uint8, device=dev)
    rows, base = [], None
    for nr in counts:
        grain = S.codec_grain(codec, level)
        if not br and total < nr * grain:
            grain = S.GRAIN_ZSTD
        ms, comp = [], 0
        for (s, e) in S.shard_ranges(total, nr, grain):
            n = e - s
            if n == 0:
                ms.append(0.0); continue
            d_src = d_all[s:e]
            flags = (enc.NO_END_MARK,) if fl2 else ()
            for _ in range(args.warmup):
                enc.code_device(d_src.data_ptr(), n, d_dst.data_ptr(), cap, *flags); c = enc.finish()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.steps):
                enc.code_device(d_src.data_ptr(), n, d_dst.data_ptr(), cap, *flags); c = enc.finish()
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) / args.steps * 1e3); comp += c
        worst = max(ms)
        rate = total / (worst * 1e-3) / 1e6
        if base is None:
            base = (nr, rate)
        rows.append({"n_gpus": nr, "bytes_per_gpu_max": max(e - s for s, e in S.shard_ranges(total, nr, grain)), "ms_per_rank": [round(v, 3) for v in ms], "ms_slowest_rank": round(worst, 3),
                     "predicted_MBps": round(rate, 1), "speedup_vs_first": round(rate / base[1], 3), "efficiency": round(rate / base[1] / (nr / base[0]), 3), "compressed_bytes": int(comp) + (1 if fl2 else 0)})
    enc.close()
    return {"codec": codec, "level": level, "corpus": corpus_name, "bytes_total": total, "steps": args.steps, "warmup": args.warmup, "rows": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--codec", default="", choices=["", "zstd", "flzma2", "brotli"], help="run ONE codec (default: the metric = zstd-L3 enwik9 + flzma2-L5 Silesia)")
    ap.add_argument("--bytes", type=int, default=0, help="corpus bytes (whole job; default: the codec's BASELINE workload)")
    ap.add_argument("--corpus", default="")
    ap.add_argument("--level", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode-check", action="store_true")
    ap.add_argument("--shard-of", default="", help="N1,N2,...: ONE GPU times every rank's range of an N-way split (both legs of the metric, or --codec) and prints the predicted "
                                                   "aggregate MB/s and efficiency per N instead of the bench line (multi-GPU readiness without an 8-GPU node)")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if rank == 0:
        g.build_hip()                    # (a no-op when the in-tree library is up to date; never two ranks at once)
    if dist is not None:
        dist.barrier()
    pkg = g.load_package()
    from importlib import util as _u
    spec = _u.spec_from_file_location("sevenzip_zstd_amd_corpus", os.path.join(ROOT, "7-zip-zstd_amd", "corpus", "__init__.py"))
    corpus_mod = _u.module_from_spec(spec); spec.loader.exec_module(corpus_mod)
    spec = _u.spec_from_file_location("sevenzip_zstd_amd.sharding", os.path.join(ROOT, "7-zip-zstd_amd", "sharding.py"))
    sharding = _u.module_from_spec(spec); spec.loader.exec_module(sharding)
    env = {"rank": rank, "local_rank": local_rank, "world": world, "dist": dist, "dev": torch.device("cuda", local_rank), "pkg": pkg,
           "corpus": corpus_mod, "sharding": sharding}

    def one(codec):
        lv, cn, nb = DEFAULTS[codec]
        return run_codec(codec, args.level or lv, args.corpus or cn, args.bytes or nb, args, env)

    if args.shard_of:
        if world != 1:
            raise SystemExit("--shard-of predicts N ranks from ONE GPU: run it without torch.distributed.run")
        counts = [int(v) for v in args.shard_of.split(",") if v]
        legs = []
        for codec in ([args.codec] if args.codec else ["zstd", "flzma2"]):
            lv, cn, nb = DEFAULTS[codec]
            legs.append(shard_sweep(codec, args.level or lv, args.corpus or cn, args.bytes or nb, counts, args, env))
        print(json.dumps({"metric": "predicted strong scaling from one GPU: every rank's range of an N-way split timed in turn, job time = slowest rank (no data-path collective)",
                          "unit": "MB/s", "n_gpus": 1, "data": "synthetic", "dtype": "u8", "legs": legs}), flush=True)
        return
    if args.codec:
        main_res, extra = one(args.codec), None
    else:
        main_res = one("zstd")
        extra = one("flzma2")
    if rank == 0:
        line = {
            "metric": main_res["metric"] if args.codec else "compress MB/s + ratio-vs-ref, enwik9 zstd-L3 & Silesia flzma2-L5 (value = the zstd-L3 enwik9 leg; flzma2_l5_silesia = the other leg)",
            "value": main_res["value"], "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": main_res["workload"], "bytes_total": main_res["bytes_total"], "bytes_per_gpu": main_res["bytes_per_gpu"],
                       "shard_grain": main_res["shard_grain"], "parallelism": "range-shard x%d at the codec grain, rank-ordered concatenation, no data-path collective" % world},
            "compressed_bytes": main_res["compressed_bytes"], "compressed_bytes_per_gpu": main_res["compressed_bytes_per_gpu"], "ratio": main_res["ratio"],
            "decodes_under_reference": main_res["decodes_under_reference"], "gpu_decode": main_res.get("gpu_decode"), "ratio_vs_ref": main_res["ratio_vs_ref"],
            "roofline": main_res["roofline"], "cpu_baseline": main_res["cpu_baseline"],
        }
        if extra is not None:
            line["flzma2_l5_silesia"] = extra
        if not args.codec and world == 1 and not args.no_cpu_baseline:
            line["real_data"] = real_data_check(pkg, local_rank)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
