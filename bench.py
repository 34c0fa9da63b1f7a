#!/usr/bin/env python3
"""bench.py -- compression throughput of the MI355X zstd path on BASELINE.json config 2.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the whole hot path (match finder W1..W6 -> K2 huf || K3 seq -> K4 plan -> K5 emit) over one
100 000 000-byte buffer per GPU that is already resident in HBM (enwik8 is not available offline; the stand-in
is the deterministic `text-zipf` corpus, labelled synthetic).  At level 3 the 128 KiB zstd blocks are grouped into
independent 8 MiB frames (windowed match finder); levels 1-2 use one frame per block (block-local finder).
With N>1 every rank compresses its own 100 MB shard (weak scaling, no data-path collective: the host
range-splits the input and concatenates frames; RCCL is only used for the timing barrier / max-reduction).

One JSON line is printed by rank 0; see DESIGN.md "Measurement" for the definition of every field.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline_flzma2(x, level, budget_s=25.0):
    """Reference Fast-LZMA2 (oracle/_ref/libflzma2_ref.so = C/fast-lzma2 compiled from /root/reference) on the host cores,
    on a bounded sample (the first 32 MiB: ~10-20 s of CPU work over both legs)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O          # cpu_baseline leg only
    if O.ref("flzma2") is None:
        return None, None
    cores = os.cpu_count() or 1
    sample = x[: min(x.size, 32 * 1024 * 1024)]
    t0 = time.perf_counter(); c1, _ = O.ref_fl2_compress(sample, level, threads=1); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); cm, _ = O.ref_fl2_compress(sample, level, threads=cores); tm = time.perf_counter() - t0
    res = {"value": round(sample.size / tm / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "reference",
           "sample": "FL2_compressCCtx level %d on the first %d bytes of the same buffer, one run; %d threads; single thread: %.1f MB/s"
                     % (level, sample.size, cores, sample.size / t1 / 1e6)}
    return res, (len(c1), sample.size)


def cpu_baseline_brotli(x, level, budget_s=25.0):
    """Reference brotli + brotli-mt framing (oracle/_ref/libbrotli_ref.so) on the host cores, bounded sample (first 64 MiB)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O          # cpu_baseline leg only
    if O.ref("brotli") is None:
        return None, None
    cores = min(os.cpu_count() or 1, 128)          # BROTLIMT_THREAD_MAX
    sample = x[: min(x.size, 64 * 1024 * 1024)]
    one = sample[: 8 * 1024 * 1024]
    t0 = time.perf_counter(); c1 = O.ref_brotlimt_compress(one, level, 1); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); cm = O.ref_brotlimt_compress(sample, level, cores); tm = time.perf_counter() - t0
    res = {"value": round(sample.size / tm / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "reference",
           "sample": "BROTLIMT_compressCCtx quality %d on the first %d bytes of the same buffer, one run, %d threads; single thread (8 MiB): %.1f MB/s"
                     % (level, sample.size, cores, one.size / t1 / 1e6)}
    return res, (len(cm), sample.size)


def cpu_baseline(x, level, budget_s=25.0):
    """Reference zstd (oracle/_ref/libzstd_ref.so = C/zstd compiled from /root/reference) timed on the host
    cores of this box, on a bounded sample of the same workload.  Reported, not the optimisation target."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O          # cpu_baseline leg only
    if O.ref("zstd") is None:
        return None, None
    cores = os.cpu_count() or 1
    sample = x[: min(x.size, 100_000_000)]
    out = {}
    ref_size = None
    for label, workers in (("1 thread (nbWorkers=0)", 0), ("%d threads (ZSTDMT nbWorkers=%d)" % (cores, cores), cores)):
        best = None
        t_spent = 0.0
        for _ in range(3):
            t0 = time.perf_counter()
            c = O.ref_zstd_compress(sample, level, workers=workers)
            dt = time.perf_counter() - t0
            t_spent += dt
            best = dt if best is None else min(best, dt)
            if workers == 0:
                ref_size = len(c)
            if t_spent > budget_s / 2:
                break
        out[label] = sample.size / best / 1e6
    mt_label = list(out.keys())[1]
    res = {"value": round(out[mt_label], 1), "unit": "MB/s", "cores": cores, "kind": "reference",
           "sample": "ZSTD_compress2 level %d on the first %d bytes of the same buffer, best of <=3 runs; %s; single thread: %.1f MB/s"
                     % (level, sample.size, mt_label, list(out.values())[0])}
    return res, ref_size


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--codec", default="zstd", choices=["zstd", "flzma2", "brotli"])
    ap.add_argument("--bytes", type=int, default=0, help="input bytes per GPU (default: 100 000 000 = enwik8 size for zstd, 211 900 000 = Silesia for flzma2)")
    ap.add_argument("--corpus", default="")
    ap.add_argument("--level", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
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

    fl2 = args.codec == "flzma2"
    br = args.codec == "brotli"
    args.bytes = args.bytes or (211_900_000 if fl2 else (1_000_000_000 if br else 100_000_000))
    args.corpus = args.corpus or ("silesia-like" if fl2 else ("web-text" if br else "text-zipf"))
    args.level = args.level or (5 if fl2 else (6 if br else 3))
    n = args.bytes
    x = corpus_mod.corpus(args.corpus, n, seed=20260921 + rank)      # each rank owns a different shard
    enc = (pkg.Flzma2Encoder if fl2 else (pkg.BrotliEncoder if br else pkg.ZstdEncoder))(device=local_rank, level=args.level)
    dev = torch.device("cuda", local_rank)
    d_src = torch.from_numpy(x).to(dev)
    cap = enc.compress_bound(n)
    d_dst = torch.empty(cap, dtype=torch.uint8, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    csize = 0
    for _ in range(args.warmup):
        enc.code_device(d_src.data_ptr(), n, d_dst.data_ptr(), cap)
        csize = enc.finish()

    kern_ms = {k: 0.0 for k in enc.KERNELS}
    mf_ms = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        enc.code_device(d_src.data_ptr(), n, d_dst.data_ptr(), cap)
        csize = enc.finish()
        t = enc.last_timing_ms()         # hipEvent pairs recorded on the library's own stream around each kernel
        for k in kern_ms:
            kern_ms[k] += t[k]
        mf = enc.mf_timing_ms()          # stage durations of the windowed match finder (None: block-local finder ran)
        if mf:
            for k, v in mf.items():
                mf_ms[k] = mf_ms.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        sz = torch.tensor([csize], dtype=torch.int64, device=dev)
        dist.all_reduce(sz, op=dist.ReduceOp.SUM)
        total_csize = int(sz.item())
    else:
        total_csize = csize
    for k in kern_ms:
        kern_ms[k] /= max(args.steps, 1)
    for k in mf_ms:
        mf_ms[k] /= max(args.steps, 1)

    if rank == 0:
        total_in = n * world
        value = total_in * args.steps / elapsed / 1e6
        ratio = n / csize
        # dominant kernel = the longest of the five; algorithmic bytes per launch = N_in * (1 + 1/ratio)  
        per_kernel = {k: v for k, v in kern_ms.items() if k != "total"}
        if mf_ms:                        # "lz" is the sum of the five finder kernels: rank them individually
            per_kernel.pop("lz")
            per_kernel.update(mf_ms)
        # single-kernel entries only: "mf.far" / "mf.shortpass" are groups of five kernels each (a second / third W1..W5 pass),
        # and with the price-based parse "mf.parse" is two launches of gc_mf_parse_kernel (rank it by its per-launch average)
        for grp in ("mf.far", "mf.shortpass"):
            per_kernel.pop(grp, None)
        if "mf.dp" in per_kernel and "mf.parse" in per_kernel:
            per_kernel["mf.parse"] = per_kernel["mf.parse"] / 2.0
        dom = max(per_kernel, key=lambda k: per_kernel[k])
        algo_bytes = n * (1.0 + 1.0 / ratio)
        achieved = algo_bytes / (per_kernel[dom] * 1e-3) / 1e9
        mf_names = {"mf.count": "gc_mf_count_kernel", "mf.scan": "gc_mf_scan_kernel", "mf.scatter": "gc_mf_scatter_kernel",
                    "mf.link": "gc_mf_link_kernel", "mf.verify": "gc_mf_verify_kernel", "mf.parse": "gc_mf_parse_kernel",
                    "mf.short": "gc_mf_short_kernel", "mf.deepen": "gc_mf_deepen_kernel", "mf.dp": "gc_mf_dp2_kernel" if fl2 else "gc_mf_dp3_kernel"}
        kname = mf_names[dom] if dom in mf_names else \
            "gc_zstd_lz_kernel" if dom == "lz" else ("gc_lzma2_%s_kernel" if fl2 else ("gc_brotli_%s_kernel" if br else "gc_zstd_%s_kernel")) % dom
        traffic = None
        try:        # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/gpu_pmc.sh), same workload only
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pm["_workload_bytes"].get(args.codec) == n and kname in pm.get(args.codec, {}):
                traffic = pm[args.codec][kname]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(algo_bytes),
                    "kernel_ms": {k: round(v, 4) for k, v in list(kern_ms.items()) + list(mf_ms.items())},
                    "pipeline_read_frac": round(n / (kern_ms["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                    "pipeline_rw_frac": round(algo_bytes / (kern_ms["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        if br:
            cpu, ref_info = (None, None) if args.no_cpu_baseline else cpu_baseline_brotli(x, args.level)
            ref_size = None
        elif fl2:
            cpu, ref_info = (None, None) if args.no_cpu_baseline else cpu_baseline_flzma2(x, args.level)
            ref_size = None
        else:
            cpu, ref_size = (None, None) if args.no_cpu_baseline else cpu_baseline(x, args.level)
            ref_info = None
        ours_on_sample = None
        if ref_info:                     # same bytes through the GPU path (outside the timed region): size against the reference's, like for like
            enc.code_device(d_src.data_ptr(), ref_info[1], d_dst.data_ptr(), cap)
            ours_on_sample = enc.finish()
        line = {
            "metric": ("brotli-q%d (brotli-mt framed) compression throughput (input MB/s)" % args.level) if br else
                      ("flzma2-L%d compression throughput (input MB/s)" % args.level) if fl2 else
                      "zstd-L%d compression throughput (input MB/s)" % args.level,
            "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": ("Brotli quality %d, synthetic web-text (%s, %d B per GPU), %d MiB brotli-mt chunks" % (args.level, args.corpus, n, args.level)) if br else
                                   ("Fast-LZMA2 level %d, Silesia stand-in (%s, %d B per GPU), 8 MiB match-finder frames, 4 KiB range-coder chunks grouped into LZMA2 chunks of <= 32 KiB, model reset every %d KiB" % (args.level, args.corpus, n, {1: 16, 2: 16, 3: 16, 4: 32, 5: 32, 6: 64, 7: 64}.get(args.level, 128))) if fl2 else
                                   "zstd level %d, enwik8 stand-in (%s, %d B per GPU), 128 KiB blocks in %s" % (
                                       args.level, args.corpus, n, "independent 8 MiB frames (windowed match finder)" if mf_ms else "one frame per block (block-local match finder)"),
                       "bytes_per_gpu": n, "blocks_per_gpu": (n + 131071) // 131072, "parallelism": "range-shard x%d, no collective" % world},
            "compressed_bytes": total_csize, "ratio": round(ratio, 4),
            "ratio_vs_ref": ({"note": "both encoders on the reference's CPU sample (the first sample_bytes of the buffer)", "sample_bytes": ref_info[1],
                              "ref_bytes": ref_info[0], "ours_bytes": ours_on_sample, "ours_over_ref": round(ours_on_sample / ref_info[0], 4),
                              "ours_ratio_whole_input": round(ratio, 4)} if ref_info else None) if (fl2 or br) else
                            (None if (not ref_size or n > 100_000_000) else
                             {"ours_over_ref_single_stream_L%d" % args.level: round(csize / ref_size, 4), "ref_bytes": ref_size}),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    enc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
