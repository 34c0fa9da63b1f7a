#!/usr/bin/env python3
"""Collect the FETCH_SIZE / WRITE_SIZE sections of tools/gpu_pmc.sh outputs into profiles/pmc_traffic.json.
usage: tools/pmc_to_json.py zstd=<pmc.md> flzma2=<pmc.md> brotli=<pmc.md> > profiles/pmc_traffic.json"""
import json, re, sys
WORKLOAD = {"zstd": 1000000000, "flzma2": 211900000, "brotli": 1000000000}
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/gpu_pmc.sh), mean per dispatch of the default "
                "bench workload of each codec on ONE GPU (zstd: text-zipf 1 GB = the enwik9 stand-in; flzma2: silesia-like 211.9 MB; brotli: web-text 1 GB). "
                "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 64 B per 128 B request on wide coalesced "
                "reads (MI355X_MICROARCH.md, HBM section); inputs <= 256 MB may be served by the Infinity Cache between bench steps, so "
                "FETCH is a lower bound there.  Kernel names with the suffix _p8 are the fast geometry of the windowed finder (gc_mf.h).",
       "_workload_bytes": WORKLOAD}
# launches per bench step of kernels that run more than once (the averages above are per launch)
LAUNCHES = {"zstd": {}, "flzma2": {"gc_mf_link_kernel": 3, "gc_mf_scan_kernel": 3, "gc_mf_parse_kernel": 2, "gc_mf_dp2_kernel": 2}, "brotli": {"gc_mf_link_kernel": 2, "gc_mf_scan_kernel": 2}}
for arg in sys.argv[1:]:
    codec, path = arg.split("=", 1)
    sec, d = None, {}
    for line in open(path):
        m = re.match(r"## (\S+)", line)
        if m:
            sec = m.group(1); continue
        if sec in ("FETCH_SIZE", "WRITE_SIZE") and line.startswith("| gc_"):
            f = [x.strip() for x in line.strip().strip("|").split("|")]
            d.setdefault(f[0], {})["fetch_kb" if sec == "FETCH_SIZE" else "write_kb"] = float(f[3])
    for k, v in d.items():
        v["hbm_bytes_per_launch"] = int((2 * v.get("fetch_kb", 0.0) + v.get("write_kb", 0.0)) * 1024)
    d["_total_hbm_bytes_per_step"] = int(sum(v["hbm_bytes_per_launch"] * LAUNCHES[codec].get(k, 1) for k, v in d.items() if k.startswith("gc_")))
    d["_hbm_bytes_per_input_byte"] = round(d["_total_hbm_bytes_per_step"] / WORKLOAD[codec], 2)
    out[codec] = d
print(json.dumps(out, indent=1))
