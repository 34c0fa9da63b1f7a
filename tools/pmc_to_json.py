#!/usr/bin/env python3
"""Collect the FETCH_SIZE / WRITE_SIZE sections of tools/gpu_pmc.sh outputs into profiles/pmc_traffic.json.
usage: tools/pmc_to_json.py --commit <sha> [--keep profiles/pmc_traffic.json:zstd_dec] zstd=<pmc.md> flzma2=<pmc.md> brotli=<pmc.md> > profiles/pmc_traffic.json
The passes run `bench.py --steps 3 --warmup 1`, i.e. 4 steps: launches per step of a kernel = its dispatches / 4.
--commit stamps every codec section with the commit the library was built from (bench.py prints it next to `roofline.traffic`, so a
kernel change without a new pass shows as a stale stamp); --keep copies sections of an older file over as they are, stamp included."""
import json, re, sys
WORKLOAD = {"zstd": 1000000000, "flzma2": 211900000, "brotli": 1000000000}
STEPS = 4
args = sys.argv[1:]
commit, keep = None, []
while args and args[0].startswith("--"):
    if args[0] == "--commit": commit = args[1]
    elif args[0] == "--keep": keep.append(args[1])
    args = args[2:]
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/gpu_pmc.sh), mean per dispatch of the default "
                "bench workload of each codec on ONE GPU (zstd: text-zipf 1 GB = the enwik9 stand-in; flzma2: silesia-like 211.9 MB; brotli: web-text 1 GB). "
                "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE counts 64 B per 128 B request on wide coalesced "
                "reads (MI355X_MICROARCH.md, HBM section); inputs <= 256 MB may be served by the Infinity Cache between bench steps, so "
                "FETCH is a lower bound there.  Kernel names with the suffix _p8 are the fast geometry of the windowed finder (gc_mf.h). "
                "launches_per_step: dispatches of the kernel in one pass over the workload; _commit: the commit the measured library was built from.",
       "_workload_bytes": dict(WORKLOAD)}
for k in keep:
    path, sec = k.split(":")
    old = json.load(open(path))
    out[sec] = old[sec]
    if sec in old.get("_workload_bytes", {}): out["_workload_bytes"][sec] = old["_workload_bytes"][sec]
for arg in args:
    codec, path = arg.split("=", 1)
    sec, d = None, {}
    for line in open(path):
        m = re.match(r"## (\S+)", line)
        if m:
            sec = m.group(1); continue
        if sec in ("FETCH_SIZE", "WRITE_SIZE") and line.startswith("| gc_"):
            f = [x.strip() for x in line.strip().strip("|").split("|")]
            e = d.setdefault(f[0], {})
            e["fetch_kb" if sec == "FETCH_SIZE" else "write_kb"] = float(f[3])
            e["launches_per_step"] = max(1, round(int(f[2]) / STEPS))
    for k, v in d.items():
        v["hbm_bytes_per_launch"] = int((2 * v.get("fetch_kb", 0.0) + v.get("write_kb", 0.0)) * 1024)
    tot = int(sum(v["hbm_bytes_per_launch"] * v["launches_per_step"] for k, v in d.items()))
    d["_total_hbm_bytes_per_step"] = tot
    d["_hbm_bytes_per_input_byte"] = round(tot / WORKLOAD[codec], 2)
    d["_commit"] = commit
    out[codec] = d
print(json.dumps(out, indent=1))
