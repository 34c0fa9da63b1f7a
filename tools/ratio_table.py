#!/usr/bin/env python3
"""profiles/rNN_ratio.md from a tools/gpu_sizes.py log (JSON lines) and tools/ref_sizes_cache.json.   usage: python tools/ratio_table.py <log> <round> <tag> <commit> > profiles/rNN_ratio.md"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log, rnd, tag, commit = sys.argv[1:5]
cache = json.load(open(os.path.join(ROOT, "tools", "ref_sizes_cache.json")))
print("# round %s: compressed size of this engine against the reference encoder at the same level on the same bytes (one MI355X, run %s = commit %s; `tools/gpu_sizes.py`," % (rnd, tag, commit))
print("# reference sizes from `tools/ref_sizes_cache.json` = `oracle/_ref` on the build container; streams of these codec / level / corpus combinations are decoded by the reference decoders in `pytest -m gpu`)\n")
print("| codec | level | corpus | bytes | ours | reference | ours / reference |\n|---|---|---|---|---|---|---|")
for l in open(log):
    l = l.strip()
    if not l.startswith("{"): continue
    d = json.loads(l)
    ref = d.get("ref") or cache.get("%s:%d:%s:%d" % (d["codec"], d["level"], d["corpus"], d["bytes"]))
    r = ("%.4f" % (d["ours"] / ref)) if ref else ""
    print("| %s | %d | %s | %d | %d | %s | %s |" % (d["codec"], d["level"], d["corpus"], d["bytes"], d["ours"], ref or "", ("**" + r + "**") if r and float(r) <= 1.02 else r))
