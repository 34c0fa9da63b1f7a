#!/usr/bin/env python3
"""Copy a round's final GPU session (gpurun_out/<tag>/ of tools/gpu_session.sh: tests, bench, prof, pmc x3, bench x3, shard, sizes, real rate) into profiles/rNN_final_*.md and
profiles/pmc_traffic.json.   usage: python tools/collect_final.py <tag> <round> <commit>"""
import json, os, re, subprocess, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd, commit = sys.argv[1], sys.argv[2], sys.argv[3]
G = os.path.join(ROOT, "gpurun_out", tag); P = os.path.join(ROOT, "profiles")
def rd(name): return open(os.path.join(G, name)).read() if os.path.exists(os.path.join(G, name)) else ""
def last_json_line(name):
    t = rd(name).strip().splitlines()
    return t[-1] if t else ""
head = "# round %s, final run (commit %s; one MI355X, `tools/gpu_session.sh %s ...`)" % (rnd, commit, tag)
# tests
t = [f for f in sorted(os.listdir(G)) if f.startswith("tests")]
open(os.path.join(P, "r%s_final_gpu_tests.md" % rnd), "w").write(head + ": `python -m pytest tests -m gpu -q`\n```\n" + "".join(l for l in rd(t[0]).splitlines(True) if not l.startswith("/opt/amdgpu")) + "```\n" + ("smoke: " + rd("smoke.log").strip().splitlines()[-1] + "\n" if rd("smoke.log").strip() else ""))
# bench lines: the default one, then the three per-codec lines in the order they ran
b = sorted([f for f in os.listdir(G) if re.match(r"bench\d+\.json", f)], key=lambda f: int(re.findall(r"\d+", f)[0]))
names = ["the default `python bench.py` line (the metric: zstd-L3 on the 1 GB enwik9 stand-in, Fast-LZMA2-L5 on the Silesia stand-in in `flzma2_l5_silesia`, `real_data`, both CPU baselines)",
         "C2: zstd level 3, 100 MB (`--codec zstd --bytes 100000000 --no-cpu-baseline`)", "C4 share: zstd level 19, 125 MB (`--codec zstd --level 19 --bytes 125000000 --no-cpu-baseline`)",
         "C5 share: brotli quality 6, 1 GB of web-text (`--codec brotli --no-cpu-baseline`)"]
out = [head + ": bench lines", ""]
for f, nm in zip(b, names):
    out += ["## " + nm, "```", last_json_line(f), "```", ""]
open(os.path.join(P, "r%s_final_bench_line.md" % rnd), "w").write("\n".join(out))
# kernel stats
ks = [f for f in sorted(os.listdir(G)) if f.startswith("kernel_stats")]
if ks:
    open(os.path.join(P, "r%s_final_metric_kernel_stats.md" % rnd), "w").write(head + ": rocprofv3 --kernel-trace --stats of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-decode-check` (both legs of the metric; 4 calls = 1 warm-up + 3 steps)\n\n" + rd(ks[0]))
# pmc
pm = sorted([f for f in os.listdir(G) if re.match(r"pmc\d+\.md", f)], key=lambda f: int(re.findall(r"\d+", f)[0]))
codecs = ["zstd", "flzma2", "brotli"][:len(pm)]
args = []
for f, c in zip(pm, codecs):
    dst = os.path.join(P, "r%s_pmc_%s.md" % (rnd, {"zstd": "zstd", "flzma2": "fl2", "brotli": "brotli"}[c]))
    open(dst, "w").write(head + ": HBM traffic counters, `bench.py --codec %s --steps 3 --warmup 1` (separate rocprofv3 --pmc passes, KiB per dispatch; hbm bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB, see profiles/pmc_traffic.json)\n" % c + rd(f))
    args.append("%s=%s" % (c, dst))
if args:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_to_json.py"), "--commit", commit, "--keep", os.path.join(P, "pmc_traffic.json") + ":zstd_dec"] + args, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    open(os.path.join(P, "pmc_traffic.json"), "w").write(r.stdout)
    d = json.loads(r.stdout); print({c: d[c]["_hbm_bytes_per_input_byte"] for c in codecs})
print("ok")
