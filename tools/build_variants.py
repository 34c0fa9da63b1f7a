#!/usr/bin/env python3
"""Library variants for A/B runs on the GPU box (tools/_variants/lib_<name>.so, git-ignored, they travel with gpurun).
usage: tools/build_variants.py name=file.hip:-DX=1,-DY=2[+file2.hip:-DZ] ...     (a bare `name` = the shipped flags)
Only the named sources are recompiled with the extra flags; the other objects come from the normal build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

os.makedirs(os.path.join(ROOT, "tools", "_variants", "obj"), exist_ok=True)
g.build_hip()
for spec in sys.argv[1:]:
    name, _, rest = spec.partition("=")
    only = {}
    for part in filter(None, rest.split("+")):
        f, _, fl = part.partition(":")
        only[f] = [x for x in fl.split(",") if x]
        if f == "gc_lz_window.hip":                   # the fast geometry is a second compile of the same source
            only.setdefault("gc_lz_window_p8.hip", only[f])
    only.setdefault("gc_api.hip", []).append("-DGC_TEST_HOOKS")       # variants are test builds: the GC_* hooks work in them
    base = g.compile_hip_objects(os.path.join(g.CSRC, "_obj"))
    objs = g.compile_hip_objects(os.path.join(ROOT, "tools", "_variants", "obj"), only=only) if only else base
    # objects of sources without flags of their own: the normal build's
    final = [o or b for o, b in zip(objs, base)]
    out = os.path.join(ROOT, "tools", "_variants", "lib_%s.so" % name)
    g.link_hip(final, out)
    print(out)
