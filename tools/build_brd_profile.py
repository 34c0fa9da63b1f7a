#!/usr/bin/env python3
"""Check build of the BROTLI decoder with -DBRD_PROFILE (chunk 0 prints the clock ticks per section): csrc/libgpucodec_brdprof.so.  usage: python tools/build_brd_profile.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import __graft_entry__ as g
objs = g.compile_hip_objects(os.path.join(g.CSRC, "_obj"))
pobjs = g.compile_hip_objects(os.path.join(g.CSRC, "_obj"), only={"gc_brotli_dec.hip": ["-DBRD_PROFILE"]})
print(g.link_hip([p or o for p, o in zip(pobjs, objs)], os.path.join(g.CSRC, "libgpucodec_brdprof.so")))
