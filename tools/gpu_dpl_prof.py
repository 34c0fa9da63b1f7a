#!/usr/bin/env python3
"""In-kernel cycle profile of W7L (gc_lz_dpl.hip built with -DDPL_PROF into tools/_variants/libgpucodec_dplprof.so): shader-clock sums of lane 0 of every
wave of the full-window kernel, per section of a node step.  usage: python tools/gpu_dpl_prof.py [corpus] [bytes]"""
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import __graft_entry__ as g
import oracle as O
out = os.path.join(ROOT, 'tools', '_variants'); os.makedirs(out, exist_ok=True)
lib = os.path.join(out, 'libgpucodec_dplprof.so')
if not os.path.exists(lib):
    objs = g.compile_hip_objects(os.path.join(g.CSRC, '_obj'))
    pobjs = g.compile_hip_objects(os.path.join(out, '_obj'), only={'gc_lz_dpl.hip': ['-DDPL_PROF']})
    g.link_hip([p or o for p, o in zip(pobjs, objs)], lib)
if len(sys.argv) > 1 and sys.argv[1] == 'build': sys.exit(0)
pkg = g.load_package()
kind = sys.argv[1] if len(sys.argv) > 1 else 'silesia-like'; n = int(sys.argv[2]) if len(sys.argv) > 2 else 211_900_000
x = O.corpus(kind, n)
e = pkg.Flzma2Encoder(level=5, device=0, lib_path=lib); c = e.code(x)
L = C.CDLL(lib); buf = (C.c_ulonglong * 32)()
L.gc_dpl_prof_read(buf, 1); c = e.code(x); L.gc_dpl_prof_read(buf, 0); e.close()
names = ['finalize', 'literal+cont', 'main+short', 'tracked repeats', 'between groups (track, shadow)', 'ring counters', '-', 'loop head + group load', 'walk back', 'waiting at the barrier']
for role in (0, 1):
    b = buf[16 * role:16 * role + 10]; tot = sum(b)
    print('wave %d of the pair: %.3e cycles' % (role, tot))
    for k, nm in enumerate(names):
        if nm != '-' and tot: print('  %-32s %6.2f %%  %.3e cycles' % (nm, 100.0 * b[k] / tot, b[k]))
print('compressed', len(c))
