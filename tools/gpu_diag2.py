"""GPU diagnostic: run tests/test_gpu_parity.py inside this process, then code one FLZMA2 input through fresh contexts and list which LZMA2 chunks came out stored
(a result that depends on what ran before in the process).  usage: python tools/gpu_diag2.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, pytest
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
x = O.corpus('silesia-like', 32 << 20)
def chunks(c):
    p = 0; raw = []; nl = 0; pos = 0
    while p < len(c):
        ctl = int(c[p])
        if ctl == 0: break
        if ctl < 0x80: u = (int(c[p+1]) << 8 | int(c[p+2])) + 1; raw.append(pos >> 12); p += 3 + u
        else: u = ((ctl & 31) << 16 | int(c[p+1]) << 8 | int(c[p+2])) + 1; cs = (int(c[p+3]) << 8 | int(c[p+4])) + 1; p += 5 + cs + (1 if ctl >= 0xC0 else 0); nl += 1
        pos += u
    return nl, len(raw), raw[:12], raw[-4:]
def one(lib, **env):
    for k, v in env.items(): os.environ[k] = v
    e = pkg.Flzma2Encoder(level=5, device=0, lib_path=lib); c = e.code(x); e.close()
    for k in env: del os.environ[k]
    return len(c), chunks(c)
print('before:', one(g.LIB_HOOKS), flush=True)
what = sys.argv[1:] or ['tests/test_gpu_parity.py']
rc = pytest.main(what + ['-m', 'gpu', '-q', '-x', '-p', 'no:cacheprovider'])
print('pytest rc', rc, flush=True)
print('after, hooks :', one(g.LIB_HOOKS), flush=True)
print('after, hooks :', one(g.LIB_HOOKS), flush=True)
print('after, shipped:', one(g.LIB), flush=True)
print('after, hooks W7:', one(g.LIB_HOOKS, GC_DPL='0'), flush=True)
print('after, hooks greedy:', one(g.LIB_HOOKS, GC_PRICE_PARSE='0'), flush=True)
print('after, hooks no rep4:', one(g.LIB_HOOKS, GC_L2_REP4='0'), flush=True)
print('after, hooks one phase:', one(g.LIB_HOOKS, GC_DP_PHASES='1'), flush=True)
import torch
torch.cuda.empty_cache()
print('after empty_cache, hooks :', one(g.LIB_HOOKS), flush=True)
