"""GPU diagnostic: FLZMA2 level 5 sizes of one input through fresh / reused contexts of the shipped and the hooks library (a result that depends on what the
workspace held before the call shows as sizes that differ between these).  usage: python tools/gpu_diag_fl2.py [bytes]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32 << 20
x = O.corpus('silesia-like', n)
libs = {'shipped': g.LIB, 'hooks': g.LIB_HOOKS}
v = os.path.join(ROOT, 'tools', '_variants', 'libgpucodec_x.so')
if os.path.exists(v): libs['variant'] = v
def one(lib, level=5, calls=1, codec='flzma2'):
    e = (pkg.Flzma2Encoder if codec == 'flzma2' else pkg.ZstdEncoder)(level=level, device=0, lib_path=lib)
    out = [len(e.code(x)) for _ in range(calls)]
    e.close()
    return out
for name, lib in libs.items():
    print(name, 'fresh ctx x3:', one(lib), one(lib), one(lib), ' one ctx, 3 calls:', one(lib, calls=3), flush=True)
for name, lib in libs.items():
    print(name, 'after a zstd-19 context:', one(lib, 19, 1, 'zstd'), one(lib), flush=True)
os.environ['GC_PRICE_PARSE'] = '0'; print('hooks greedy', one(libs['hooks'])); del os.environ['GC_PRICE_PARSE']
print('hooks priced after greedy', one(libs['hooks']), one(libs['hooks']), flush=True)
os.environ['GC_DPL'] = '0'; print('hooks W7', one(libs['hooks'])); del os.environ['GC_DPL']
