#!/bin/bash
# Builds the library variants that tools/gpu_seqv_variants.py runs on the GPU box (profiles/r02_dpp_combine.md): one hypothesis per library.
# (ZV_KEEP_DPP_FOLD=1 takes the empty asm behind the quad-permute moves out again, i.e. the code as first written.)
cd "$(dirname "$0")/.." && mkdir -p tools/_variants
SRCS=$(python3 -c "
import __graft_entry__ as g, os
print(' '.join(os.path.join(g.CSRC, s) for s in g.HIP_SOURCES))")
build() { name=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-value -Iinclude -I7-zip-zstd_amd/csrc "$@" $SRCS -o tools/_variants/lib_$name.so 2>&1 | grep -i "error" | head -3; }
build shipped -O3 &
build d0 -O3 -DZV_KEEP_DPP_FOLD=1 -DZV_DBG=0 &
build d1 -O3 -DZV_KEEP_DPP_FOLD=1 -DZV_DBG=1 &
build d2 -O3 -DZV_KEEP_DPP_FOLD=1 -DZV_DBG=2 &
build d0_nodppc -O3 -DZV_KEEP_DPP_FOLD=1 -DZV_DBG=0 -mllvm -amdgpu-dpp-combine=false &
build d0_sync -O3 -DZV_KEEP_DPP_FOLD=1 -DZV_DBG=0 -DZV_STRONG_SYNC &
build d0_zero -O3 -DZV_KEEP_DPP_FOLD=1 -DZV_DBG=0 -DZV_ZERO_LDS &
wait; ls -la tools/_variants
