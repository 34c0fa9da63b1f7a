// test-only driver: run K1 under the SIMT emulator
#include "hipemu.h"
#include "../../7-zip-zstd_amd/csrc/gc_common.h"
extern "C" void gc_zstd_lz_kernel(const uint8_t* src, uint64_t srcSize, GcSeqRaw* seqRaw, uint8_t* lit, GcBlockMeta* meta);
extern "C" void emu_lz(const uint8_t* src, uint64_t n, GcSeqRaw* seqRaw, uint8_t* lit, GcBlockMeta* meta)
{
    uint32_t nb = gc_num_blocks(n);
    HIPEMU_LAUNCH(gc_zstd_lz_kernel, dim3(nb), dim3(1024), src, n, seqRaw, lit, meta);
}
