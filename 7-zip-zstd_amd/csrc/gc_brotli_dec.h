// gc_brotli_dec.h -- the BROTLI decoder's workspace, owned by a gc_ctx (gc_api.hip) and used by gc_brotli_dec.hip
#pragma once
#include <stdint.h>
#include <stddef.h>
struct GcBrDecWork {
    uint8_t* stage; size_t stageCap;      // every chunk decodes into a slot of its hint size
    uint8_t* pages; uint32_t nPages;      // HBM behind the LDS arenas (meta-blocks with hundreds of prefix codes)
    uint8_t* meta; size_t metaCap;        // chunk descriptors, results, offsets, totals, the page cursor
    uint8_t* dict; uint64_t dictStamp;    // the static dictionary on this device, if the process holds one
    uint32_t instance;                    // test hook (GC_BRD_INSTANCE): 1-4 = the kernel instance (LDS arena / ring size) whatever the number of chunks; 0 = by the number of chunks
    uint32_t ldsCap;                      // test hook (GC_BRD_LDS): a smaller LDS arena, so that small inputs take the HBM pages; 0 = the kernel's own
    void* ev0; void* ev1; float ms;       // HIP events around the kernels of the last call
};
