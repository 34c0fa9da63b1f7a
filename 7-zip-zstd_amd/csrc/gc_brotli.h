// gc_brotli.h -- shared definitions of the BROTLI GPU path.
#pragma once
#include <stdint.h>

#define GC_BR_STAGE_STRIDE (GC_ZSTD_BLOCK_MAX + 4096u)   // bytes of zero-initialised bit staging per 128 KiB block
struct GcBrotliBlockInfo { uint32_t size; uint32_t stored; uint32_t hdrBits; uint32_t lastInChunk; };
struct GcBrotliPlan { uint64_t off; uint32_t chunkSize; uint32_t pad; };   // off: where the block's bytes go; chunkSize: brotli bytes of the whole chunk (first block only)
