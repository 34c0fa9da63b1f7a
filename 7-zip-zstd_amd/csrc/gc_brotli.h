// gc_brotli.h -- shared definitions of the BROTLI GPU path.
#pragma once
#include <stdint.h>

// stream header of every brotli-mt chunk: WBITS = 24 ('1' then n = 7: WBITS = 17 + n, RFC 7932 section 9.1), the window the
// reference sets (lgwin 24, C/zstdmt/brotli-mt_compress.c:284-287).  Copies reach back at most to the start of their chunk.
#define GC_BR_WBITS_CODE 0xFu
#define GC_BR_STAGE_STRIDE (GC_ZSTD_BLOCK_MAX + 4096u)   // bytes of zero-initialised bit staging per 128 KiB block
struct GcBrotliBlockInfo { uint32_t size; uint32_t stored; uint32_t hdrBits; uint32_t lastInChunk; };
struct GcBrotliPlan { uint64_t off; uint32_t chunkSize; uint32_t pad; };   // off: where the block's bytes go; chunkSize: brotli bytes of the whole chunk (first block only)
