// gc_lzma2.h -- shared definitions of the FLZMA2 (LZMA2) GPU path.
#pragma once
#include <stdint.h>

#define GC_LZMA_LC 3u                       // lc=3 lp=0 pb=2: the reference's defaults (fl2_compress.c:116-138)
#define GC_LZMA_LP 0u
#define GC_LZMA_PB 2u
#define GC_LZMA_PROPS ((GC_LZMA_PB * 5u + GC_LZMA_LP) * 9u + GC_LZMA_LC)     // 0x5D
#define GC_LZMA_CHUNK_LOG_MAX 16u           // LZMA2: packed chunk <= 64 KiB; raw chunk <= 64 KiB (Lzma2Dec.c:97)
#define GC_LZMA_CHUNK_LOG_MIN 12u

// probability model layout (indices into one uint16 array per chunk; the layout is private to the encoder)
#define LZP_ISMATCH    0u                   // [12 states][4 posStates]
#define LZP_ISREP      48u                  // [12]
#define LZP_ISREPG0    60u
#define LZP_ISREPG1    72u
#define LZP_ISREPG2    84u
#define LZP_ISREP0LONG 96u                  // [12][4]
#define LZP_LEN        144u                 // length coder: choice, choice2, low[4][8], mid[4][8], high[256]
#define LZL_CHOICE     0u
#define LZL_CHOICE2    1u
#define LZL_LOW        2u
#define LZL_MID        34u
#define LZL_HIGH       66u
#define LZL_SIZE       322u
#define LZP_REPLEN     (LZP_LEN + LZL_SIZE)             // 466
#define LZP_POSSLOT    (LZP_REPLEN + LZL_SIZE)          // 788: [4 length states][64]
#define LZP_SPECPOS    (LZP_POSSLOT + 256u)             // 1044: slots 4..13, reverse trees of <= 5 bits
#define LZP_ALIGN      (LZP_SPECPOS + 320u)             // 1364: 4-bit reverse tree
#define LZP_LITERAL    (LZP_ALIGN + 16u)                // 1380: 0x300 << lc
#define LZP_TOTAL      (LZP_LITERAL + (0x300u << GC_LZMA_LC))

struct GcLzmaChunkInfo { uint32_t usize; uint32_t csize; uint32_t nWords; uint32_t pad; };   // csize 0xFFFFFFFF: store raw; usize 0: chunk does not exist
// words of (p, bit) stream reserved per chunk: a literal is the densest symbol (9 coded bits per byte); keeps 16-byte alignment
#define GC_LZMA_STREAM_WORDS(chunkLog) ((9u << (chunkLog)) + 64u)
struct GcLzmaPlan { uint64_t off; uint32_t size; uint32_t kind; };   // kind 0 absent, 1 LZMA, 2 raw
