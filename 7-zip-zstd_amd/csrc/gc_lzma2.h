// gc_lzma2.h -- shared definitions of the FLZMA2 (LZMA2) GPU path.
#pragma once
#include <stdint.h>

#define GC_LZMA_LC 3u                       // lc=3 lp=0 pb=2: the reference's defaults (fl2_compress.c:116-138)
#define GC_LZMA_LP 0u
#define GC_LZMA_PB 2u
#define GC_LZMA_PROPS ((GC_LZMA_PB * 5u + GC_LZMA_LP) * 9u + GC_LZMA_LC)     // 0x5D

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

// Units of the FLZMA2 path (all sizes are powers of two and nest: rc chunk <= model segment <= 128 KiB match-finder block):
//   model segment  2^segLog bytes (level dependent, 16..128 KiB): the adaptive model runs through it sequentially and is
//                  reset at its start (LZMA2 control 0xC0 / 0xE0) -- the unit of parallelism of the model kernel (one wave)
//   rc chunk       4 KiB: one LZMA2 chunk (or, where the data compresses well, a group of up to 8 neighbours of one segment
//                  coded as one: GC_LZMA_RC_MERGE_WORDS).  The LZMA2 format restarts the range coder at every chunk anyway (and only the
//                  range coder: control 0x80 keeps probabilities, state and repeat distances), so the chunks of a segment can
//                  be range-coded independently once the model has resolved their probabilities -- the unit of parallelism of
//                  the range-coder kernel (one lane)
#define GC_LZMA_RC_LOG      12u
#define GC_LZMA_RC_SIZE     (1u << GC_LZMA_RC_LOG)
#define GC_LZMA_RC_PER_BLOCK (GC_ZSTD_BLOCK_MAX >> GC_LZMA_RC_LOG)
#define GC_LZMA_RC_STRIDE   (GC_LZMA_RC_SIZE + 1024u)     // bytes of range-coder output reserved per rc chunk (LZMA expands < 2 %)
#define GC_LZMA_RC_GROUP_MAX 32u                          // rc chunks that may be coded as one LZMA2 chunk: up to a whole 128 KiB segment where it compresses into GC_LZMA_RC_MERGE_WORDS coded bits
                                                          // (round 3; 8 before: on data of ratio 12 a chunk header + range-coder flush per 32 KiB was 0.4 % of the stream).  An LZMA2 chunk holds 2 MiB / 64 KiB coded.
#define GC_LZMA_RC_MERGE_WORDS 49152u                     // ... while their coded bits stay within this many words (4 KiB of literals cost 36864: 9 per byte)
#define GC_LZMA_SEG_LOG_MIN 14u
#define GC_LZMA_SEG_LOG_MAX 17u

// L1 -> L2: the block's symbols as a position-ordered item list, M[j] = pos | len << 18 | off << 34 (pos block-relative, 0..131072).
//   len >= 2  match piece: never crosses a 4 KiB boundary, never longer than 273 (the longest LZMA match)
//   len == 0  cut: no match, only closes the literal run in front of it (off repeats the previous match's distance so that the
//             repeat-distance scans can ignore it)
// Every item is preceded by at most GC_LZMA_LIT_CUT literals, and every 4 KiB boundary (and the block end) is the end of an
// item, so the events of an rc chunk are a contiguous range of its segment's event stream.
#define GC_LZMA_LIT_CUT     16u
#define GC_LZMA_MAX_ITEMS   (GC_ZSTD_BLOCK_MAX / 5u + GC_ZSTD_BLOCK_MAX / GC_LZMA_LIT_CUT + 2u * GC_LZMA_RC_PER_BLOCK + 64u)

// per rc chunk: usize 0 = the chunk does not exist; words [wordStart, wordEnd) of its segment's (p, bit) stream;
// csize = range-coder bytes, 0xFFFFFFFF = did not fit the staging area (the segment is stored)
struct GcLzmaChunkInfo { uint32_t usize; uint32_t csize; uint32_t wordStart; uint32_t wordEnd; };
// words of (p, bit) stream reserved per segment: a literal is the densest symbol (9 coded bits per byte); keeps 16-byte alignment
#define GC_LZMA_STREAM_WORDS(segLog) ((9u << (segLog)) + 64u)
struct GcLzmaPlan { uint64_t off; uint32_t size; uint32_t kind; };   // kind 0 absent, 1 LZMA, 2 raw
