// gc_common.h -- shared host/device definitions of the MI355X zstd block-parallel encoder.
//
// Pipeline (one 128 KiB zstd block = one single-segment frame = one workgroup per stage):
//
//   K1 gc_zstd_lz_kernel    match finding + parse      src block  -> raw sequences + literals   (gc_zstd_lz.hip)
//   K2 gc_zstd_huf_kernel   literals section (HUF)     literals   -> literals section bytes     (gc_zstd_huf.hip)
//   K3 gc_zstd_seq_kernel   sequences section (FSE)    sequences  -> sequences section bytes    (gc_zstd_seq.hip)
//   K4 gc_zstd_plan_kernel  frame sizes + offsets      sizes      -> exclusive scan             (gc_zstd_frame.hip)
//   K5 gc_zstd_emit_kernel  frame assembly / raw fallback         -> contiguous output          (gc_zstd_frame.hip)
//
// Reference functions replaced: ZSTD_compressBlock_doubleFast (zstd_double_fast.c:105-323),
// ZSTD_storeSeq/ZSTD_updateRep (zstd_compress_internal.h:776,818), HIST_count (hist.c:76),
// HUF_buildCTable/HUF_writeCTable/HUF_compress4X (huf_compress.c:756,248,1168), ZSTD_compressLiterals
// (zstd_compress_literals.c:129), ZSTD_seqToCodes (zstd_compress.c:2693), ZSTD_buildSequencesStatistics
// (:2763), FSE_normalizeCount/FSE_writeNCount/FSE_buildCTable_wksp (fse_compress.c:465,330,68),
// ZSTD_encodeSequences (zstd_compress_sequences.c:291-383), ZSTD_writeFrameHeader (:4695), block headers
// (:4655-4658).  The parse and the statistics heuristics are free choices; every
// format-normative step is reproduced exactly so that the reference decoder regenerates the input.
#pragma once
#include <stdint.h>

#define GC_ZSTD_BLOCK_MAX   (128u * 1024u)   // zstd block size limit (ZSTD_BLOCKSIZE_MAX)
#define GC_MIN_MATCH        5u               // shortest match the finder emits
#define GC_MATCH_CAP        64u              // per-position compare cap; longer matches are chained + merged in K3
#define GC_MAX_SEQ_PER_BLOCK (GC_ZSTD_BLOCK_MAX / GC_MIN_MATCH + 8u)

// K1 -> K2/K3 interface, per block b (strides in elements):
//   seqRaw[b * GC_MAX_SEQ_PER_BLOCK + i] = { litRank, (offset << 8) | matchLength }
//       litRank = number of literals that precede sequence i in the block's literal stream
//   lit[b * GC_ZSTD_BLOCK_MAX + k]       = k-th literal byte
//   meta[b] = { nSeqRaw, nLit }
struct GcSeqRaw { uint32_t litRank; uint32_t offml; };
struct GcBlockMeta { uint32_t nSeqRaw; uint32_t nLit; };

// K2/K3 results per block
struct GcSectionInfo {
    uint32_t litSecSize;   // bytes of the literals section (header included)
    uint32_t seqSecSize;   // bytes of the sequences section; 0xFFFFFFFF = not representable below the raw size
    uint32_t nSeq;         // merged sequence count (diagnostics)
    uint32_t flags;        // bit0: literals stored raw, bit1: literals RLE
};

// per-block strides of the section staging buffers
#define GC_LITSEC_STRIDE (GC_ZSTD_BLOCK_MAX + 64u)
#define GC_SEQSEC_STRIDE (GC_ZSTD_BLOCK_MAX + 1024u)

// worst-case frame: 4 magic + 1 FHD + 4 FCS + 3 block header + payload
#define GC_FRAME_OVERHEAD 12u

#define GC_SEQ_T      256u  // K3a / K3d: threads per block
#define GC_LZ_PHASES  7   // K1 phase profile slots: probe, insert, verify, double, chain, walk, emit
#define GC_SEQ_PHASES 9   // K3 phase profile slots: merge, codes, tables, chains, pack + chain sub-phases: stage, warm-up, walk, copy-out

// K3 between its kernels (gc_zstd_seq.hip)
#define GC_SEQ_CHAIN_TILE 4096u                                   // sequences per state-chain tile (64 lanes x 64)
#define GC_SEQ_ST_STRIDE  (((GC_MAX_SEQ_PER_BLOCK + GC_SEQ_CHAIN_TILE - 1u) / GC_SEQ_CHAIN_TILE) * GC_SEQ_CHAIN_TILE)   // states of one table of one block
struct GcSeqHist { uint32_t count[3][64]; };                      // K3a -> K3b: code histograms (LL, OF, ML)
struct GcSeqSymG { int32_t deltaFindState; uint32_t deltaNbBits; };
struct GcSeqTabG {                                                // K3b -> K3c, K3d: one FSE table of one block
    uint16_t state[512];
    GcSeqSymG tt[64];
    int16_t  norm[64];
    uint8_t  desc[96];                                            // table description bytes for the section header
    uint32_t descSize, mode, tableLog, maxSym, tabMaxSym, finalState;
    uint32_t pad[2];
};

static inline uint32_t gc_num_blocks(uint64_t n) { return (uint32_t)((n + GC_ZSTD_BLOCK_MAX - 1) / GC_ZSTD_BLOCK_MAX); }
