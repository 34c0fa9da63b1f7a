// gc_mf.h -- geometry of the windowed match finder (gc_lz_window.hip), shared by host and device.
//
// Unit of independence = FRAME = F consecutive 128 KiB blocks (F <= 64, i.e. <= 8 MiB).  Matches reach back to the start of
// their frame.  For every position of a frame the finder delivers two candidates: the most recent earlier position of the frame
// whose 8 bytes hash to the same "long" key and the most recent one whose 5 bytes hash to the same "short" key -- what two hash
// tables updated position by position would return, the idea behind ZSTD_compressBlock_doubleFast's two tables
// (C/zstd/zstd_double_fast.c:105-323), Fast-LZMA2's per-dictionary-block match table (C/fast-lzma2/radix_engine.h:920-981)
// and brotli's H6 buckets (C/brotli/enc/hash_longest_match64_inc.h:157-290).  Tables of that size cannot live in LDS and a
// table in HBM would be hit by random atomics, so the key space is partitioned instead (an MSD radix step on the top bits of
// the SHORT hash: equal 8 bytes imply equal 5 bytes, so both keys of a position live in the same partition).
// 1024 partitions: a frame's list of a partition holds 8 Ki entries on average, so the private tables of W4 (2^12 + 2^11 slots
// per partition = 2^22 + 2^21 per frame for 2^23 positions) keep a candidate that lies MiBs back: a slot is overwritten after
// ~4 Ki insertions of its partition, i.e. ~4 MiB of input.  (Round 1 had 256 partitions = 2^20 + 2^19 slots per frame, and lost
// most candidates beyond ~1 MiB; measured with tools/lzma_parse_lab.c, LAB_TBITS: 2^20 -> 2^22 slots per key is worth 1.4-1.6 %
// of the FLZMA2 level-5 size, exact tables another 0.8 %.)
//
//   W1 count    workgroup per 16 KiB tile: histogram of the tile's positions over 1024 partitions
//   W2 scan     workgroup per frame: exclusive offsets in (partition, tile) order
//   W3 scatter  workgroup per tile: stable counting sort of the tile in LDS, then one coalesced run per partition of 8-byte
//               entries {position, long key, short key}
//   W4 link     ONE WAVE per (frame, partition, list segment): streams its position-ordered list through two private LDS
//               tables, 64 entries per step, one returning ds_max per table and step = the semantics of sequential insertion
//               (most recent wins); no barriers.  Rewrites every entry as {position in tile, long candidate, short candidate}
//   W5 verify   workgroup per tile, fully parallel: compares both candidates with the input, writes one match record
//               (offset << 8 | length, length <= GC_MATCH_CAP) per position
//   W6 parse    workgroup per block: greedy/lazy parse of the records by hierarchical composition of per-segment exit maps
//               (64 positions -> 2048 positions -> block), then literals + sequences are placed by ballots and prefix sums
#pragma once
#include <stdint.h>
#include "gc_common.h"

// Two geometries, compiled as two sets of W1..W5 kernels from the one source (gc_lz_window.hip; gc_lz_window_p8.hip defines GC_MF_FAST
// and includes it: the kernels get the suffix _p8):
//   wide  1024 partitions, 16 KiB tiles: 2^22 + 2^21 table slots per frame -- the levels that search far (FLZMA2, zstd >= 7, brotli >= 5)
//   fast   256 partitions,  8 KiB tiles: 2^20 + 2^19 slots per frame (still 8 x the reference's level-3 tables, clevels.h:31) -- zstd 3-6,
//          brotli 3-4; W3..W5 take 20 % less time for 1.3 % more size (run r2_b: 1 GB of text, 34.3 -> 27.4 ms of finder)
#define GC_MF_WIDE_TILE_LOG 14u
#define GC_MF_WIDE_PART_LOG 10u
#define GC_MF_WIDE_VERIFY_T 1024u
#define GC_MF_FAST_TILE_LOG 13u
#define GC_MF_FAST_PART_LOG 8u
#define GC_MF_FAST_VERIFY_T 512u
#ifdef GC_MF_FAST
#define GC_MF_TILE_LOG    GC_MF_FAST_TILE_LOG
#define GC_MF_PART_LOG    GC_MF_FAST_PART_LOG
#else
#define GC_MF_TILE_LOG    GC_MF_WIDE_TILE_LOG
#define GC_MF_PART_LOG    GC_MF_WIDE_PART_LOG
#endif
#define GC_MF_TILE        (1u << GC_MF_TILE_LOG)          // positions per tile
#define GC_MF_TILES_PER_BLOCK (GC_ZSTD_BLOCK_MAX >> GC_MF_TILE_LOG)
#define GC_MF_PARTS       (1u << GC_MF_PART_LOG)
#define GC_MF_MAX_FRAME_BLOCKS 64u                        // 8 MiB: frame-relative positions fit 23 bits (the fast geometry, and the default of the wide one)
#define GC_MF_WIDE_MAX_FRAME_BLOCKS 128u                  // 16 MiB: the wide geometry numbers positions with 24 bits (round 6: FLZMA2 7-9, zstd 20-22, brotli 9-11)

// W3 -> W4 entry (64 bit), fast geometry:  pos[0..22] | long key[23..42] (12-bit slot, 8-bit tag) | short key[43..61] (11-bit slot, 8-bit tag)
// W4 -> W5 entry (64 bit), fast geometry:  position in tile[0..12] | (long candidate + 1)[13..36] | (short candidate + 1)[37..60]   (0 = none;
//                           candidates are frame-relative).  The wide geometry: one more bit per position (below)
typedef uint64_t GcMfEntry;
// Round 6: the wide geometry has 24-bit positions (16 MiB frames).  W3 -> W4: pos[24] | long key[20] | short key[19] = 63 bits; W4's table words (position + 1) << 7 | tag[7]
// (25 + 7 = 32 bits: one tag bit less than the fast geometry's (position + 1) << 8 | tag[8]); W4 -> W5: position in tile[14] | (long candidate + 1)[25] | (short candidate + 1)[25] = 64 bits.
#ifdef GC_MF_FAST
#define GC_MF_POS_BITS    23u
#define GC_MF_TAG_BITS    8u
#else
#define GC_MF_POS_BITS    24u
#define GC_MF_TAG_BITS    7u
#endif
#define GC_MF_CAND_BITS   (GC_MF_POS_BITS + 1u)           // candidate + 1 (0 = none)
#define GC_MF_KL_BITS     20u
#define GC_MF_KS_BITS     19u
#ifndef GC_MF_LSLOT_LOG
#define GC_MF_LSLOT_LOG   12u                             // W4 long table: 2^12 slots per partition (2^20 per frame)
#endif
#ifndef GC_MF_SSLOT_LOG
#define GC_MF_SSLOT_LOG   11u                             // W4 short table: 2^11 slots per partition (2^19 per frame)
#endif

#define GC_MF_PARSE_T     1024u                           // W6: threads per block
#ifdef GC_MF_FAST
#define GC_MF_VERIFY_T    GC_MF_FAST_VERIFY_T
#else
#define GC_MF_VERIFY_T    GC_MF_WIDE_VERIFY_T             // W5: threads per tile (>= GC_MF_PARTS: one thread per run start)
#endif
#define GC_MF_LINK_SEGS   8u                              // W4: waves per (frame, partition): long lists are linked in segments

// Overlapping frames (round 5).  A frame is the finder's WINDOW: F blocks whose positions are numbered with 23 bits.  Without overlap the frames tile the input and a
// position's history is what lies between it and its frame's start -- 4 MiB on average of an 8 MiB frame, where the reference's window slides (zstd level 19: windowLog 23,
// clevels.h:47; ZSTDMT jobs overlap by a whole window at the btultra levels, zstdmt_compress.c:741-747; Fast-LZMA2 level 7: 64 MiB dictionaries, fl2_compress.c:52-63).
// Measured on the reference itself: zstd 19 on independent 8 MiB pieces against one stream, 32 MiB: text 1.030, lz-7zip 1.053, real sources 1.043 x.
// With a stride S < F the frames of a GROUP of C = F + k S blocks overlap: frame i of a group is the window [i S, i S + F) of the group's blocks; it lists and links all
// of its positions (W1..W4) but only VERIFIES the ones no earlier frame has (its last S blocks; frame 0: all F).  A position then sees between F - S and F blocks of
// history (from the group's first F blocks on).  Groups are independent of each other (the unit of sharding and, for zstd, the zstd frame).
// The kernels take the three numbers as ONE 32-bit argument (the old `frameBlocks`: a plain F means S = C = F):
#define GC_MF_GEOM_ARG(F, S, C) ((uint32_t)(F) | ((uint32_t)(S) << 8) | ((uint32_t)(C) << 16))
#define MF_F(a) ((a) & 0xFFu)
#define MF_S(a) ((((a) >> 8) & 0xFFu) ? (((a) >> 8) & 0xFFu) : MF_F(a))
#define MF_C(a) (((a) >> 16) ? ((a) >> 16) : MF_F(a))
#define MF_FPG(a) (1u + (MF_C(a) - MF_F(a)) / MF_S(a))          // frames per group

// W5 -> W6: one 32-bit match record per input position, (offset << 8) | length; 0 = no match
struct GcMfGeom {
    uint32_t tileLog, partLog, verifyT;    // which geometry (wide / fast)
    uint32_t frameBlocks;     // F
    uint32_t nBlocks;
    uint32_t nFrames;
    uint32_t tilesPerFrame;   // F * 16
    uint32_t nTiles;          // nFrames * tilesPerFrame (tiles past the end of the input are empty)
    uint64_t frameBytes;      // F * 128 KiB = entries per frame
    uint64_t cntWords;        // nFrames * (tilesPerFrame + 1) * GC_MF_PARTS
};

static inline GcMfGeom gc_mf_geom(uint64_t n, uint32_t frameArg, bool fast)
{
    GcMfGeom g;
    const uint32_t frameBlocks = MF_F(frameArg);
    g.tileLog = fast ? GC_MF_FAST_TILE_LOG : GC_MF_WIDE_TILE_LOG; g.partLog = fast ? GC_MF_FAST_PART_LOG : GC_MF_WIDE_PART_LOG;
    g.verifyT = fast ? GC_MF_FAST_VERIFY_T : GC_MF_WIDE_VERIFY_T;
    g.frameBlocks = frameBlocks;
    g.nBlocks = gc_num_blocks(n);
    g.nFrames = ((g.nBlocks + MF_C(frameArg) - 1u) / MF_C(frameArg)) * MF_FPG(frameArg);      // (no overlap: C = F, one frame per group)
    g.tilesPerFrame = frameBlocks * (GC_ZSTD_BLOCK_MAX >> g.tileLog);
    g.nTiles = g.nFrames * g.tilesPerFrame;
    g.frameBytes = (uint64_t)frameBlocks * GC_ZSTD_BLOCK_MAX;
    g.cntWords = (uint64_t)g.nFrames * (g.tilesPerFrame + 1u) << g.partLog;
    return g;
}
// cnt / offsets: [frame][tile 0..tilesPerFrame][partition]  (uint32, partition fastest).  After W2, row `tile` holds the
// frame-relative entry index where the tile's run of each partition starts; the extra row `tilesPerFrame` holds the partition
// ends, so run (tile, g) = [row[tile][g], row[tile + 1][g]) for every tile.
// entries: [frame][frameBytes]

// Static prices of one block for the price-based parse W7 (gc_lz_price.hip), written by W6 from the statistics of its own
// (greedy) parse of the block; uint16, units of 1/16 bit.  The same index space is used for the counts inside W6.
#define GC_PRICE_LIT    0u                 // [8 contexts = top 3 bits of the previous byte][256]
#define GC_PRICE_LEN    2048u              // [piece length 0..64] (0, 1 unused)
#define GC_PRICE_NLEN   80u
#define GC_PRICE_SLOT   (GC_PRICE_LEN + GC_PRICE_NLEN)            // [LZMA distance slot 0..63] (without the footer bits)
#define GC_PRICE_FLAGS  (GC_PRICE_SLOT + 64u)                     // literal flag, match flag
#define GC_PRICE_REPLEN (GC_PRICE_FLAGS + 8u)                     // W7L: [length 0..79] of a repeat match (LZMA codes those with a coder of their own, LzmaEnc.c repLenEnc)
#define GC_PRICE_WORDS  (GC_PRICE_REPLEN + GC_PRICE_NLEN)         // 2280: a multiple of 8 (16-byte rows)
#define GC_PRICE_MAX    240u               // 15 bits: literal + flag of 4096 positions stay below 2^21 units (the cost field of a W7 node)
// Symbol counts of the price-based parse's own path (W7 phase A, a sample of the windows of every block): what phase B prices
// lengths, distance slots and the literal / match flag with.  uint32 per block.
#define GC_DPS_LEN      0u                 // [piece length 0..79]
#define GC_DPS_SLOT     GC_PRICE_NLEN      // [distance slot 0..63]
#define GC_DPS_NLIT     (GC_DPS_SLOT + 64u)
#define GC_DPS_NMAT     (GC_DPS_NLIT + 1u)
#define GC_DPS_NREP      (GC_DPS_NLIT + 2u)     // repeats of >= 2 bytes found through the path's own distance
#define GC_DPS_NSREP     (GC_DPS_NLIT + 3u)     // short repeats (one byte)
#define GC_DPS_NREP1     (GC_DPS_NLIT + 4u)     // W7L: repeats of the second / third / fourth last distance (rep1, rep2, rep3: consecutive words)
#define GC_DPS_NREP2     (GC_DPS_NLIT + 5u)
#define GC_DPS_NREP3     (GC_DPS_NLIT + 6u)
#define GC_DPS_REPLEN   160u               // [length 0..79] of the repeat matches
#define GC_DPS_WORDS    240u
// Phase B of a block runs in W7L (a lane per window, the repeat distances at every node) where phase A's paths repeated a distance in at least one match symbol of
// twenty, in W7 (a wave per window, a third of the time) elsewhere: bit 5 of both kernels' phase argument = "only my kind of block".  C = the block's counts.
#ifndef GC_DPL_THREADS
#define GC_DPL_THREADS  128u              // W7L: two waves per group of 64 windows (gc_lz_dpl.hip; 64 = one wave does everything; 192 = a third wave for the literal, the
                                           // capped rest and the short candidate: measured 39.2 ms against 23.9 -- the kernel then runs in two rounds, run r4k3)
#endif
#define GC_DP_SELECT    32u
#define GC_DP_ALLLEN    128u              // W7L (zstd): every length of a candidate is an edge, not only the short ones and the last four (gc_lz_dpl.hip relax())
#define GC_DPS_RICH(C)  ((C)[GC_DPS_NMAT] != 0u && ((C)[GC_DPS_NREP] + (C)[GC_DPS_NSREP] + (C)[GC_DPS_NREP1] + (C)[GC_DPS_NREP2] + (C)[GC_DPS_NREP3]) * 20u >= (C)[GC_DPS_NMAT])
#define GC_SHORT_NONE   0xFFFFu            // W5s -> W7: uint16 per position, (distance - 1) << 4 | (length - 2), or none

// Workgroup index -> work item such that each of the 8 XCDs (workgroups are dealt round-robin to XCDs) owns one contiguous
// range of items: neighbouring tiles / blocks then share an L2.  The grid is 8 * per workgroups, per = ceil(n / 8).
#define GC_XCDS 8u
static inline uint32_t gc_xcd_per(uint32_t n) { return (n + GC_XCDS - 1u) / GC_XCDS; }
