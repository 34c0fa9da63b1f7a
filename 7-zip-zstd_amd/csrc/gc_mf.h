// gc_mf.h -- geometry of the windowed match finder (gc_lz_window.hip), shared by host and device.
//
// Unit of independence = FRAME = F consecutive 128 KiB blocks (F <= 64, i.e. <= 8 MiB).  Matches reach back to the start of
// their frame.  For every position of a frame and for two key kinds (long = 8-byte hash, short = 5-byte hash) the finder
// delivers "the most recent earlier position of the frame with the same key" -- what a hash table of unbounded size that is
// updated position by position would return, the idea behind ZSTD_compressBlock_doubleFast's two tables
// (C/zstd/zstd_double_fast.c:105-323), Fast-LZMA2's per-dictionary-block match table (C/fast-lzma2/radix_engine.h:920-981)
// and brotli's H6 buckets (C/brotli/enc/hash_longest_match64_inc.h:157-290).  A table of that size cannot live in LDS and a
// table in HBM would be hit by random atomics, so the key space is partitioned instead (an MSD radix step):
//
//   W1 count    wave per 8 KiB tile: histogram of the tile's keys over 128 partitions (top hash bits)
//   W2 scan     per frame and kind: exclusive offsets in (partition, tile) order
//   W3 scatter  wave per tile: stable scatter of (position, key) entries -> every partition is a position-ordered list
//   W4 link     workgroup per (frame, kind, partition): streams its list through an LDS table (most recent wins) and replaces
//               every key by the previous position with that key
//   W5 parse    workgroup per block: gathers the candidates of a tile back into position order (LDS), verifies them against
//               the input, parses and emits literals + sequences (same steps as K1)
#pragma once
#include <stdint.h>
#include "gc_common.h"

#define GC_MF_TILE_LOG    13u
#define GC_MF_TILE        (1u << GC_MF_TILE_LOG)          // positions per tile
#define GC_MF_TILES_PER_BLOCK (GC_ZSTD_BLOCK_MAX >> GC_MF_TILE_LOG)
#define GC_MF_PART_LOG    7u
#define GC_MF_PARTS       (1u << GC_MF_PART_LOG)
#define GC_MF_KINDS       2u                              // 0 = long key, 1 = short key
#define GC_MF_MAX_FRAME_BLOCKS 64u                        // 8 MiB: frame-relative positions + 1 fit 24 bits

// one list entry: W3 writes {frame-relative position, 32-bit key}; W4 replaces `key` by (candidate position + 1), 0 = none
struct GcMfEntry { uint32_t pos; uint32_t key; };

struct GcMfGeom {
    uint32_t frameBlocks;     // F
    uint32_t nBlocks;
    uint32_t nFrames;
    uint32_t tilesPerFrame;   // F * 16
    uint64_t frameBytes;      // F * 128 KiB
    uint64_t entStride;       // entries per kind = nFrames * frameBytes
};

static inline GcMfGeom gc_mf_geom(uint64_t n, uint32_t frameBlocks)
{
    GcMfGeom g;
    g.frameBlocks = frameBlocks;
    g.nBlocks = gc_num_blocks(n);
    g.nFrames = (g.nBlocks + frameBlocks - 1u) / frameBlocks;
    g.tilesPerFrame = frameBlocks * GC_MF_TILES_PER_BLOCK;
    g.frameBytes = (uint64_t)frameBlocks * GC_ZSTD_BLOCK_MAX;
    g.entStride = (uint64_t)g.nFrames * g.frameBytes;
    return g;
}
// cnt / offsets: [frame][kind][partition][tile]   (uint32, tilesPerFrame fastest)
// partStart:     [frame][kind][GC_MF_PARTS + 1]
