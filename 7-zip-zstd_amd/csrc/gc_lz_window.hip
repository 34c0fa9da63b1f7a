// gc_lz_window.hip -- the windowed match finder W1..W6 (geometry, entry formats and overview: gc_mf.h).
//
// Serves the higher levels of all three codecs: matches reach back to the start of their frame (<= 8 MiB) instead of the start
// of their 128 KiB block.  Replaces, per position, the table lookups of ZSTD_compressBlock_doubleFast
// (C/zstd/zstd_double_fast.c:105-323: long + short table, most recent position wins), and stands in for RMF_buildTable /
// RMF_getMatch (C/fast-lzma2/radix_engine.h:920, radix_get.h:60) and brotli's H6 FindLongestMatch
// (C/brotli/enc/hash_longest_match64_inc.h:157-290) as the source of candidates; the parse replaces the greedy / one-step-lazy
// loops around them (zstd_double_fast.c:169-300, backward_references_inc.h:80-130).
//
// Why partition: "most recent earlier position with the same key" is a sequential notion.  For a multi-MiB window the tables
// do not fit LDS, and tables in HBM would be hammered by random atomics from all CUs (one 128-byte line per 4-byte update).
// Splitting the key space 256 ways turns the problem into 256 independent position-ordered lists per frame, each served by
// ONE WAVE with private LDS tables: the wave reproduces sequential insertion (64 entries per step, one returning ds_max per
// table: the LDS unit itself serialises the lanes that share a slot), so there are no barriers and no cross-wave atomics, and
// every byte the passes move through HBM is a coalesced run.
//
// Determinism: W3 is a stable counting sort (ranks from ballots), W4 is sequential semantics by construction, W5 and W6 are
// pure functions of their inputs -- the compressed bytes do not depend on wave timing.
//
// HBM traffic per input byte: W1 1 R; W3 1 R + 8 W; W4 8 R + 8 W (two entry arrays); W5 8 R + 4 W (+ candidate windows, mostly L2);
// W6 8 R (records, twice) + sequences and literals out.
#include "gc_mf.h"
#include "gc_lz_parse.h"
#ifdef HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif

// kernel names of this geometry (gc_mf.h: the fast geometry is compiled from gc_lz_window_p8.hip with the suffix _p8)
#ifdef GC_MF_FAST
#define MFK(name) name##_p8
#else
#define MFK(name) name
#endif

// everything below depends on the geometry: one namespace per geometry, so that the two copies of the helpers never meet at link time
namespace MFK(gc_mf_ns) {

#define MF_T         GC_MF_PARTS      // W1 / W3: threads per tile (one per partition)
#define MF_WAVES     (MF_T / 64u)
static_assert(MF_T == GC_MF_PARTS, "W1/W3 use one thread per partition for the histogram rows");
#define MF_STAGE_PAD 16u              // bytes staged in front of / behind the tile
#define MF_STAGE_WORDS ((GC_MF_TILE + 2u * MF_STAGE_PAD + 32u) / 4u)     // (+ 32: the keys of MF_FAR2 read 32 bytes from a position)

__device__ __forceinline__ uint32_t mf_item(uint32_t bid, uint32_t per) { return (bid & (GC_XCDS - 1u)) * per + (bid >> 3); }

struct MfTile {
    uint32_t frame, tif;              // frame index, tile in frame
    uint64_t frameStart, frameEnd;    // absolute
    uint64_t tileStart;               // absolute
    uint32_t len;                     // positions of the tile that exist (0: the tile lies past the end of the input)
    bool own;                         // the frame verifies this tile (overlapping frames, gc_mf.h: false = the tile is only listed and linked here, an earlier frame has its records)
};
// frameBlocks: F, or F | S << 8 | C << 16 (gc_mf.h GC_MF_GEOM_ARG)
__device__ __forceinline__ MfTile mf_tile(uint32_t tile, uint32_t frameBlocks, uint64_t srcSize)
{
    MfTile T;
    const uint32_t F = MF_F(frameBlocks), S = MF_S(frameBlocks), C = MF_C(frameBlocks), fpg = 1u + (C - F) / S;
    const uint32_t TPF = F * GC_MF_TILES_PER_BLOCK;
    const uint64_t frameBytes = (uint64_t)F * GC_ZSTD_BLOCK_MAX;
    T.frame = tile / TPF; T.tif = tile % TPF;
    const uint32_t gi = T.frame / fpg, fi = T.frame % fpg;
    const uint64_t groupStart = (uint64_t)gi * C * GC_ZSTD_BLOCK_MAX;
    const uint64_t groupEnd = (groupStart + (uint64_t)C * GC_ZSTD_BLOCK_MAX) < srcSize ? (groupStart + (uint64_t)C * GC_ZSTD_BLOCK_MAX) : srcSize;
    T.frameStart = groupStart + (uint64_t)fi * S * GC_ZSTD_BLOCK_MAX;
    T.frameEnd = (T.frameStart + frameBytes) < groupEnd ? (T.frameStart + frameBytes) : groupEnd;
    T.tileStart = T.frameStart + (uint64_t)T.tif * GC_MF_TILE;
    T.len = T.tileStart < T.frameEnd ? (uint32_t)((T.frameEnd - T.tileStart) < GC_MF_TILE ? (T.frameEnd - T.tileStart) : GC_MF_TILE) : 0u;
    T.own = fi == 0u || T.tif >= (F - S) * GC_MF_TILES_PER_BLOCK;
    return T;
}

// Stage input bytes [tileStart - 16, tileStart - 16 + 4 * nWords) into LDS (zero outside the input) with 16-byte loads.
// LDS byte i <-> input position tileStart - 16 + i.
__device__ __forceinline__ void mf_stage(uint32_t* sW, uint32_t nWords, const uint8_t* __restrict__ src, uint64_t srcSize, uint64_t tileStart, uint32_t t, uint32_t nThreads)
{
    for (uint32_t c = t; c < nWords / 4u; c += nThreads) {
        GcU4 v; v.x = v.y = v.z = v.w = 0;
        if (tileStart + 16ull * c >= MF_STAGE_PAD) {
            const uint64_t pos = tileStart + 16ull * c - MF_STAGE_PAD;
            if (pos + 16u <= srcSize) __builtin_memcpy(&v, src + pos, 16);
            else if (pos < srcSize) { uint8_t tmp[16]; for (uint32_t i = 0; i < 16u; i++) tmp[i] = pos + i < srcSize ? src[pos + i] : (uint8_t)0; __builtin_memcpy(&v, tmp, 16); }
        }
        sW[4u * c] = v.x; sW[4u * c + 1u] = v.y; sW[4u * c + 2u] = v.z; sW[4u * c + 3u] = v.w;
    }
}
// bytes i .. i+7 of the staged tile: three aligned LDS words + two funnel shifts (v_alignbit).  (Round 6 measured the obvious alternative: gfx950 accepts LDS reads at
// any byte address and hipcc turns an unaligned 8-byte memcpy into ONE ds_read_b64 -- 19 % fewer instructions in the fused verify + parse kernel, the same bytes
// out -- but the hardware serves a misaligned ds_read_b64 several times slower than three aligned words: W1 0.66 -> 1.97 ms, W3 3.25 -> 6.23 ms, the fused kernel
// unchanged, zstd-L3 on 1 GB 45.3 -> 37.7 GB/s, run s1.  Aligned words it is.)
__device__ __forceinline__ uint64_t mf_lds_ld64(const uint32_t* sW, uint32_t i)
{
    const uint32_t w = i >> 2, sh = (i & 3u) * 8u;
    const uint32_t w0 = sW[w], w1 = sW[w + 1u], w2 = sW[w + 2u];
    const uint32_t lo = (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh);
    const uint32_t hi = (uint32_t)((((uint64_t)w2 << 32) | w1) >> sh);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t mf_lds_byte(const uint32_t* sW, uint32_t i) { return (sW[i >> 2] >> ((i & 3u) * 8u)) & 0xFFu; }

// Is tile position q listed, and with which keys?  Not listed:
//   - no full compare window (GC_MATCH_CAP + 16 bytes) left in the FRAME: such positions never match.  The limit is the frame
//     end, not the input end, so that a frame's bytes do not depend on what follows it (a frame-aligned range shard produces
//     exactly the frames the whole input would)
//   - inside a run of one byte value (the 8 bytes at P equal the 8 bytes at P-1): all those positions share one key and would
//     pile into one list; W5 gives them the candidate P-1 instead, which is what the tables would have returned
// FAR = the second pass of the finder at higher levels: the same machinery with keys of 16 ("long") and 12 ("short") bytes.
// The most recent position with the same 16 bytes is usually a LONGER match than the most recent one with the same 8 bytes; it
// is what brings the candidates close to the longest match the reference's exhaustive structures return (RMF_buildTable
// radix_engine.h:920: depth 42 at level 5; ZSTD_insertBtAndGetAllMatches zstd_opt.c:590).  Equal 16 bytes imply equal 12 bytes,
// so both keys of a position again live in one partition.
// MODE 2 = the third pass at the levels that parse by price: keys of 4 ("long") and 3 ("short") bytes, i.e. the nearest
// occurrence of a position's first 4 / 3 bytes anywhere in the frame -- the short matches that the reference's structures
// deliver at every position (2-byte radix heads radix_engine.h:106-171; ZSTD's hash3 table zstd_opt.c:408) and that pay where
// literals are expensive.  Its records (lengths from 3) only feed the price-based parse W7.
#define MF_BASE  0
#define MF_FAR   1
#define MF_SHORT 2
// MF_FAR2 = one more pass of the far kind with keys of 32 ("long") and 24 ("short") bytes (round 5: zstd levels >= 7; round 6: >= 5; FLZMA2 levels >= 7): in data made of many near-copies of
// the same text (source trees, archives of similar files) the most recent position with the same 16 bytes is often a copy that diverges a few dozen bytes later, where the
// reference's chains and trees return the LONGEST match (real sources, zstd level 9 and 19: 1.12 x the reference with 10 % more sequences of the same cost each).
#define MF_FAR2  4
struct MfKeys { bool ok, run; uint32_t part; uint64_t entry; };
template <int MODE>
__device__ __forceinline__ MfKeys mf_keys(const uint32_t* sW, uint32_t q, const MfTile& T)
{
    MfKeys r; r.ok = false; r.run = false; r.part = 0; r.entry = 0;
    const uint64_t P = T.tileStart + q;
    const uint64_t x = mf_lds_ld64(sW, q + MF_STAGE_PAD);
    if (P > T.frameStart) r.run = ((x << 8) | (uint64_t)mf_lds_byte(sW, q + MF_STAGE_PAD - 1u)) == x;
    if (q < T.len && P + GC_MATCH_CAP + 16u <= T.frameEnd && !r.run) {
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        uint32_t hL = lz_hash_long(lo, hi), hS = lz_hash_short(lo, hi);
        if (MODE == MF_SHORT) {
            // three equal bytes are not listed in this pass (round 5): the hottest 3-byte keys of machine code and tables (00 00 00, FF FF FF, CC CC CC) sit in ONE partition each,
            // whose last list segment then is the tail of W4 -- 211.9 MB of shared objects at FLZMA2 level 5: this pass 16.5 -> 8.5 ms (the call 103.7 -> 95.5 ms) for +0.003 % size;
            // such a position keeps what the other passes found and the near candidates of W5s
            if (((lo ^ (lo >> 8)) & 0xFFFFu) == 0u) return r;
            hS = (lo & 0xFFFFFFu) * 0x9E3779B1u; hS ^= hS >> 15; hS *= 0x2C1B3C6Du;     // bytes 0..2
            hL = lo * 0x9E3779B1u; hL ^= hL >> 15; hL *= 0x85EBCA77u;                   // bytes 0..3
        }
        if (MODE == MF_FAR2) {
            const uint64_t y = mf_lds_ld64(sW, q + MF_STAGE_PAD + 8u), z = mf_lds_ld64(sW, q + MF_STAGE_PAD + 16u), w = mf_lds_ld64(sW, q + MF_STAGE_PAD + 24u);
            uint32_t h = hL + (uint32_t)y * 0xC2B2AE3Du; h ^= h >> 15; h *= 0x2C1B3C6Du;
            h += (uint32_t)(y >> 32) * 0x27D4EB2Fu; h ^= h >> 13; h *= 0x165667B1u;
            h += (uint32_t)z * 0x9E3779B1u; h ^= h >> 16; h *= 0x85EBCA6Bu;
            hS = h + (uint32_t)(z >> 32) * 0xC2B2AE35u;           // bytes 0..23
            hS ^= hS >> 15; hS *= 0x2C1B3C6Du;
            hL = hS + (uint32_t)w * 0x27D4EB2Fu; hL ^= hL >> 13; hL *= 0x165667B1u;
            hL += (uint32_t)(w >> 32) * 0x9E3779B1u;              // bytes 0..31
            hL ^= hL >> 16; hL *= 0x85EBCA77u;
        }
        if (MODE == MF_FAR) {
            const uint64_t y = mf_lds_ld64(sW, q + MF_STAGE_PAD + 8u);
            const uint32_t lo1 = (uint32_t)y, hi1 = (uint32_t)(y >> 32);
            hS = hL + lo1 * 0xC2B2AE3Du;                          // bytes 0..11
            hS ^= hS >> 15; hS *= 0x2C1B3C6Du;
            hL = hS + hi1 * 0x27D4EB2Fu;                          // bytes 0..15
            hL ^= hL >> 13; hL *= 0x165667B1u;
        }
        r.ok = true;
        r.part = hS >> (32u - GC_MF_PART_LOG);
        r.entry = (uint64_t)(uint32_t)(P - T.frameStart)
                | ((uint64_t)(hL >> (32u - GC_MF_KL_BITS)) << GC_MF_POS_BITS)
                | ((uint64_t)((hS >> (32u - GC_MF_PART_LOG - GC_MF_KS_BITS)) & ((1u << GC_MF_KS_BITS) - 1u)) << (GC_MF_POS_BITS + GC_MF_KL_BITS));
    }
    return r;
}

// ------------------------------------------------------------------------------------------------ W1 count
template <int MODE>
__device__ __forceinline__ void mf_count_body(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per, uint32_t* __restrict__ cnt)
{
    __shared__ uint32_t sW[MF_STAGE_WORDS];
    __shared__ uint32_t sHist[GC_MF_PARTS];
    const uint32_t t = threadIdx.x;
    const uint32_t tile = mf_item(blockIdx.x, per);
    if (tile >= nTiles) return;
    const MfTile T = mf_tile(tile, frameBlocks, srcSize);
    const uint32_t TPF = MF_F(frameBlocks) * GC_MF_TILES_PER_BLOCK;
    sHist[t] = 0;
    if (T.len) mf_stage(sW, MF_STAGE_WORDS, src, srcSize, T.tileStart, t, MF_T);
    __syncthreads();
    for (uint32_t q = t; q < T.len; q += MF_T) {
        const MfKeys k = mf_keys<MODE>(sW, q, T);
        if (k.ok) atomicAdd(&sHist[k.part], 1u);
    }
    __syncthreads();
    cnt[((uint64_t)T.frame * (TPF + 1u) + T.tif) * GC_MF_PARTS + t] = sHist[t];
}
extern "C" __global__ void __launch_bounds__(MF_T)
MFK(gc_mf_count_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per, uint32_t* __restrict__ cnt)
{
    mf_count_body<MF_BASE>(src, srcSize, frameBlocks, nTiles, per, cnt);
}
extern "C" __global__ void __launch_bounds__(MF_T)
MFK(gc_mf_count_short_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per, uint32_t* __restrict__ cnt)
{
    mf_count_body<MF_SHORT>(src, srcSize, frameBlocks, nTiles, per, cnt);
}
extern "C" __global__ void __launch_bounds__(MF_T)
MFK(gc_mf_count_far_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per, uint32_t* __restrict__ cnt)
{
    mf_count_body<MF_FAR>(src, srcSize, frameBlocks, nTiles, per, cnt);
}
extern "C" __global__ void __launch_bounds__(MF_T)
MFK(gc_mf_count_far2_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per, uint32_t* __restrict__ cnt)
{
    mf_count_body<MF_FAR2>(src, srcSize, frameBlocks, nTiles, per, cnt);
}

// ------------------------------------------------------------------------------------------------ W2 scan
// One workgroup per frame: counts[tile][partition] -> exclusive offsets in (partition, tile) order, in place; row
// `tilesPerFrame` receives the partition ends.  Thread (q, g) walks quarter q of the tiles for partition g (row reads are
// coalesced over g).
extern "C" __global__ void __launch_bounds__(1024)
MFK(gc_mf_scan_kernel)(uint32_t* __restrict__ cnt, uint32_t tilesPerFrame)
{
    constexpr uint32_t Q = 1024u / GC_MF_PARTS;                   // groups of tile rows walked side by side (1 with 1024 partitions)
    constexpr uint32_t NW = GC_MF_PARTS / 64u;                    // waves that hold one partition per lane
    __shared__ uint32_t sPart[Q][GC_MF_PARTS];
    __shared__ uint32_t sStart[GC_MF_PARTS];
    __shared__ uint32_t sWave[NW];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, q = t >> GC_MF_PART_LOG, g = t & (GC_MF_PARTS - 1u);
    uint32_t* A = cnt + (uint64_t)blockIdx.x * (tilesPerFrame + 1u) * GC_MF_PARTS;
    const uint32_t R = (tilesPerFrame + Q - 1u) / Q;
    const uint32_t r0 = q * R < tilesPerFrame ? q * R : tilesPerFrame, r1 = (q + 1u) * R < tilesPerFrame ? (q + 1u) * R : tilesPerFrame;
    uint32_t sum = 0;
    for (uint32_t r = r0; r < r1; r++) sum += A[(uint64_t)r * GC_MF_PARTS + g];
    sPart[q][g] = sum;
    __syncthreads();
    uint32_t tot = 0, incl = 0;
    if (t < GC_MF_PARTS) {
        for (uint32_t qq = 0; qq < Q; qq++) tot += sPart[qq][t];
        incl = gc_wave_incl_sum(tot);
        if (lane == 63u) sWave[wave] = incl;
    }
    __syncthreads();
    if (t < GC_MF_PARTS) {
        uint32_t before = 0;
        for (uint32_t w = 0; w < NW; w++) if (w < wave) before += sWave[w];
        sStart[t] = before + incl - tot;
    }
    __syncthreads();
    uint32_t run = sStart[g];
    for (uint32_t qq = 0; qq < Q; qq++) if (qq < q) run += sPart[qq][g];
    for (uint32_t r = r0; r < r1; r++) {
        const uint64_t i = (uint64_t)r * GC_MF_PARTS + g;
        const uint32_t v = A[i]; A[i] = run; run += v;
    }
    if (q == Q - 1u) A[(uint64_t)tilesPerFrame * GC_MF_PARTS + g] = run;     // end of partition g
}

// ------------------------------------------------------------------------------------------------ W3 scatter
// Stable counting sort of one tile by partition, staged through LDS as a permutation (16-bit tile positions), so that every
// partition's run leaves the CU as one contiguous, coalesced store stream.  Wave w owns quarter w of the tile; ranks inside a
// 64-position round come from ballots (position order = lane order), so the order inside a partition is position order.
template <int MODE>
__device__ __forceinline__ void mf_scatter_body(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                     const uint32_t* __restrict__ offs, GcMfEntry* __restrict__ ent)
{
    __shared__ uint32_t sW[MF_STAGE_WORDS];
    __shared__ uint32_t sRun[MF_WAVES][GC_MF_PARTS];              // pass A: counts; pass B: next free slot of (wave, partition)
    __shared__ uint32_t sLocal[GC_MF_PARTS], sGlob[GC_MF_PARTS];
    __shared__ uint32_t sWaveTot[MF_WAVES];
    __shared__ uint16_t sPerm[GC_MF_TILE];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t tile = mf_item(blockIdx.x, per);
    if (tile >= nTiles) return;
    const MfTile T = mf_tile(tile, frameBlocks, srcSize);
    if (T.len == 0u) return;
    const uint32_t TPF = MF_F(frameBlocks) * GC_MF_TILES_PER_BLOCK;
    for (uint32_t w = 0; w < MF_WAVES; w++) sRun[w][t] = 0;
    mf_stage(sW, MF_STAGE_WORDS, src, srcSize, T.tileStart, t, MF_T);
    __syncthreads();
    const uint32_t qBase = wave * (GC_MF_TILE / MF_WAVES);
    // pass A: per-wave histograms
    for (uint32_t r = 0; r < GC_MF_TILE / MF_T; r++) {                // (a wave owns GC_MF_TILE / MF_WAVES positions = GC_MF_TILE / MF_T rounds of 64)
        const MfKeys k = mf_keys<MODE>(sW, qBase + r * 64u + lane, T);
        if (k.ok) atomicAdd(&sRun[wave][k.part], 1u);
    }
    __syncthreads();
    // offsets: thread t = partition t
    uint32_t c[MF_WAVES], tot = 0;
    for (uint32_t w = 0; w < MF_WAVES; w++) { c[w] = sRun[w][t]; tot += c[w]; }
    const uint32_t incl = gc_wave_incl_sum(tot);
    if (lane == 63u) sWaveTot[wave] = incl;
    __syncthreads();
    {
        uint32_t before = 0;
        for (uint32_t w = 0; w < MF_WAVES; w++) if (w < wave) before += sWaveTot[w];
        uint32_t ls = before + incl - tot;
        sLocal[t] = ls;
        for (uint32_t w = 0; w < MF_WAVES; w++) { sRun[w][t] = ls; ls += c[w]; }
        sGlob[t] = offs[((uint64_t)T.frame * (TPF + 1u) + T.tif) * GC_MF_PARTS + t];
    }
    __syncthreads();
    uint32_t nEnt = 0;
    for (uint32_t w = 0; w < MF_WAVES; w++) nEnt += sWaveTot[w];
    // pass B: stable ranks -> permutation
    for (uint32_t r = 0; r < GC_MF_TILE / MF_T; r++) {
        const uint32_t q = qBase + r * 64u + lane;
        const MfKeys k = mf_keys<MODE>(sW, q, T);
        // one returning ds_add per position: the LDS unit serves the lanes that hit one counter in lane order (the property W4's ds_max
        // relies on), lanes are positions, rounds follow each other in program order -- so the value returned is the position's stable rank
        if (k.ok) sPerm[atomicAdd(&sRun[wave][k.part], 1u)] = (uint16_t)q;
        gc_wave_step();
    }
    __syncthreads();
    // output: slot j of the sorted tile -> its partition's run in HBM
    GcMfEntry* E = ent + (uint64_t)T.frame * ((uint64_t)MF_F(frameBlocks) * GC_ZSTD_BLOCK_MAX);
    for (uint32_t j = t; j < nEnt; j += MF_T) {
        const MfKeys k = mf_keys<MODE>(sW, sPerm[j], T);
        E[sGlob[k.part] + (j - sLocal[k.part])] = k.entry;
    }
}
extern "C" __global__ void __launch_bounds__(MF_T)
MFK(gc_mf_scatter_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                     const uint32_t* __restrict__ offs, GcMfEntry* __restrict__ ent)
{
    mf_scatter_body<MF_BASE>(src, srcSize, frameBlocks, nTiles, per, offs, ent);
}
extern "C" __global__ void __launch_bounds__(MF_T)
MFK(gc_mf_scatter_short_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                           const uint32_t* __restrict__ offs, GcMfEntry* __restrict__ ent)
{
    mf_scatter_body<MF_SHORT>(src, srcSize, frameBlocks, nTiles, per, offs, ent);
}
extern "C" __global__ void __launch_bounds__(MF_T)
MFK(gc_mf_scatter_far_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                         const uint32_t* __restrict__ offs, GcMfEntry* __restrict__ ent)
{
    mf_scatter_body<MF_FAR>(src, srcSize, frameBlocks, nTiles, per, offs, ent);
}
extern "C" __global__ void __launch_bounds__(MF_T)
MFK(gc_mf_scatter_far2_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                         const uint32_t* __restrict__ offs, GcMfEntry* __restrict__ ent)
{
    mf_scatter_body<MF_FAR2>(src, srcSize, frameBlocks, nTiles, per, offs, ent);
}

// ------------------------------------------------------------------------------------------------ W4 link
// One wave per (frame, partition, segment).  Per step of 64 list entries, for each of the two keys, ONE returning ds_max:
// the table entry is (position + 1) << 8 | tag and positions grow along the list, so the maximum is the most recent insertion;
// the LDS unit serialises the lanes that hit the same slot, and the value returned to a lane is the slot's content when its
// turn came -- the state left by all earlier steps and by the lower lanes of this step, i.e. what a table updated entry by
// entry would hold when the entry arrives.  (Should a lane ever be served before a lower one, it would see a position that is
// not earlier than its own; such a result is discarded, so every candidate handed on is a genuine earlier position.)
// A candidate is accepted only if its 8-bit tag agrees.
//
// Lists longer than LINK_SEG entries (hot keys: one 5-gram can own 2 % of a text, and every 8-byte key that starts with it
// shares its partition) are cut into segments linked by different waves, so the longest list does not become the tail of the
// launch.  A later segment first replays the LINK_WARM entries in front of it, insert only: a slot of a 2^12-entry table is
// overwritten after ~4 Ki insertions on average, so the tables are then what the single walk would have left, up to slots not
// touched for 16 Ki insertions (e^-4 of them), whose stale candidates are simply not offered.  Output goes to a second entry
// array because the replay reads what the previous segment's wave is working on.
#ifndef LINK_DEPTH
#define LINK_DEPTH 8u                 // steps per register set
#endif
#ifdef GC_MF_FAST
#define LINK_SEG   49152u             // entries per segment (1.5 x the mean list length of a full 8 MiB frame: 256 partitions)
#else
#define LINK_SEG   16384u             // entries per segment (2 x the mean list length of a full 8 MiB frame: 1024 partitions)
#endif
#define LINK_WARM  16384u             // entries replayed in front of a segment
#define LINK_SEGS  GC_MF_LINK_SEGS                 // segments per list; the last one takes whatever is left
#define LINK_BATCH 16u                             // items of the upper segments per ticket

// what the table held when the entry arrived -> candidate position + 1, or 0
__device__ __forceinline__ uint32_t mf_link_pick(uint32_t seen, uint32_t mine)
{
    return (seen != 0u && seen < mine && ((seen ^ mine) & ((1u << GC_MF_TAG_BITS) - 1u)) == 0u) ? (seen >> GC_MF_TAG_BITS) : 0u;
}

__device__ __forceinline__ void mf_link_load(uint64_t q[LINK_DEPTH], const GcMfEntry* __restrict__ E, uint32_t s0, uint32_t end, uint32_t lane)
{
#pragma unroll
    for (uint32_t d = 0; d < LINK_DEPTH; d++) {                   // unconditional (index clamped): the loads stay countable, so the
        const uint32_t i = s0 + d * 64u + lane;                   // waits in the caller are for exactly the set that is needed
        q[d] = E[i < end ? i : end - 1u];
    }
}

// LINK_DEPTH steps.  All their table operations are issued back to back (the LDS unit executes the operations of a wave in
// order, so the tables see the steps in order) and their results are collected afterwards: one LDS latency per call instead of
// two per step.
__device__ __forceinline__ void mf_link_steps(const uint64_t q[LINK_DEPTH], uint32_t* tabL, uint32_t* tabS, GcMfEntry* __restrict__ EO, uint32_t s0,
                                              uint32_t end, uint32_t lane)
{
    if (s0 >= end) return;                                        // uniform
    uint32_t pos[LINK_DEPTH], mL[LINK_DEPTH], mS[LINK_DEPTH], rL[LINK_DEPTH], rS[LINK_DEPTH];
#pragma unroll
    for (uint32_t d = 0; d < LINK_DEPTH; d++) {
        const uint32_t i = s0 + d * 64u + lane;
        const uint64_t e = q[d];
        pos[d] = (uint32_t)e & ((1u << GC_MF_POS_BITS) - 1u);
        const uint32_t kL = (uint32_t)(e >> GC_MF_POS_BITS) & ((1u << GC_MF_KL_BITS) - 1u);
        const uint32_t kS = (uint32_t)(e >> (GC_MF_POS_BITS + GC_MF_KL_BITS)) & ((1u << GC_MF_KS_BITS) - 1u);
        mL[d] = ((pos[d] + 1u) << GC_MF_TAG_BITS) | (kL & ((1u << GC_MF_TAG_BITS) - 1u)); mS[d] = ((pos[d] + 1u) << GC_MF_TAG_BITS) | (kS & ((1u << GC_MF_TAG_BITS) - 1u));
        rL[d] = 0; rS[d] = 0;
        if (i < end) { rL[d] = atomicMax(&tabL[kL >> (GC_MF_KL_BITS - GC_MF_LSLOT_LOG)], mL[d]); rS[d] = atomicMax(&tabS[kS >> (GC_MF_KS_BITS - GC_MF_SSLOT_LOG)], mS[d]); }
        gc_wave_step();
    }
#pragma unroll
    for (uint32_t d = 0; d < LINK_DEPTH; d++) {
        const uint32_t i = s0 + d * 64u + lane;
        const uint32_t cL = mf_link_pick(rL[d], mL[d]), cS = mf_link_pick(rS[d], mS[d]);
        if (i < end) EO[i] = (uint64_t)(pos[d] & (GC_MF_TILE - 1u)) | ((uint64_t)cL << GC_MF_TILE_LOG) | ((uint64_t)cS << (GC_MF_TILE_LOG + GC_MF_CAND_BITS));
    }
}

// The launch is PERSISTENT: as many one-wave workgroups as the chip holds at once (the tables of a wave take 24 KiB of LDS: six per CU), each
// drawing work items (frame, partition, segment) from a ticket counter until none is left -- seven of eight items are segments that do not
// exist, and a launch of one short-lived workgroup per item left the CUs a wave or two each (run 33: 1.6 resident waves per CU on average).
extern "C" __global__ void __launch_bounds__(64)
MFK(gc_mf_link_kernel)(const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent, GcMfEntry* __restrict__ entOut, uint32_t tilesPerFrame,
                  uint64_t frameBytes, uint32_t nLists, uint32_t* __restrict__ ticket)
{
    __shared__ uint32_t tabL[1u << GC_MF_LSLOT_LOG];
    __shared__ uint32_t tabS[1u << GC_MF_SSLOT_LOG];
    const uint32_t lane = threadIdx.x;
    // Work items = (list, segment) pairs, highest segment first: the segments beyond the first belong to the long lists and are the longest
    // pieces of work (LINK_SEG + LINK_WARM entries), so they start first instead of forming the tail of the launch.  Most of them do not
    // exist: tickets of the upper segments stand for LINK_BATCH items each, which as many lanes test side by side (one trip to memory for the
    // batch instead of one per item; a batch is small enough that the segments found in it are no tail of their own); the first segments, nearly all of which exist, are drawn one by one, so that the work stays evenly dealt.
    const uint32_t nUpper = nLists * (LINK_SEGS - 1u), nBatches = (nUpper + LINK_BATCH - 1u) / LINK_BATCH, nTickets = nBatches + nLists;
    for (;;) {
        uint32_t tk = 0;
        if (lane == 0u) tk = atomicAdd(ticket, 1u);
        tk = __shfl(tk, 0);
        if (tk >= nTickets) break;
        uint64_t todo;                                            // lanes whose item exists
        uint32_t myItem, myStart = 0, myEnd = 0;
        {
            myItem = tk < nBatches ? tk * LINK_BATCH + lane : nUpper + (tk - nBatches);
            const bool inRange = tk < nBatches ? (lane < LINK_BATCH && myItem < nUpper) : lane == 0u;
            bool exists = false;
            if (inRange) {
                const uint32_t seg = LINK_SEGS - 1u - myItem / nLists, fg = myItem % nLists;
                const uint32_t frame = fg >> GC_MF_PART_LOG, g = fg & (GC_MF_PARTS - 1u);
                const uint32_t* row = offs + (uint64_t)frame * (tilesPerFrame + 1u) * GC_MF_PARTS;
                const uint32_t listStart = row[g], listEnd = row[(uint64_t)tilesPerFrame * GC_MF_PARTS + g];
                myStart = listStart + seg * LINK_SEG;
                exists = myStart < listEnd;
                myEnd = (seg + 1u < LINK_SEGS && myStart + LINK_SEG < listEnd) ? myStart + LINK_SEG : listEnd;
            }
            todo = __ballot(exists);
        }
      while (todo != 0ull) {
        const uint32_t src_ = gc_ctz64(todo);
        todo &= todo - 1ull;
        const uint32_t item = __shfl(myItem, (int)src_), start = __shfl(myStart, (int)src_), end = __shfl(myEnd, (int)src_);
        const uint32_t seg = LINK_SEGS - 1u - item / nLists, fg = item % nLists;
        const uint32_t frame = fg >> GC_MF_PART_LOG;
        const GcMfEntry* E = ent + (uint64_t)frame * frameBytes;
        GcMfEntry* EO = entOut + (uint64_t)frame * frameBytes;
        gc_wave_sync();                                           // (the previous item's table operations are done)
        for (uint32_t i = lane; i < (1u << GC_MF_LSLOT_LOG); i += 64u) tabL[i] = 0;
        for (uint32_t i = lane; i < (1u << GC_MF_SSLOT_LOG); i += 64u) tabS[i] = 0;
        gc_wave_sync();
        if (seg != 0u) {                                          // warm the tables: insert only (ds_max keeps the most recent)
            for (uint32_t i = start - LINK_WARM + lane; i < start; i += 64u) {
                const uint64_t e = E[i];
                const uint32_t pos = (uint32_t)e & ((1u << GC_MF_POS_BITS) - 1u);
                const uint32_t kL = (uint32_t)(e >> GC_MF_POS_BITS) & ((1u << GC_MF_KL_BITS) - 1u);
                const uint32_t kS = (uint32_t)(e >> (GC_MF_POS_BITS + GC_MF_KL_BITS)) & ((1u << GC_MF_KS_BITS) - 1u);
                atomicMax(&tabL[kL >> (GC_MF_KL_BITS - GC_MF_LSLOT_LOG)], ((pos + 1u) << GC_MF_TAG_BITS) | (kL & ((1u << GC_MF_TAG_BITS) - 1u)));
                atomicMax(&tabS[kS >> (GC_MF_KS_BITS - GC_MF_SSLOT_LOG)], ((pos + 1u) << GC_MF_TAG_BITS) | (kS & ((1u << GC_MF_TAG_BITS) - 1u)));
            }
            gc_wave_sync();
        }
        // Two register sets of LINK_DEPTH steps each: while one set is linked, the loads of the other are in flight.  (The set is
        // requested right after the previous one has been consumed, so every wait in the loop body is for loads that are one whole
        // set old.)
        uint64_t qa[LINK_DEPTH], qb[LINK_DEPTH];
        mf_link_load(qa, E, start, end, lane);
        for (uint32_t s0 = start; s0 < end; s0 += 2u * 64u * LINK_DEPTH) {
            mf_link_load(qb, E, s0 + 64u * LINK_DEPTH, end, lane);
            mf_link_steps(qa, tabL, tabS, EO, s0, end, lane);
            mf_link_load(qa, E, s0 + 2u * 64u * LINK_DEPTH, end, lane);
            mf_link_steps(qb, tabL, tabS, EO, s0 + 64u * LINK_DEPTH, end, lane);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------ W5 verify
// One workgroup per tile; every thread is independent.  Listed positions are taken in list order (the tile's 256 runs viewed as
// one flat array; the run of a flat index is found by bisection over the run starts in LDS), unlisted ones in position order.
// What the kernel costs is the number of random 16-byte reads (each one is a separate L1 access and L2 request), so:
//   - the position's own bytes come from the staged tile in LDS, never from memory
//   - the short candidate is only read when the long one is missing or did not verify (a verified long candidate has >= 8
//     equal bytes; a different short candidate is a more recent position that matches 5..7 bytes, which loses)
//   - the records are collected in LDS and leave as full lines (stores are written through the L2 at the granularity of the
//     request: 4-byte stores in list order would reach HBM as 94 M partial-line writes)
#define MFV_T GC_MF_VERIFY_T
#ifndef MFV_B
#define MFV_B 4u                      // listed positions per thread and round
#endif
#define MFV_NONE 0xFFFFFFFFu             // sRec: no record written yet (a record's length byte never exceeds GC_MATCH_CAP)
#define MFV_PAD_AFTER (GC_MATCH_CAP + 32u)                        // staged bytes behind the tile: own side of every compare
#define MFV_STAGE_WORDS ((MF_STAGE_PAD + GC_MF_TILE + MFV_PAD_AFTER) / 4u)

__device__ __forceinline__ LzW16 mf_lds_ld16(const uint32_t* sW, uint32_t i)
{
    LzW16 w; w.a = mf_lds_ld64(sW, i); w.b = mf_lds_ld64(sW, i + 8u); return w;
}
// first 16 bytes of candidate c (frame-relative position + 1) against the own window; 0 if shorter than GC_MIN_MATCH
__device__ __forceinline__ uint32_t mfv_len16(const LzW16& me, const LzW16& cw, uint32_t maxLen, uint32_t minLen = GC_MIN_MATCH)
{
    uint32_t len = lz_cmp16(me, cw);
    if (len > maxLen) len = maxLen;
    return len >= minLen ? len : 0u;
}

// Catch-up (BACK): the 16-byte windows start MFV_BACK bytes IN FRONT of the position and of the candidate.  Returns the common prefix of the bytes
// from the position on (0..13; 0 if below minLen or maxLen == 0) and how many of the bytes in front agree as well (0..3, counted backwards).
#define MFV_BACK 3u
__device__ __forceinline__ uint32_t mfv_len13(const LzW16& me, const LzW16& cw, uint32_t maxLen, uint32_t& ext, uint32_t minLen = GC_MIN_MATCH)
{
    const uint64_t d0 = me.a ^ cw.a, d1 = me.b ^ cw.b;
    const bool b2 = ((d0 >> 16) & 0xFFull) == 0ull, b1 = ((d0 >> 8) & 0xFFull) == 0ull, b0 = (d0 & 0xFFull) == 0ull;
    ext = b2 ? (b1 ? (b0 ? 3u : 2u) : 1u) : 0u;
    const uint64_t e0 = (d0 >> 24) | (d1 << 40), e1 = (d1 >> 24) | (1ull << 40);
    uint32_t len = e0 ? gc_ctz64(e0) >> 3 : 8u + (gc_ctz64(e1) >> 3);
    if (len > maxLen) len = maxLen;
    return len >= minLen ? len : 0u;
}

// optional phase profile (thread 0's shader-clock deltas, added to prof[i] as they are taken: no registers held across the kernel)
#define VP_PHASE(prof, tprev, i) do { if ((prof) != nullptr && threadIdx.x == 0) { const unsigned long long now_ = gc_clock(); atomicAdd(&(prof)[i], now_ - (tprev)); (tprev) = now_; } } while (0)

// The verification of ONE tile: fills sRec[0 .. T.len) (LDS) and ends with a workgroup barrier.  Shared by the stand-alone verify
// kernels (records -> HBM) and the fused verify + parse kernel (records never leave the CU).  Called by all MFV_T threads.
// BACK (round 6) = "catch-up": a verified candidate is also compared over the (up to MFV_CATCH = 3) bytes IN FRONT of the position and of the candidate (the 16-byte windows of
// both start three bytes early: no extra memory request -- a second, 4-byte read per candidate cost brotli quality 6 on web-text 14.5 -> 12.1 GB/s, run s2), and the positions
// in front that the match covers as well take it over, lengthened, if that beats their own record by gain -- ZSTD_compressBlock_doubleFast's and the lazy matchers' catch-up
// loop (`while (ip > anchor && match > lowest && ip[-1] == match[-1]) { ip--; match--; mLength++; }`, C/zstd/zstd_double_fast.c:255-262, zstd_lazy.c:1640-1646), which here has
// to be a property of the RECORDS because the parse does not exist yet: the most recent earlier position with the same 5 / 8 bytes is often a short match that the greedy parse
// takes one or two bytes in front of a long one (and the one-step lazy look-ahead only sees one position on).  tools/zstd_parse_lab.c (32 MiB each, estimate, first pass + lazy 1):
// text -4.0 %, web-text -4.6 %, real sources -6.1 %, shared objects -1.3 %, lz-7zip -2.0 %, the Silesia stand-in -1.4 %.  How many bytes in front agree travels in two free bits of the
// record while it is in LDS -- bit 7: lengths end at 64 -- plus one bit of a bitmap of the tile (the wide geometry's distances fill the other 24 bits), and is taken out again
// before the records are read by anything else.
#define MFV_CATCH 3u
#define MFV_XBITS 0x80u
template <int MODE, bool TILE_LIMIT = false, bool BACK = false>
__device__ __forceinline__ void mf_verify_tile(const MfTile& T, const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks,
                    const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent, const uint32_t* recIn,
                    uint32_t* sW, uint32_t* sRec, uint8_t* sExt, uint32_t* sStart, uint32_t* sLocal, uint32_t* sWaveTot,
                    unsigned long long* prof = nullptr, unsigned long long* tprev = nullptr,
                    const uint32_t* __restrict__ changedIn = nullptr /* merging passes: bitmap of the positions whose record a kernel between the passes has changed (W5b) */)
{
    constexpr bool FAR = MODE == MF_FAR || MODE == MF_FAR2 || MODE == MF_SHORT;      // a merging pass
    constexpr bool SHIFT = BACK;                                  // the 16-byte compare windows start MFV_BACK bytes in front of the position and of the candidate (mfv_len13)
    constexpr uint32_t MINLEN = MODE == MF_SHORT ? 3u : GC_MIN_MATCH;
    constexpr uint32_t LONGLEN = MODE == MF_SHORT ? 4u : ((MODE == MF_FAR || MODE == MF_FAR2) ? 16u : 8u);   // a verified long candidate has this many bytes
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t TPF = MF_F(frameBlocks) * GC_MF_TILES_PER_BLOCK;
    uint32_t c = 0, incl = 0;
    if (t < GC_MF_PARTS) {
        const uint32_t* row = offs + ((uint64_t)T.frame * (TPF + 1u) + T.tif) * GC_MF_PARTS;
        const uint32_t o = row[t];
        c = row[GC_MF_PARTS + t] - o;
        sStart[t] = o;
        incl = gc_wave_incl_sum(c);
        if (lane == 63u) sWaveTot[wave] = incl;
    }
    mf_stage(sW, MFV_STAGE_WORDS, src, srcSize, T.tileStart, t, MFV_T);
    uint32_t* const sDirty = (uint32_t*)sExt;                     // (merging passes only)
    uint32_t* const sBack2 = (uint32_t*)(sExt + GC_MF_TILE / 8u + 16u);      // (BACK only) bit q: the match recorded at q also covers two or three bytes in front of it
    if (BACK) { for (uint32_t i = t; i < GC_MF_TILE / 32u; i += MFV_T) sBack2[i] = 0u; }      // (before the barrier below)
    if (FAR) {                                                    // records of the first pass
        const GcU4* R4 = (const GcU4*)(recIn + T.tileStart);
        GcU4* S4 = (GcU4*)sRec;
        for (uint32_t i = t; i < (T.len + 3u) / 4u; i += MFV_T) S4[i] = R4[i];
        const uint32_t* D = changedIn != nullptr ? changedIn + (T.tileStart >> 5) : (const uint32_t*)nullptr;      // (a changed record concerns its own row and the one 64 behind it: two words on)
        const uint32_t nW = (T.len + 31u) / 32u;
        for (uint32_t i = t; i < GC_MF_TILE / 32u + 4u; i += MFV_T) sDirty[i] = D != nullptr ? ((i < nW ? D[i] : 0u) | ((i >= 2u && i - 2u < nW) ? D[i - 2u] : 0u)) : 0u;
    } else {                                                      // "no record yet": what the listed positions leave behind is exactly the unlisted ones
        GcU4 none; none.x = none.y = none.z = none.w = MFV_NONE;
        GcU4* S4 = (GcU4*)sRec;
        for (uint32_t i = t; i < GC_MF_TILE / 4u; i += MFV_T) S4[i] = none;
    }
    __syncthreads();
    if (t < GC_MF_PARTS) {
        uint32_t before = 0;
        for (uint32_t w = 0; w < GC_MF_PARTS / 64u; w++) if (w < wave) before += sWaveTot[w];
        sLocal[t] = before + incl - c;
        if (t == GC_MF_PARTS - 1u) sLocal[GC_MF_PARTS] = before + incl;
    }
    __syncthreads();
    const uint32_t nEnt = sLocal[GC_MF_PARTS];
    VP_PHASE(prof, *tprev, 0);                                     // stage + run offsets
    const uint8_t* wsrc = src + T.frameStart;
    const GcMfEntry* E = ent + (uint64_t)T.frame * ((uint64_t)MF_F(frameBlocks) * GC_ZSTD_BLOCK_MAX);
    const uint64_t blockBase = T.tileStart & ~(uint64_t)(GC_ZSTD_BLOCK_MAX - 1u);     // a tile never straddles blocks
    const uint32_t nBlk = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const uint32_t pTile = (uint32_t)(T.tileStart - blockBase);
    const uint32_t wTile = (uint32_t)(T.tileStart - T.frameStart);
    // unlisted positions: byte runs match at distance 1, the tail of the frame does not match at all
    auto unlisted = [&]() {
        for (uint32_t q = t; q < T.len; q += MFV_T) {
            if (!FAR && sRec[q] != MFV_NONE) continue;              // a listed position: its record is in place (nineteen in twenty on text)
            const uint64_t x = mf_lds_ld64(sW, q + MF_STAGE_PAD);
            const bool run = T.tileStart + q > T.frameStart && ((x << 8) | (uint64_t)mf_lds_byte(sW, q + MF_STAGE_PAD - 1u)) == x;
            const bool windowed = T.tileStart + q + GC_MATCH_CAP + 16u <= T.frameEnd;
            if (windowed && !run) continue;                       // listed
            const uint32_t p = pTile + q;
            uint32_t len = 0;
            if (run && windowed && p + 8u <= nBlk) {              // both sides of the compare lie in the staged tile
                uint32_t maxLen = (nBlk - p) < GC_MATCH_CAP ? (nBlk - p) : GC_MATCH_CAP;
                if (TILE_LIMIT && maxLen > T.len - q) maxLen = T.len - q;
                while (len < maxLen) {
                    const uint32_t more = lz_cmp16(mf_lds_ld16(sW, q + MF_STAGE_PAD + len), mf_lds_ld16(sW, q + MF_STAGE_PAD - 1u + len));
                    len += more;
                    if (more < 16u) break;
                }
                if (len > maxLen) len = maxLen;
                if (len < GC_MIN_MATCH) len = 0;
            }
            sRec[q] = len ? ((1u << 8) | len) : 0u;
        }
    };
    // listed positions, MFV_B per thread and round, stage by stage over small arrays so that the entry loads, then the long
    // candidates, then the short candidates that are still needed are in flight together
    // Every wave takes a contiguous share of the tile's flat list of entries, 64 * MFV_B of them per round.  The run (= partition) of an index
    // is found by walking the run starts from the run of the round's first index (a wave-uniform cursor that only moves forward: a run holds
    // 32 entries on average, so a lane looks at two or three starts) -- not by a bisection per entry, which cost eight dependent LDS reads and
    // two dozen vector instructions of a kernel that is bound by vector issue.
    // 16 bytes at frame-relative position c: from the staged tile where they lie inside it (a candidate a few KiB back is the common case on text, and what this kernel is
    // bound by is the rate of scattered 16-byte requests to the L1: two per listed position), from memory otherwise.  Same bytes either way (round 5).
    const uint32_t stagedLo = wTile >= MF_STAGE_PAD ? wTile - MF_STAGE_PAD : wTile + 0xFFFFFFFFu /* never */, stagedHi = wTile + (MFV_STAGE_WORDS * 4u - MF_STAGE_PAD);
    auto ld16c = [&](uint64_t c) -> LzW16 {
        if (wTile >= MF_STAGE_PAD && c >= stagedLo && c + 20u <= stagedHi) return mf_lds_ld16(sW, (uint32_t)(c - stagedLo));      // (+ 4: the three-word reads of mf_lds_ld64 stay inside the array)
        return lz_ld16(wsrc, c);
    };
    const uint32_t perWave = ((nEnt + MFV_T - 1u) / MFV_T) * 64u;
    const uint32_t jBeg = wave * perWave, jEnd = jBeg + perWave < nEnt ? jBeg + perWave : nEnt;
    uint32_t g0 = 0;                                              // run of index jBeg (largest g with sLocal[g] <= jBeg)
    if (jBeg < jEnd) {
#pragma unroll
        for (uint32_t step = GC_MF_PARTS / 2u; step != 0u; step >>= 1) if (sLocal[g0 + step] <= jBeg) g0 += step;
    }
    for (uint32_t j0 = jBeg; j0 < jEnd; j0 += 64u * MFV_B) {
        uint64_t e[MFV_B];
        bool live[MFV_B];
#pragma unroll
        for (uint32_t k = 0; k < MFV_B; k++) {
            const uint32_t j = j0 + k * 64u + lane;
            live[k] = j < jEnd;
            const uint32_t jj = live[k] ? j : jEnd - 1u;
            uint32_t lo = g0;
            while (sLocal[lo + 1u] <= jj) lo++;                   // (jj < nEnt = sLocal[GC_MF_PARTS]: ends at a run that exists)
            e[k] = E[sStart[lo] + (jj - sLocal[lo])];
            g0 = gc_readlane(lo, 63u);
        }
        uint32_t q[MFV_B], cS[MFV_B], maxLen[MFV_B], bestLen[MFV_B], bestC[MFV_B];
        uint32_t bestExt[MFV_B];
        LzW16 cw[MFV_B];
#pragma unroll
        for (uint32_t k = 0; k < MFV_B; k++) {                    // long candidates
            q[k] = (uint32_t)e[k] & (GC_MF_TILE - 1u);
            const uint32_t cL = (uint32_t)(e[k] >> GC_MF_TILE_LOG) & ((1u << GC_MF_CAND_BITS) - 1u);
            cS[k] = (uint32_t)(e[k] >> (GC_MF_TILE_LOG + GC_MF_CAND_BITS)) & ((1u << GC_MF_CAND_BITS) - 1u);
            const uint32_t p = pTile + q[k];                      // block-relative
            const bool can = live[k] && p + 8u <= nBlk;
            maxLen[k] = can ? ((nBlk - p) < GC_MATCH_CAP ? (nBlk - p) : GC_MATCH_CAP) : 0u;
            if (TILE_LIMIT && maxLen[k] > T.len - q[k]) maxLen[k] = T.len - q[k];      // matches end with their tile (the parse of a tile starts at its first byte)
            if (cS[k] == cL) cS[k] = 0;
            bestC[k] = (can && cL) ? cL : 0u;
            bestExt[k] = 0;
            if (SHIFT) { if (bestC[k] <= MFV_BACK) bestC[k] = 0; if (cS[k] <= MFV_BACK) cS[k] = 0; }      // (a candidate at the very frame start has no bytes in front of it)
            if (bestC[k]) cw[k] = ld16c(bestC[k] - 1u - (SHIFT ? MFV_BACK : 0u));
        }
#pragma unroll
        for (uint32_t k = 0; k < MFV_B; k++) {
            if (SHIFT) { uint32_t x = 0; bestLen[k] = bestC[k] ? mfv_len13(mf_lds_ld16(sW, q[k] + MF_STAGE_PAD - MFV_BACK), cw[k], maxLen[k], x, MINLEN) : 0u; bestExt[k] = (BACK && x > q[k]) ? q[k] : x; }
            else bestLen[k] = bestC[k] ? mfv_len16(mf_lds_ld16(sW, q[k] + MF_STAGE_PAD), cw[k], maxLen[k], MINLEN) : 0u;
            if (bestLen[k] >= (SHIFT && LONGLEN > 13u ? 13u : LONGLEN) || maxLen[k] == 0u) cS[k] = 0;  // verified long candidate: the short one is not needed (shifted windows hold 13 bytes from the position on)
            if (cS[k]) cw[k] = ld16c(cS[k] - 1u - (SHIFT ? MFV_BACK : 0u));
        }
#pragma unroll
        for (uint32_t k = 0; k < MFV_B; k++) {
            const uint32_t pw = wTile + q[k];
            if (cS[k]) {
                uint32_t x = 0;
                const uint32_t len = SHIFT ? mfv_len13(mf_lds_ld16(sW, q[k] + MF_STAGE_PAD - MFV_BACK), cw[k], maxLen[k], x, MINLEN) : mfv_len16(mf_lds_ld16(sW, q[k] + MF_STAGE_PAD), cw[k], maxLen[k], MINLEN);
                if (BACK && x > q[k]) x = q[k];                   // (never beyond the tile's first position: the positions in front belong to another workgroup)
                if (len && (bestLen[k] == 0u || lz_gain(len, pw - (cS[k] - 1u)) > lz_gain(bestLen[k], pw - (bestC[k] - 1u)))) { bestLen[k] = len; bestC[k] = cS[k]; bestExt[k] = x; }
            }
            uint32_t len = bestLen[k];
            while ((SHIFT ? (len & 15u) == 13u : (len >= 16u && (len & 15u) == 0u)) && len < maxLen[k]) {      // saturated: extend 16 bytes per round (own side from LDS)
                const uint32_t more = lz_cmp16(mf_lds_ld16(sW, q[k] + MF_STAGE_PAD + len), ld16c((uint64_t)(bestC[k] - 1u) + len));
                len += more;
                if (len > maxLen[k]) len = maxLen[k];
                if (more < 16u) break;
            }
            if (live[k]) {
                const uint32_t nr = len ? (((pw - (bestC[k] - 1u)) << 8) | len) : 0u;
                if (MODE == MF_BASE && len == GC_MATCH_CAP) sWaveTot[1] = 0xFFFFFFFFu;      // "the tile has a capped record": the word held a count (dead since the run offsets were made), never this value
                const uint32_t xb = (BACK && nr != 0u) ? ((bestExt[k] & 1u) << 7) : 0u;      // how far the match reaches in front of the position (taken out again below)
                const bool x2 = BACK && nr != 0u && (bestExt[k] & 2u) != 0u;
                if (!FAR) { sRec[q[k]] = nr | xb; if (x2) atomicOr(&sBack2[q[k] >> 5], 1u << (q[k] & 31u)); }
                else if (nr) {
                    const uint32_t old = sRec[q[k]];
                    bool take = old == 0u || lz_gain(len, nr >> 8) > lz_gain(old & 0xFFu, old >> 8);
                    if (MODE == MF_FAR2 && old != 0u && len == GC_MATCH_CAP && (old & 0xFFu) == GC_MATCH_CAP && (old >> 8) != (nr >> 8)) {
                        // both records fill the cap: the gain only sees the distances, and the nearer copy is the one that diverges first in a tree of near-copies.  Compare up to
                        // 64 bytes BEHIND the cap at both distances (inside the frame) and keep the one that goes on further (the nearer one if both go on as far).
                        const uint64_t frameLen = T.frameEnd - T.frameStart;
                        const uint64_t room = frameLen - ((uint64_t)pw + GC_MATCH_CAP);           // (listed positions have CAP + 16 bytes of frame behind them)
                        const uint32_t lim = room < 80u ? (room < 16u ? 0u : (uint32_t)room - 16u) : 64u;
                        uint32_t eN = 0, eO = 0;
                        const uint64_t own = (uint64_t)pw + GC_MATCH_CAP;
                        while (eN < lim) { const uint32_t m = lz_cmp16(lz_ld16(wsrc, own + eN), lz_ld16(wsrc, own - (nr >> 8) + eN)); eN += m; if (m < 16u) break; }
                        while (eO < lim) { const uint32_t m = lz_cmp16(lz_ld16(wsrc, own + eO), lz_ld16(wsrc, own - (old >> 8) + eO)); eO += m; if (m < 16u) break; }
                        if (eN > lim) eN = lim; if (eO > lim) eO = lim;
                        take = eN > eO || (eN == eO && (nr >> 8) < (old >> 8));
                    }
                    if (take) {                                   // the continuation below looks again at this row and at the one 64 behind it (nowhere else: what has not changed was decided in the pass before)
                        sRec[q[k]] = nr | xb; if (x2) atomicOr(&sBack2[q[k] >> 5], 1u << (q[k] & 31u));
                        atomicOr(&sDirty[q[k] >> 5], 1u << (q[k] & 31u)); atomicOr(&sDirty[(q[k] + GC_MATCH_CAP) >> 5], 1u << ((q[k] + GC_MATCH_CAP) & 31u));
                    }
                }
            }
        }
    }
    VP_PHASE(prof, *tprev, 1);                                     // listed positions
    if (!FAR) { __syncthreads(); unlisted(); }                     // (the pass reads which records are in place)
    if (BACK) {
        // catch-up: position x takes over the record of x + j (j <= 3) lengthened by j where that record reaches back to x and beats x's own by gain.  Every decision is taken
        // from the records as the passes above left them: a wave walks its share of the tile upwards, 64 positions per step (what a step looks at beyond its own 64 positions the
        // wave has not rewritten yet; the three records beyond the share's end are read before anybody writes), so the result does not depend on the order in which the waves run.
        // The two marker bits go.
        __syncthreads();
        constexpr uint32_t NSUBC = MFV_T / 64u;
        const uint32_t shareLen = GC_MF_TILE / NSUBC, cBeg = wave * shareLen, cEnd = cBeg + shareLen;
        uint32_t beyond[MFV_CATCH];
#pragma unroll
        for (uint32_t j = 0; j < MFV_CATCH; j++) beyond[j] = cEnd + j < T.len ? sRec[cEnd + j] : 0u;
        __syncthreads();
        for (uint32_t x0 = cBeg; x0 < cEnd && x0 < T.len; x0 += 64u) {
            const uint32_t x = x0 + lane;
            const uint32_t old = x < T.len ? sRec[x] : 0u;
            const uint32_t own = old & ~MFV_XBITS;
            uint32_t best = own;
            int bg = (own & 0xFFu) ? lz_gain(own & 0xFFu, own >> 8) : -100000;
#pragma unroll
            for (uint32_t j = 1; j <= MFV_CATCH; j++) {
                const uint32_t y = x + j;
                uint32_t ry = (y < cEnd && y < T.len) ? sRec[y] : 0u;
                if (y >= cEnd) ry = y - cEnd == 0u ? beyond[0] : (y - cEnd == 1u ? beyond[1] : beyond[2]);
                const uint32_t e = (ry & 0x7Fu) == 0u ? 0u : (((ry >> 7) & 1u) | (((sBack2[(y & (GC_MF_TILE - 1u)) >> 5] >> (y & 31u)) & 1u) << 1));      // (ry != 0: y lies inside the tile)
                if (e >= j) {
                    uint32_t l = (ry & 0x7Fu) + j; if (l > GC_MATCH_CAP) l = GC_MATCH_CAP;
                    const uint32_t off = ry >> 8;
                    const int g = lz_gain(l, off);
                    if (g > bg) { bg = g; best = (off << 8) | l; }
                }
            }
            gc_wave_sync();                                       // (every lane has read what it needs of this step's records)
            if (x < T.len && old != best) {
                sRec[x] = best;
                if (own != best) {                                // a record of another match, not just the markers taken out
                    if (FAR) { atomicOr(&sDirty[x >> 5], 1u << (x & 31u)); atomicOr(&sDirty[(x + GC_MATCH_CAP) >> 5], 1u << ((x + GC_MATCH_CAP) & 31u)); }
                    if (MODE == MF_BASE && (best & 0xFFu) == GC_MATCH_CAP) sWaveTot[1] = 0xFFFFFFFFu;
                }
            }
            gc_wave_sync();
        }
    }
    __syncthreads();
    // Continuation of capped matches (round 3).  A record holds at most GC_MATCH_CAP bytes; the parse goes on at the position behind it, whose own
    // candidate -- the most recent earlier position with the same 8 (5, 16, 12) bytes -- is, in data with many copies of the same text (headers,
    // generated code, archives of similar files), usually ANOTHER copy than the one the match came from: a match of 500 bytes was coded as eight
    // sequences with eight offsets (on 8 MiB of C++ headers: 32 087 sequences of exactly 64 bytes against the reference's 414, the stream a third
    // larger).  So the record GC_MATCH_CAP behind a capped one prefers that one's distance when the match goes on there at least as far (within two
    // bytes): the pieces then share their offset and leave as one sequence / one LZMA match (K3a, L1 and W7 merge equal offsets).  A lane walks one
    // residue class mod 64 of its wave's share of the tile upwards, so that the record it looks back at is final; the chain is cut where the shares
    // meet (every 1 KiB), which keeps the result independent of the order in which the waves run.
    // (every pass that merges candidates by gain may have replaced a continued record by a nearer one; the first pass knows whether it wrote a capped record at all --
    //  byte runs aside, which keep their distance 1 anyway -- and most tiles of ordinary text have none: they skip the nine barriers below)
    // (round 5: a merging pass only looks again at rows whose record, or the record 64 in front, CHANGED in this pass -- one bit per position.  What the walk found again and
    //  again in every pass were the rows whose continuation FAILS: the same compares, the same memory round trips as in the pass before.  FLZMA2 level 5 on 211.9 MB of shared
    //  objects spent 17.8 ms in the pass with 4- / 3-byte keys against 4.8 ms on the Silesia stand-in; without the continuation there real sources come out 1.6 % larger.)
    if ((MODE == MF_BASE && sWaveTot[1] == 0xFFFFFFFFu) || MODE == MF_FAR || MODE == MF_FAR2 || MODE == MF_SHORT) {
        constexpr uint32_t NSUB = MFV_T / 64u;
        const uint32_t subLen = ((T.len + NSUB * 64u - 1u) / (NSUB * 64u)) * 64u;
        const uint32_t sBeg = wave * subLen, sEnd = sBeg + subLen < T.len ? sBeg + subLen : T.len;
        // The walk down one residue class mod 64 (a lane's column of the share): row q looks at the record GC_MATCH_CAP in front of it; where that one is capped at a distance d
        // other than q's own, the match at d is measured from q (<= four 16-byte pieces of the source, one round trip) and q's record takes d if that goes on at least as far
        // (within two bytes).  follow: stop at the first row that does not change (phase B).
        // Round 5: a chain of such rows was one memory round trip PER ROW, and on real sources / binaries the pass took as long as the whole entry loop (zstd level 3 on
        // 1 GB: 30.5 / 31.3 GB/s against 44.7 on text; without phase B 36.1 / 35.4, run s4).  When row q is about to be measured over a full 64 bytes the source of row
        // q + 64 at the same distance (the next 64 bytes) is requested in the same round: if q comes out capped at d -- the match goes on -- row q + 64 is decided from
        // what has arrived already.  Two rows per round trip; the records are those of the row-by-row walk (same reads, same tests, same order per column).
        auto piece4 = [&](uint32_t q, uint64_t cpos, uint32_t maxLen, const LzW16& c0, const LzW16& c1, const LzW16& c2, const LzW16& c3) -> uint32_t {
            const bool h1 = 16u < maxLen, h2 = 32u < maxLen, h3 = 48u < maxLen;
            uint32_t more = lz_cmp16(mf_lds_ld16(sW, q + MF_STAGE_PAD), c0);
            bool full = more == 16u;
            if (full && h1) { const uint32_t m = lz_cmp16(mf_lds_ld16(sW, q + MF_STAGE_PAD + 16u), c1); more += m; full = m == 16u; }
            if (full && h2) { const uint32_t m = lz_cmp16(mf_lds_ld16(sW, q + MF_STAGE_PAD + 32u), c2); more += m; full = m == 16u; }
            if (full && h3) { const uint32_t m = lz_cmp16(mf_lds_ld16(sW, q + MF_STAGE_PAD + 48u), c3); more += m; full = m == 16u; }
            (void)cpos;
            return more > maxLen ? maxLen : more;
        };
        auto walk = [&](uint32_t q, const uint32_t qEnd, const bool follow) {
            while (q < qEnd) {
                const uint32_t prev = sRec[q - GC_MATCH_CAP];
                const uint32_t d = prev >> 8, cur = sRec[q];
                const uint32_t p = pTile + q;
                const bool fresh = !FAR || ((sDirty[q >> 5] >> (q & 31u)) & 1u) != 0u;
                const bool need0 = fresh && (prev & 0xFFu) == GC_MATCH_CAP && (cur >> 8) != d && d <= wTile + q      // (a record taken over from an overlapping frame may reach in front of THIS frame: the pass with 4- / 3-byte keys over frames that tile)
                                   && T.tileStart + q + GC_MATCH_CAP + 16u <= T.frameEnd && p + 8u <= nBlk;
                if (!need0) { if (follow) return; q += 64u; continue; }
                uint32_t max0 = (nBlk - p) < GC_MATCH_CAP ? (nBlk - p) : GC_MATCH_CAP;
                if (TILE_LIMIT && max0 > T.len - q) max0 = T.len - q;
                const uint64_t cpos = (uint64_t)(wTile + q) - d;   // frame-relative position of the continued source
                // row q + 64, ahead of time: only a full row, and only if it would be measured at all
                const uint32_t q1 = q + 64u;
                uint32_t cur1 = 0u;
                bool spec = max0 == GC_MATCH_CAP && q1 < qEnd;
                if (spec) {
                    cur1 = sRec[q1];
                    spec = (cur1 >> 8) != d && T.tileStart + q1 + GC_MATCH_CAP + 16u <= T.frameEnd && p + 64u + GC_MATCH_CAP <= nBlk && (!TILE_LIMIT || q1 + GC_MATCH_CAP <= T.len);
                }
                const bool h1 = 16u < max0, h2 = 32u < max0, h3 = 48u < max0;       // (a piece is read only where the match can still go on: offset < maxLen)
                const LzW16 c0 = lz_ld16(wsrc, cpos), c1 = lz_ld16(wsrc, cpos + (h1 ? 16u : 0u)), c2 = lz_ld16(wsrc, cpos + (h2 ? 32u : 0u)), c3 = lz_ld16(wsrc, cpos + (h3 ? 48u : 0u));
                LzW16 e0 = c0, e1 = c0, e2 = c0, e3 = c0;
                if (spec) { e0 = lz_ld16(wsrc, cpos + 64u); e1 = lz_ld16(wsrc, cpos + 80u); e2 = lz_ld16(wsrc, cpos + 96u); e3 = lz_ld16(wsrc, cpos + 112u); }
                const uint32_t len0 = piece4(q, cpos, max0, c0, c1, c2, c3);
                if (len0 < MINLEN || len0 + 2u < (cur & 0xFFu)) { if (follow) return; q += 64u; continue; }
                sRec[q] = (d << 8) | len0;
                if (FAR) atomicOr(&sDirty[q1 >> 5], 1u << (q1 & 31u));     // (the row below has a new record in front of it)
                if (!(spec && len0 == GC_MATCH_CAP)) { q += 64u; continue; }          // (the row below is looked at in the next round, the ordinary way)
                const uint32_t len1 = piece4(q1, cpos + 64u, GC_MATCH_CAP, e0, e1, e2, e3);
                if (len1 < MINLEN || len1 + 2u < (cur1 & 0xFFu)) { if (follow) return; q += 128u; continue; }
                sRec[q1] = (d << 8) | len1;
                if (FAR) atomicOr(&sDirty[(q1 + 64u) >> 5], 1u << ((q1 + 64u) & 31u));
                q += 128u;
            }
        };
        // phase A: every wave inside its share, from the share's second row on (the first row looks back into the share in front)
        walk(sBeg + GC_MATCH_CAP + lane, sEnd, false);
        __syncthreads();
        // phase B: the first rows, share by share in order (what a share looks back at is final), followed down the share while records change
        // Round 5: one share at a time meant seven memory round trips one after the other with seven of the eight waves waiting (real sources: 37 K of the tile's ~150 K
        // cycles, run s7) although a share's first row hardly ever depends on the repair of the share above: that needs a chain that runs through ALL of that share.
        // So every share first repairs its first row (and what follows from it) AT THE SAME TIME, from the record above as phase A left it -- read before anybody writes --
        // and the pass in order that makes it exact only looks again where the record above has changed since (no memory access anywhere else).
        const uint32_t qFirst = sBeg + lane;
        const uint32_t prevA = (wave != 0u && qFirst < sEnd) ? sRec[qFirst - GC_MATCH_CAP] : 0u;
        __syncthreads();
        if (wave != 0u) walk(qFirst, sEnd, true);
        __syncthreads();
        for (uint32_t sIdx = 2; sIdx < NSUB; sIdx++) {             // (share 1 looked at share 0, which phase B never touches)
            if (wave == sIdx && qFirst < sEnd && sRec[qFirst - GC_MATCH_CAP] != prevA) walk(qFirst, sEnd, true);
            __syncthreads();
        }
    }
}

// MF_FAR / MF_SHORT: the candidates come from a later pass; recIn holds the records so far and a position's record is replaced
// only by a candidate of better gain (recIn == rec: in place; a workgroup reads and writes its own tile only).
template <int MODE, bool BACK = true>
__device__ __forceinline__ void mf_verify_body(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                    const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent, const uint32_t* recIn, uint32_t* rec, const uint32_t* __restrict__ changedIn = nullptr)
{
    __shared__ uint32_t sW[MFV_STAGE_WORDS];
    __shared__ uint32_t sRec[GC_MF_TILE];
    __shared__ __attribute__((aligned(16))) uint8_t sExt[2u * (GC_MF_TILE / 8u + 16u)];   // merging passes: one bit per position, "its record or the one 64 in front changed in this pass"; behind it the catch-up's bitmap
    __shared__ uint32_t sStart[GC_MF_PARTS], sLocal[GC_MF_PARTS + 1u];
    __shared__ uint32_t sWaveTot[GC_MF_PARTS / 64u];
    const uint32_t t = threadIdx.x;
    const uint32_t tile = mf_item(blockIdx.x, per);
    if (tile >= nTiles) return;
    const MfTile T = mf_tile(tile, frameBlocks, srcSize);
    if (T.len == 0u || !T.own) return;                            // (overlapping frames: the tiles a frame shares with the one in front have their records from that one)
    mf_verify_tile<MODE, false, BACK>(T, src, srcSize, frameBlocks, offs, ent, recIn, sW, sRec, sExt, sStart, sLocal, sWaveTot, nullptr, nullptr, changedIn);
    // records out: 16 bytes per lane, full lines
    GcU4* R4 = (GcU4*)(rec + T.tileStart);
    const GcU4* S4 = (const GcU4*)sRec;
    for (uint32_t i = t; i < (T.len + 3u) / 4u; i += MFV_T) R4[i] = S4[i];
}
extern "C" __global__ void __launch_bounds__(MFV_T)
MFK(gc_mf_verify_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                    const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent, uint32_t* __restrict__ rec)
{
    mf_verify_body<MF_BASE>(src, srcSize, frameBlocks, nTiles, per, offs, ent, rec, rec);
}
extern "C" __global__ void __launch_bounds__(MFV_T)
MFK(gc_mf_verify_far_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                        const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent, uint32_t* __restrict__ rec)
{
    mf_verify_body<MF_FAR>(src, srcSize, frameBlocks, nTiles, per, offs, ent, rec, rec);
}
extern "C" __global__ void __launch_bounds__(MFV_T)
MFK(gc_mf_verify_far2_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                        const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent, uint32_t* __restrict__ rec)
{
    mf_verify_body<MF_FAR2>(src, srcSize, frameBlocks, nTiles, per, offs, ent, rec, rec);
}
extern "C" __global__ void __launch_bounds__(MFV_T)
MFK(gc_mf_verify_short_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                          const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent, const uint32_t* __restrict__ recIn, uint32_t* __restrict__ recOut,
                          const uint32_t* __restrict__ changedIn)
{
    mf_verify_body<MF_SHORT, false>(src, srcSize, frameBlocks, nTiles, per, offs, ent, recIn, recOut, changedIn);
}
// The same pass WITH the catch-up.  Where the pass runs over frames that tile the input (FLZMA2 5-6) the catch-up bought < 0.01 % for a millisecond per 212 MB (the Silesia
// stand-in, shared objects, text); where it keeps the overlapping frames -- zstd, FLZMA2 7-9: data whose short matches come from far away, lz-7zip -- it is what puts zstd 19 on
// lz-7zip inside the band (1.0193 with, 1.0206 without, 32 MiB on the device).
extern "C" __global__ void __launch_bounds__(MFV_T)
MFK(gc_mf_verify_shortb_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per,
                          const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent, const uint32_t* __restrict__ recIn, uint32_t* __restrict__ recOut,
                          const uint32_t* __restrict__ changedIn)
{
    mf_verify_body<MF_SHORT, true>(src, srcSize, frameBlocks, nTiles, per, offs, ent, recIn, recOut, changedIn);
}

// ------------------------------------------------------------------------------------------------ W5b deepen
// Higher levels search deeper, as the reference does with hash chains / binary trees of growing search depth
// (ZSTD_HcFindBestMatch zstd_lazy.c:667, searchLog in clevels.h; RMF depth in fl2_compress.c:37-104).  Here the chain is implicit:
// W5 left at every position p the offset of its best match, i.e. a link to an earlier position c with the same context; c's own
// record links to a still earlier occurrence, and so on.  One thread per position follows such links and keeps the
// candidate with the best gain.  Reads the records of W5, writes a second record array (other threads still follow the old links).
// How MANY links is decided per tile and per position (round 3; measured on the emulator, 4 MiB per corpus, FLZMA2 level 5, size against two
// links everywhere / links followed):
//   - the links pay where the data has long repeats with many earlier copies: C++ / Python sources -0.9 % at twelve links for starts and two inside
//     matches (7.6 M links; six links everywhere: -0.8 % for 13.3 M), shared objects -0.5 %, the Python library -0.4 %; on data whose matches are
//     short they buy nothing for three times the reads (text -0.18 %, the Silesia stand-in -0.04 %, lz-7zip -0.06 %).  So a tile goes deep
//     (`depth` links) only if at least one position in 16 has a match of >= 32 bytes, and follows `shallow` links (two; none at brotli 5-6) otherwise;
//   - a position INSIDE a match (its record continues the record in front of it) follows `shallow` links: what the deeper ones find there, the
//     match's start has found already, one byte earlier.
#define MFD_T 256u
#define MFD_LONG 32u
extern "C" __global__ void __launch_bounds__(MFD_T)
MFK(gc_mf_deepen_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per, uint32_t depths /* deep | shallow << 8 */,
                    const uint32_t* __restrict__ recIn, uint32_t* __restrict__ recOut,
                    uint32_t* __restrict__ changed /* optional: one bit per position of the input, "the record is another one than recIn's" -- every word of a tile this workgroup owns is written */)
{
    __shared__ uint32_t sLong;
    const uint32_t depth = depths & 0xFFu, shallow = depths >> 8;
    const uint32_t t = threadIdx.x;
    const uint32_t tile = mf_item(blockIdx.x, per);
    if (tile >= nTiles) return;
    const MfTile T = mf_tile(tile, frameBlocks, srcSize);
    if (T.len == 0u || !T.own) return;
    const uint8_t* wsrc = src + T.frameStart;
    const uint32_t* RI = recIn + T.frameStart;                     // frame-relative indexing, like the candidates
    const uint64_t blockBase = T.tileStart & ~(uint64_t)(GC_ZSTD_BLOCK_MAX - 1u);
    const uint32_t nBlk = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const uint32_t pTile = (uint32_t)(T.tileStart - blockBase), wTile = (uint32_t)(T.tileStart - T.frameStart);
    // does the tile have long matches?  (one coalesced pass over its records, which the loop below reads again from the cache)
    if (t == 0u) sLong = 0;
    __syncthreads();
    uint32_t nLong = 0;
    for (uint32_t q = t; q < T.len; q += MFD_T) nLong += (RI[wTile + q] & 0xFFu) >= MFD_LONG ? 1u : 0u;
    nLong = gc_wave_sum(nLong);
    if ((t & 63u) == 0u && nLong) atomicAdd(&sLong, nLong);
    __syncthreads();
    const uint32_t depthStart = sLong * 16u >= T.len ? depth : shallow;
    for (uint32_t q0 = 0; q0 < T.len; q0 += MFD_T) {
        const uint32_t q = q0 + t;
        const bool in = q < T.len;
        const uint32_t pw = wTile + q, p = pTile + q;
        const uint32_t r = in ? RI[pw] : 0u;
        uint32_t bestLen = r & 0xFFu, bestOff = r >> 8;
        // (a capped record cannot be beaten: the links lead to EARLIER positions, i.e. larger offsets at no more than the same length)
        if (bestLen != 0u && bestLen < GC_MATCH_CAP && T.tileStart + q + GC_MATCH_CAP + 16u <= T.frameEnd) {
            const uint32_t rp = pw ? RI[pw - 1u] : 0u;
            const bool inside = (rp >> 8) == bestOff && (rp & 0xFFu) > bestLen;        // the record in front covers this position with the same distance
            const uint32_t nLinks = inside ? shallow : depthStart;
            const uint32_t maxLen = (nBlk - p) < GC_MATCH_CAP ? (nBlk - p) : GC_MATCH_CAP;
            const LzW16 me = lz_ld16(wsrc, pw);
            int bestGain = lz_gain(bestLen, bestOff);
            uint32_t c = pw - bestOff;
            for (uint32_t d = 0; d < nLinks; d++) {
                const uint32_t rc = RI[c];
                if ((rc & 0xFFu) == 0u || (rc >> 8) > c) break;              // (overlapping frames: the record of a position this frame shares with the one in front may reach back beyond this frame's start)
                const uint32_t c2 = c - (rc >> 8);
                uint32_t len = lz_cmp16(me, lz_ld16(wsrc, c2));
                while (len >= 16u && (len & 15u) == 0u && len < maxLen) {
                    const uint32_t more = lz_cmp16(lz_ld16(wsrc, (uint64_t)pw + len), lz_ld16(wsrc, (uint64_t)c2 + len));
                    len += more;
                    if (more < 16u) break;
                }
                if (len > maxLen) len = maxLen;
                if (len >= GC_MIN_MATCH) { const int g = lz_gain(len, pw - c2); if (g > bestGain) { bestGain = g; bestLen = len; bestOff = pw - c2; } }
                c = c2;
            }
        }
        const uint32_t nr = (bestOff << 8) | bestLen;
        if (in) recOut[T.tileStart + q] = nr;
        if (changed != nullptr) {                                  // (a wave's 64 positions are two whole words of the bitmap: no atomics, every word written)
            const uint64_t m = __ballot(in && nr != r);
            if ((t & 63u) == 0u && q < ((T.len + 63u) & ~63u)) { uint32_t* W = changed + ((T.tileStart + q) >> 5); W[0] = (uint32_t)m; W[1] = (uint32_t)(m >> 32); }
        }
    }
}

// ------------------------------------------------------------------------------------------------ W6 parse
// Greedy parse with one-step lazy evaluation: next(p) = p + len if the match at p is taken, else p + 1; the block's sequences
// are the matches on the path from position 0.  The path is found without walking the block serially:
//   phase 1  every 64-position segment computes, for ALL 64 possible entry lanes, where the path leaves it (pointer doubling
//            through ds_bpermute, 6 rounds; a capped match leaves its segment by at most 64, so the exit is a lane of the next
//            segment); a wave composes the exit maps of the 32 segments of a group (2048 positions) into one group map
//   phase 2  one wave chains the 64 group maps from position 0 (v_readlane hops) -> real entry lane of every group
//   phase 3  every group is walked from its real entry, segment by segment (scalar hops), the path becomes two lane masks per
//            segment (match starts, literals); group totals -> prefix sums -> every sequence and literal knows its slot
#define PZ_T       GC_MF_PARSE_T
#define PZ_WAVES   (PZ_T / 64u)
#define PZ_GSEGS   32u                                           // segments per group
#define PZ_GROUPS  (GC_ZSTD_BLOCK_MAX / 64u / PZ_GSEGS)          // 64 groups per full block

struct PzSeg { uint32_t r0; bool take; uint32_t nxt; };
// record, lazy decision and next pointer of position p = seg * 64 + lane (R = records of the block, n = block length).
// lazy: 0 = follow the records as they are (they already are a parse: W7 wrote them); 1 = give the match at p up if the one at p + 1 is clearly better (ZSTD_compressBlock_lazy, zstd_lazy.c:1516; brotli's
// one-step lazy matching, backward_references_inc.h:80-130); 2 = also look at p + 2 (lazy2).  The decision only looks ahead,
// never at decisions made for other positions, so it stays a pure function of the records.
__device__ __forceinline__ PzSeg pz_seg(const uint32_t* __restrict__ R, uint32_t p, uint32_t n, uint32_t lane, uint32_t lazy)
{
    PzSeg s;
    s.r0 = p < n ? R[p] : 0u;
    const uint32_t r1 = p + 1u < n ? R[p + 1u] : 0u;
    const uint32_t len = s.r0 & 0xFFu, l1 = r1 & 0xFFu;
    s.take = len != 0u;
    const int g0 = lz_gain(len, s.r0 >> 8);
    if (lazy >= 1u && s.take && l1 > len && lz_gain(l1, r1 >> 8) > g0 + 4) s.take = false;
    if (lazy >= 2u) {
        const uint32_t r2 = p + 2u < n ? R[p + 2u] : 0u;
        const uint32_t l2 = r2 & 0xFFu;
        if (s.take && l2 > len + 1u && lz_gain(l2, r2 >> 8) > g0 + 8) s.take = false;
    }
    s.nxt = s.take ? lane + len : lane + 1u;                      // >= 64: leaves the segment
    return s;
}
// exit of the segment for every entry lane: position after following nxt until it is >= 64
__device__ __forceinline__ uint32_t pz_exit(uint32_t nxt)
{
    uint32_t cur = nxt;
#pragma unroll
    for (int r = 0; r < 6; r++) { const uint32_t o = __shfl(cur, (int)(cur & 63u)); if (cur < 64u) cur = o; }
    return cur;                                                   // 64 .. 127
}


// ------------------------------------------------------------------------------------------------ W5 + W6 fused: verify + parse
// The levels that parse the first pass's records as they are (no far pass, no link following, no price-based parse) never need the
// records in HBM: a tile is verified into LDS (mf_verify_tile) and parsed right there (rounds 2-3 had one workgroup per BLOCK take the block's
// tiles in order; since round 3 every tile has a workgroup of its own, below).  The parse of a tile is W6's scheme at tile scale -- every wave composes the exit maps of its 16 segments, the maps
// are chained from the lane at which the path entered the tile, every wave walks its segments from its real entry -- and because the
// tiles are taken in order the entry lane, the sequence count and the literal count simply carry over from tile to tile: no group
// maps, no second reading of the records.  Literal bytes come from the staged tile.  The lazy look-ahead stops at the tile's end
// (the records of the next tile do not exist yet): a position in the last two bytes of a tile takes its match as it is.
// HBM traffic per input byte: 8 R (linked entries) + candidate windows + 1 R (tile) + sequences and literals out; W5's 4 W and W6's
// 8-12 R of records are gone.
#define VP_WAVES   (MFV_T / 64u)
#define VP_SEGS    (GC_MF_TILE / 64u)                            // segments per tile
#define VP_SPW     (VP_SEGS / VP_WAVES)                          // segments per wave (16 in both geometries)
#if defined(GC_MF_FAST)
#define VP_MIN_WAVES 6                                           // three workgroups of 512 per CU (LDS allows three): <= 80 VGPRs
#else
#define VP_MIN_WAVES 4
#endif
// ------------------------------------------------------------------------------------------------ W5 + W6 fused, one workgroup per TILE
// The same two stages with the tiles of a block taken by different workgroups AT THE SAME TIME, as W5 takes them: neighbouring tiles run
// on the same XCD at the same moment, so the candidate windows of one are the input the others have just staged (an L2 hit), where the
// block-serial kernel above has every workgroup of an XCD in a block of its own (96 blocks = 12 MiB under a 4 MiB L2: its candidate reads
// all go to HBM, 56 GB per GB of input by the counters).  What made the tiles of a block depend on each other is cut:
//   - a match ends with its tile (TILE_LIMIT): every tile's path starts at its first byte.  The rest of a match that was cut is found
//     again at the next tile's first position; if it comes back with the same offset the sequences kernel merges the two records
//     (chains of records with equal offset and no literals between them are one sequence), so most cuts cost nothing;
//   - where a tile's sequences and literals go in the block's arrays depends on the counts of the tiles in front: every tile publishes
//     its counts in one word (ready << 31 | sequences << 15 | literals) as soon as its path is known, and reads the words of the (at most
//     15) tiles in front of it.  No chain: each word is its tile's own count.  A workgroup draws its tile from a ticket counter of its XCD
//     class (workgroup index mod 8), so the tiles it waits for have been drawn before it and are running or done whatever the dispatch order.
extern "C" __global__ void __launch_bounds__(MFV_T, VP_MIN_WAVES)
MFK(gc_mf_vparse_tile_kernel)(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nTiles, uint32_t per /* a multiple of the tiles of a block */,
                    uint32_t lazy, const uint32_t* __restrict__ offs, const GcMfEntry* __restrict__ ent,
                    GcSeqRaw* __restrict__ seqRaw, uint8_t* __restrict__ lit, GcBlockMeta* __restrict__ meta,
                    uint32_t* __restrict__ tickets /* 8, zeroed */, uint32_t* __restrict__ tileWord /* per tile, zeroed */,
                    unsigned long long* __restrict__ prof)
{
    __shared__ uint32_t sW[MFV_STAGE_WORDS];
    __shared__ uint32_t sRec[GC_MF_TILE];
    __shared__ uint8_t sExt[4u];
    __shared__ uint32_t sAux[GC_MF_TILE / 4u];                    // verify: the tile's run table; parse: take << 7 | next position, one byte per position
    uint32_t* sStart = sAux; uint32_t* sLocal = sAux + GC_MF_PARTS; uint32_t* sWaveTot = sAux + 2u * GC_MF_PARTS + 1u;
    uint8_t* sNxt = (uint8_t*)sAux;
    __shared__ uint8_t  sExitW[VP_WAVES][64];
    __shared__ uint64_t sMaskSeq[VP_SEGS], sMaskLit[VP_SEGS];
    __shared__ uint32_t sCntSeq[VP_WAVES], sCntLit[VP_WAVES];
    __shared__ uint32_t sTicket, sBase[2];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) sTicket = atomicAdd(&tickets[blockIdx.x & (GC_XCDS - 1u)], 1u);
    __syncthreads();
    const uint32_t tile = (blockIdx.x & (GC_XCDS - 1u)) * per + sTicket;
    if (tile >= nTiles) return;
    const MfTile T = mf_tile(tile, frameBlocks, srcSize);
    if (T.len == 0u) return;                                      // (past the end of the input: nobody waits for such a tile)
    unsigned long long tprev = prof ? gc_clock() : 0ull;
    const uint32_t b = tile / GC_MF_TILES_PER_BLOCK, ti = tile % GC_MF_TILES_PER_BLOCK;
    GcSeqRaw* mySeq = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* myLit = lit + (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint64_t lt = gc_lanemask_lt();
    mf_verify_tile<MF_BASE, true>(T, src, srcSize, frameBlocks, offs, ent, nullptr, sW, sRec, sExt, sStart, sLocal, sWaveTot, prof, &tprev);
    const uint32_t n = T.len;
    VP_PHASE(prof, tprev, 2);
    // ---- exit map of this wave's segments
    const uint32_t seg0 = wave * VP_SPW;
    {
        uint32_t comp = lane;
#pragma unroll 1
        for (uint32_t k = 0; k < VP_SPW; k++) {
            const uint32_t seg = seg0 + k;
            if (seg * 64u >= n) break;                            // (uniform) past the end: the identity
            const PzSeg s = pz_seg(sRec, seg * 64u + lane, n, lane, lazy);
            sNxt[seg * 64u + lane] = (uint8_t)(s.nxt | (s.take ? 0x80u : 0u));
            const uint32_t ex = pz_exit(s.nxt) - 64u;
            comp = __shfl(ex, (int)comp);
        }
        sExitW[wave][lane] = (uint8_t)comp;
    }
    __syncthreads();
    VP_PHASE(prof, tprev, 3);
    // ---- real entry lane of this wave: lane 0 of the tile chained through the waves in front
    uint32_t e = 0;
    for (uint32_t w = 0; w < wave; w++) e = sExitW[w][e];
    e = gc_uniform(e);
    VP_PHASE(prof, tprev, 4);
    // ---- walk the segments from the real entry: path masks + counts
    uint32_t nS = 0, nL = 0;
#pragma unroll 1
    for (uint32_t k = 0; k < VP_SPW; k++) {
        const uint32_t seg = seg0 + k;
        if (seg * 64u >= n) break;
        const uint32_t p = seg * 64u + lane;
        const uint32_t nb = sNxt[p], nxt = nb & 0x7Fu;
        uint64_t path = 0;
        uint32_t c = e;
        while (c < 64u) { path |= 1ull << c; c = gc_readlane(nxt, c); }
        e = c - 64u;
        const uint64_t takeMask = __ballot((nb & 0x80u) != 0u), inMask = __ballot(p < n);
        const uint64_t mS = path & takeMask, mL = path & ~takeMask & inMask;
        if (lane == 0) { sMaskSeq[seg] = mS; sMaskLit[seg] = mL; }
        nS += (uint32_t)__popcll(mS); nL += (uint32_t)__popcll(mL);
    }
    if (lane == 0) { sCntSeq[wave] = nS; sCntLit[wave] = nL; }
    __syncthreads();
    uint32_t sBefore = 0, lBefore = 0, sAll = 0, lAll = 0;
    for (uint32_t w = 0; w < VP_WAVES; w++) {
        const uint32_t cs = sCntSeq[w], cl = sCntLit[w];
        if (w < wave) { sBefore += cs; lBefore += cl; }
        sAll += cs; lAll += cl;
    }
    // ---- this tile's counts out, the counts of the tiles in front of it in
    if (wave == 0) {
        if (lane == 0) {
#ifdef HIPEMU
            __atomic_store_n(&tileWord[tile], 0x80000000u | (sAll << 15) | lAll, __ATOMIC_RELEASE);
#else
            __hip_atomic_store(&tileWord[tile], 0x80000000u | (sAll << 15) | lAll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        }
        uint32_t w = 0x80000000u;
        // (bounded: tiles are drawn from a ticket counter, so whatever a tile waits for is running or done -- but if that ever fails to hold the call must end with
        //  an error code, not hang: after ~4 s of polling the lane reports through the word in front of the ticket counters and goes on with empty counts)
        if (lane < ti) { uint32_t spins = 0; do { w = gc_poll_device(&tileWord[tile - ti + lane]); if (!(w >> 31)) { gc_nap(); if (++spins > (1u << 22)) { atomicOr(tickets - 1, 1u); w = 0x80000000u; } } } while (!(w >> 31)); }
        const uint32_t sb = gc_wave_sum(lane < ti ? (w >> 15) & 0xFFFFu : 0u), lb = gc_wave_sum(lane < ti ? w & 0x7FFFu : 0u);      // (a 16 KiB tile of literals only has 2^14 of them: 15 bits)
        if (lane == 0) { sBase[0] = sb; sBase[1] = lb; }
    }
    __syncthreads();
    VP_PHASE(prof, tprev, 5);
    const uint32_t seqBase = sBase[0], litBase = sBase[1];
    // ---- emit
    uint32_t sr = seqBase + sBefore, lr = litBase + lBefore;
#pragma unroll 1
    for (uint32_t k = 0; k < VP_SPW; k++) {
        const uint32_t seg = seg0 + k;
        if (seg * 64u >= n) break;
        const uint64_t mS = sMaskSeq[seg], mL = sMaskLit[seg];
        const uint32_t q = seg * 64u + lane;
        const uint32_t myLitRank = lr + (uint32_t)__popcll(mL & lt);
        if ((mS >> lane) & 1ull) { GcSeqRaw r; r.litRank = myLitRank; r.offml = sRec[q]; mySeq[sr + (uint32_t)__popcll(mS & lt)] = r; }
        if ((mL >> lane) & 1ull) myLit[myLitRank] = (uint8_t)mf_lds_byte(sW, q + MF_STAGE_PAD);
        sr += (uint32_t)__popcll(mS); lr += (uint32_t)__popcll(mL);
    }
    // the block's totals: the last tile of the block that exists
    const bool lastOfBlock = ti + 1u == GC_MF_TILES_PER_BLOCK || T.tileStart + n >= T.frameEnd;
    if (t == 0 && lastOfBlock) { GcBlockMeta m; m.nSeqRaw = seqBase + sAll; m.nLit = litBase + lAll; meta[b] = m; }
    VP_PHASE(prof, tprev, 6);
}

#ifndef GC_MF_FAST       // (W6 works on blocks, not tiles: one copy)
// ------------------------------------------------------------------------------------------------ W6r ring-aware parse (BROTLI qualities 5-6, round 6)
// The reference's hasher tries the LAST DISTANCES first at every position and scores them without their distance bits (C/brotli/enc/hash_longest_match64_inc.h:185-222,
// BackwardReferenceScoreUsingLastDistance / ...PenaltyUsingLastDistance enc/hash.h:113-131), inside the greedy loop with one-step look-ahead of
// enc/backward_references_inc.h:38-140.  On data that repeats with small differences -- machine code, tables of records -- it stays at one distance from copy to copy:
// on ROCm shared objects 45 K of its 120 K commands per 4 MiB reuse the last distance against 25 K of this engine's with W6 + B1's substitution, and those distance
// bits were the whole difference in size (tests/emu -DBRD_STATS).  Which distances are "last" is a property of the PARSE, so this cannot be a finder pass: W6r walks
// the finder's records in order and knows its own ring.  To keep the walk short a block is cut into sub-blocks ("streams", 16 of 8 KiB with 256 threads) that are
// walked at the same time, each by 16 lanes: lane (k, c) of a stream compares chunk c (16 bytes) behind position p with the bytes ring distance k in front of them; the
// four chunks' byte-equality masks side by side (two shuffles) give the length at p and, shifted by one, the length at p + 1 -- one memory round trip per step tests four
// distances over 64 bytes at both positions (a longer copy goes on in the next step: its distance is the ring's first then).  A stream starts 256 positions in front of
// its sub-block and stores nothing there, so that it arrives with the ring of the data in front; it clips its copies at its end (B1 joins pieces of one distance that
// touch).  Where neither p nor p + 1 has a candidate and four positions behind the last copy have been looked at one by one, the walk jumps to the next position with a
// record (16 records are read per step).  Sub-blocks write sequences and literals to their own share of the block's arrays; the
// workgroup closes the gaps at the end.  Scores in W6's units (lz_gain: 4 per byte, 1 per distance bit): last distance 4 len + 4, ring entries 1-3 4 len + 1, the
// position behind wins with 7 more (the reference: + 15 and - 24 ... - 28 on a scale of 135 per byte and 30 per distance bit, cost_diff_lazy 175 = 2 / -1 / 5 in these
// units; emulator, 4 MiB of shared objects: 1.0409 x the reference with those, 1.0351 with 4 / 1 / 7; real sources and lz-7zip do not care).
#define PZR_G0   4                    // score of a copy at the last distance: 4 per byte + this (W6's lz_gain: 4 per byte - 1 per distance bit)
#define PZR_GK   1                    // ... at ring entries 1-3
#define PZR_LAZY 7                    // the position behind wins with this much more
#define PZR_LPS 16u                   // lanes per sub-block: 4 ring distances x 4 chunks of 16 bytes
#define PZR_MAX_STREAMS 16u           // 256 threads
// bit i of the result: byte i of the two 16-byte windows is the same
__device__ __forceinline__ uint32_t pzr_eq16(LzW16 x, LzW16 y)
{
    uint32_t m = 0;
    const uint32_t d[4] = { (uint32_t)(x.a ^ y.a), (uint32_t)((x.a ^ y.a) >> 32), (uint32_t)(x.b ^ y.b), (uint32_t)((x.b ^ y.b) >> 32) };
#pragma unroll
    for (uint32_t i = 0; i < 4u; i++) {
        const uint32_t nzb = (((d[i] & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d[i]) & 0x80808080u;       // bit 7 of every byte that differs
        const uint32_t eq = (~nzb & 0x80808080u) >> 7;                                           // bit 0 of every byte that is the same
        m |= (((eq * 0x00204081u) >> 21) & 0xFu) << (4u * i);                                    // (bits 0 / 8 / 16 / 24 -> bits 21..24 of the product)
    }
    return m;
}
extern "C" __global__ void __launch_bounds__(256)
gc_mf_ringparse_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t nBlocks, uint32_t per, uint32_t lazy, uint32_t ringMin, const uint32_t* __restrict__ rec,
                       GcSeqRaw* __restrict__ seqRaw, uint8_t* __restrict__ lit, GcBlockMeta* __restrict__ meta)
{
    __shared__ uint32_t sCntS[PZR_MAX_STREAMS], sCntL[PZR_MAX_STREAMS];
    const uint32_t t = threadIdx.x, lane = t & 63u, T = blockDim.x, nStreams = T / PZR_LPS;
    const uint32_t b = mf_item(blockIdx.x, per);
    if (b >= nBlocks) return;
    const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t n = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
    const uint32_t* R = rec + base;
    GcSeqRaw* mySeq = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* myLit = lit + (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint8_t* bsrc = src + base;
    const uint32_t st = t / PZR_LPS, j = t % PZR_LPS, k = j >> 2, ch = j & 3u, g0 = lane & ~(PZR_LPS - 1u);
    const uint32_t sub = ((n + nStreams * 64u - 1u) / (nStreams * 64u)) * 64u;
    const uint32_t seqCap = GC_MAX_SEQ_PER_BLOCK / nStreams;
    const uint32_t p0 = st * sub, pEnd = p0 < n ? (n - p0 < sub ? n : p0 + sub) : p0;
    const uint32_t ringLim = (srcSize - base) >= 96u ? (uint32_t)((srcSize - base - 96u) < n ? (srcSize - base - 96u) : n) : 0u;   // (16-byte reads up to 80 bytes behind p)
    GcSeqRaw* sq = mySeq + st * seqCap;
    uint8_t* lq = myLit + p0;
    const uint32_t quietMin = (ringMin >> 16) & 0xFFu, warm = (ringMin >> 24) * 16u, selShift = (ringMin >> 8) & 0xFFu;
    ringMin &= 0xFFu;
    // ---- which blocks: W6 has parsed the block already; only where its sequences come back to a distance of the three in front of them (one in 2^selShift or more: sources,
    //      machine code, tables; text does it once in a thousand) is the block walked again -- elsewhere W6r's result is W6's (measured: web-text, text 0.0 / + 0.1 %)
    if (selShift) {
        __shared__ uint32_t sSel;
        if (t == 0u) sSel = 0u;
        __syncthreads();
        const uint32_t nRaw = meta[b].nSeqRaw, nScan = nRaw < 4096u ? nRaw : 4096u;
        uint32_t cnt = 0;
        for (uint32_t i = 3u + t; i < nScan; i += T) {
            const GcSeqRaw a = mySeq[i], a1 = mySeq[i - 1u], a2 = mySeq[i - 2u], a3 = mySeq[i - 3u];
            const uint32_t o = a.offml >> 8, o1 = a1.offml >> 8;
            if (o == o1 ? a.litRank != a1.litRank : (o == (a2.offml >> 8) || o == (a3.offml >> 8))) cnt++;        // (same distance without literals in between: pieces of one copy)
        }
        cnt = gc_wave_sum(cnt);
        if (lane == 0u && cnt) atomicAdd(&sSel, cnt);
        // ... or where the NEXT copy could use the distance of the one in front (W6 takes whatever the finder lists there, so its parse need not show it): a sample of T
        // records -- the first one at or behind position t n / T, and at the first record behind its end three bytes against the bytes the first one's distance in front
        __shared__ uint32_t sGo, sSeen;
        if (t == 0u) { sGo = 0u; sSeen = 0u; }
        __syncthreads();
        uint32_t go = 0, seen = 0;
        {
            uint32_t q = (uint32_t)(((uint64_t)t * n) / T), r = 0;
            for (uint32_t i = 0; i < 32u && q < n; i++, q++) { r = R[q]; if (r) break; }
            const uint32_t len = r & 0xFFu, off = r >> 8;
            uint32_t e = q + len, r2 = 0;
            if (r != 0u && len < GC_MATCH_CAP) for (uint32_t i = 0; i < 32u && e < n; i++, e++) { r2 = R[e]; if (r2) break; }
            if (r2 != 0u && e + 8u < ringLim) {
                seen = 1u;
                const uint8_t* a = bsrc + e; const uint8_t* c = a - off;
                if (a[0] == c[0] && a[1] == c[1] && a[2] == c[2]) go = 1u;
            }
        }
        go = gc_wave_sum(go); seen = gc_wave_sum(seen);
        if (lane == 0u) { if (go) atomicAdd(&sGo, go); if (seen) atomicAdd(&sSeen, seen); }
        __syncthreads();
#ifdef HIPEMU
        if (t == 0u && getenv("PZR_COUNT")) printf("PZR block %u: %u of %u sequences return to a distance, %u of %u sampled copies go on\n", b, sSel, nScan, sGo, sSeen);
#endif
        if ((sSel << selShift) < nScan && sGo * 16u < sSeen) return;
    }
    // the walk starts `warm` positions in front of the sub-block and stores nothing there: it arrives with the ring (and the copy that is under way) of the data in front
    uint32_t p = p0 < pEnd ? (p0 > warm ? p0 - warm : 0u) : p0, nS = 0, nL = 0, r0 = 0, r1 = 0, r2 = 0, r3 = 0;
    uint32_t quiet = 0;                                           // positions without a candidate since the last copy
    const uint32_t nm1 = n - 1u;
    for (;;) {
        const bool act = p < pEnd;
        if (!__any(act)) break;
        // the records of p .. p + 15 (lane j: p + j) and the bytes of p .. p + 63 (lane j: p + 4 j ..): candidates of this step, and where to go if there are none.
        // ALL loads of the step are issued together from addresses that are valid in every lane (clamped; what a lane may not use is zeroed afterwards): a load behind
        // a per-lane condition is waited for on the spot (hipcc 7.2 put seven round trips into the step that way)
        const bool wide = act && p < ringLim;                                  // (80 bytes behind p can be read)
        const uint32_t dk = k == 0u ? r0 : (k == 1u ? r1 : (k == 2u ? r2 : r3));
        const bool tst = wide && dk != 0u;
        const uint32_t q = p + j;
        uint32_t rw = R[q < nm1 ? q : nm1];
        uint32_t lw = bsrc[p < nm1 ? p : nm1];
        uint32_t eq = 0;                                                       // bit i: byte p + 16 ch + i equals the byte dk in front of it
        if (ringLim) {                                                         // (uniform: the input has 96 bytes behind the block's first position)
            uint32_t lw4; __builtin_memcpy(&lw4, bsrc + (wide ? p + 4u * j : 0u), 4);
            const uint64_t a = base + (tst ? p : 0u) + 16u * ch; const uint32_t dd = tst ? dk : 0u;
            LzW16 xa = lz_ld16(src, a), ya = lz_ld16(src, a - dd);
#ifndef HIPEMU
            asm volatile("" : "+v"(xa.a), "+v"(xa.b), "+v"(ya.a), "+v"(ya.b), "+v"(lw4), "+v"(rw), "+v"(lw));
#endif
            if (wide) lw = lw4;
            if (tst) eq = pzr_eq16(xa, ya);
        }
        if (!act || q >= pEnd) rw = 0u;
        // the four chunks of a distance side by side: 64 bits, from which the length at p and the length at p + 1 are read (the second without loads of its own)
        const uint32_t e1 = __shfl_xor(eq, 1), pair = (ch & 1u) ? (e1 | (eq << 16)) : (eq | (e1 << 16));
        const uint32_t e2 = __shfl_xor(pair, 2);
        const uint64_t M = (ch & 2u) ? ((uint64_t)e2 | ((uint64_t)pair << 32)) : ((uint64_t)pair | ((uint64_t)e2 << 32));
        const uint32_t lenA = ~M ? gc_ctz64(~M) : 64u, lenB = ~(M >> 1) ? gc_ctz64(~(M >> 1)) : 64u;              // (lenB <= 63: M >> 1 has a zero on top)
        const uint32_t lenAB = lenA | (lenB << 8);
        const uint32_t q0 = __shfl(lenAB, (int)g0), q1 = __shfl(lenAB, (int)(g0 + 4u)), q2 = __shfl(lenAB, (int)(g0 + 8u)), q3 = __shfl(lenAB, (int)(g0 + 12u));
        const uint32_t rec0 = __shfl(rw, (int)g0), rec1 = __shfl(rw, (int)(g0 + 1u)), rec2 = __shfl(rw, (int)(g0 + 2u));
        const uint64_t bz = __ballot(rw != 0u);
        const uint32_t nz = (uint32_t)(bz >> g0) & 0xFFFFu;                    // bit i: position p + i has a record
        if (!act) continue;
        // ---- the best candidate at p and at p + 1 (identical in the lanes of the stream)
        const bool em = p >= p0;                                              // (false: still in front of the sub-block)
        const uint32_t room = (em ? pEnd : p0) - p;
        uint32_t bLen = 0, bOff = 0; int bG = -1000;
        uint32_t cLen = 0; int cG = -1000;
        {
            uint32_t l = rec0 & 0xFFu; if (l > room) l = room;
            if (l >= 2u) { bLen = l; bOff = rec0 >> 8; bG = lz_gain(l, bOff); }
            l = rec1 & 0xFFu; if (l + 1u > room) l = room - 1u;
            if (l >= 2u) { cLen = l; cG = lz_gain(l, rec1 >> 8); }
        }
#define PZR_TRY(qq, d, kk) { uint32_t l = (qq) & 0xFFu; if (l > room) l = room; int g = (int)(4u * l) + ((kk) == 0u ? PZR_G0 : PZR_GK); \
                            if ((d) != 0u && l >= ringMin + ((kk) >= 2u ? 1u : 0u) && g > bG) { bLen = l; bOff = (d); bG = g; } \
                            l = (qq) >> 8; if (l + 1u > room) l = room - 1u; g = (int)(4u * l) + ((kk) == 0u ? PZR_G0 : PZR_GK); \
                            if ((d) != 0u && l >= ringMin + ((kk) >= 2u ? 1u : 0u) && g > cG) { cLen = l; cG = g; } }
        PZR_TRY(q0, r0, 0u) PZR_TRY(q1, r1, 1u) PZR_TRY(q2, r2, 2u) PZR_TRY(q3, r3, 3u)
#undef PZR_TRY
        bool take = bLen != 0u && nS < seqCap;
        if (take && lazy >= 1u && cLen != 0u && cG >= bG + PZR_LAZY) take = false;
        if (take && lazy >= 2u) { const uint32_t l2 = rec2 & 0xFFu; if (l2 > bLen + 1u && l2 + 2u <= room && lz_gain(l2, rec2 >> 8) > bG + 8) take = false; }
        if (take) {
            if (em && j == 0u) { GcSeqRaw r; r.litRank = nL; r.offml = (bOff << 8) | bLen; sq[nS] = r; }
            nS += em ? 1u : 0u; p += bLen; quiet = 0;
            if (bOff != r0) { r3 = r2; r2 = r1; r1 = r0; r0 = bOff; }
        } else {
            // literals: one -- or, where neither p nor p + 1 has a candidate and the positions just behind a copy have been looked at, all up to the next position with a
            // record (the ring distances are not tried in between: what ends a copy is a few differing bytes, and the finder lists the continuation behind them itself)
            uint32_t run = 1u;
            if (wide && bLen == 0u && cLen == 0u && quiet >= quietMin) { const uint32_t m = nz & ~3u; run = m ? (uint32_t)__builtin_ctz(m) : PZR_LPS; if (run > room) run = room; }
            if (em && 4u * j < run) {
                lq[nL + 4u * j] = (uint8_t)lw;
                if (4u * j + 1u < run) lq[nL + 4u * j + 1u] = (uint8_t)(lw >> 8);
                if (4u * j + 2u < run) lq[nL + 4u * j + 2u] = (uint8_t)(lw >> 16);
                if (4u * j + 3u < run) lq[nL + 4u * j + 3u] = (uint8_t)(lw >> 24);
            }
            nL += em ? run : 0u; p += run; quiet += run;
        }
    }
    if (j == 0u) { sCntS[st] = nS; sCntL[st] = nL; }
    __syncthreads();
    // ---- close the gaps between the sub-blocks' shares (every move goes towards the front: a share is read T entries at a time, then stored)
    uint32_t preS = sCntS[0], preL = sCntL[0];
    for (uint32_t s = 1; s < nStreams; s++) {
        const uint32_t cS = sCntS[s], cL = sCntL[s];
        const GcSeqRaw* fromS = mySeq + s * seqCap; const uint8_t* fromL = myLit + s * sub;
        for (uint32_t i0 = 0; i0 < cS; i0 += T) {
            GcSeqRaw r; r.litRank = 0; r.offml = 0;
            if (i0 + t < cS) r = fromS[i0 + t];
            __syncthreads();
            if (i0 + t < cS) { r.litRank += preL; mySeq[preS + i0 + t] = r; }
            __syncthreads();
        }
        for (uint32_t i0 = 0; i0 < cL; i0 += T) {
            uint8_t v = 0;
            if (i0 + t < cL) v = fromL[i0 + t];
            __syncthreads();
            if (i0 + t < cL) myLit[preL + i0 + t] = v;
            __syncthreads();
        }
        preS += cS; preL += cL;
    }
    if (t == 0u) { GcBlockMeta m; m.nSeqRaw = preS; m.nLit = preL; meta[b] = m; }
}

extern "C" __global__ void __launch_bounds__(PZ_T)
gc_mf_parse_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t nBlocks, uint32_t per, uint32_t lazy, const uint32_t* __restrict__ rec,
                   GcSeqRaw* __restrict__ seqRaw, uint8_t* __restrict__ lit, GcBlockMeta* __restrict__ meta, uint16_t* __restrict__ priceTab,
                   uint32_t litCtxArg /* price statistics: bits of (previous byte >> 5) that select the literal row; bit 31: the byte in
                                         front of src exists (src is a later part of one buffer), so position 0 has a real context */)
{
    const uint32_t litCtxMask = litCtxArg & 0xFFu, hasPrev = litCtxArg >> 31;
    __shared__ uint32_t sStat[GC_PRICE_WORDS];                    // symbol statistics of this parse (only when priceTab != nullptr)
    __shared__ uint8_t  sGExit[PZ_GROUPS][64];
    __shared__ uint32_t sEntry[PZ_GROUPS];
    __shared__ uint32_t sGSeq[PZ_GROUPS], sGLit[PZ_GROUPS];      // phase 3a: counts; then exclusive prefix
    __shared__ uint64_t sMaskSeq[GC_ZSTD_BLOCK_MAX / 64u], sMaskLit[GC_ZSTD_BLOCK_MAX / 64u];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t b = mf_item(blockIdx.x, per);
    if (b >= nBlocks) return;
    const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t n = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
    const uint32_t nSeg = (n + 63u) >> 6, nGroups = (nSeg + PZ_GSEGS - 1u) / PZ_GSEGS;
    const uint32_t* R = rec + base;
    GcSeqRaw* mySeq = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* myLit = lit + (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint8_t* bsrc = src + base;

    const bool stats = priceTab != nullptr;
    if (stats) for (uint32_t i = t; i < GC_PRICE_WORDS; i += PZ_T) sStat[i] = 0;
    // ---- phase 1: group exit maps
    for (uint32_t g = wave; g < nGroups; g += PZ_WAVES) {
        uint32_t comp = lane;                                     // where the path that enters the group at lane `lane` stands
        const uint32_t sEnd = (g + 1u) * PZ_GSEGS < nSeg ? (g + 1u) * PZ_GSEGS : nSeg;
        for (uint32_t seg = g * PZ_GSEGS; seg < sEnd; seg++) {
            const PzSeg s = pz_seg(R, seg * 64u + lane, n, lane, lazy);
            const uint32_t ex = pz_exit(s.nxt) - 64u;
            comp = __shfl(ex, (int)comp);
        }
        sGExit[g][lane] = (uint8_t)comp;
    }
    __syncthreads();
    // ---- phase 2: chain the groups from position 0
    if (wave == 0) {
        uint32_t x4[PZ_GROUPS / 4u];                              // lane e: exits of entry lane e, four groups per register
#pragma unroll
        for (uint32_t k = 0; k < PZ_GROUPS / 4u; k++) {
            uint32_t v = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4u; i++) if (4u * k + i < nGroups) v |= (uint32_t)sGExit[4u * k + i][lane] << (8u * i);
            x4[k] = v;
        }
        uint32_t c = 0, mine = 0;
#pragma unroll
        for (uint32_t g = 0; g < PZ_GROUPS; g++) {
            if (lane == g) mine = c;
            c = (gc_readlane(x4[g >> 2], c) >> (8u * (g & 3u))) & 0xFFu;
        }
        sEntry[lane] = mine;
    }
    __syncthreads();
    // ---- phase 3a: walk every group from its real entry: path masks + counts
    for (uint32_t g = wave; g < nGroups; g += PZ_WAVES) {
        uint32_t e = gc_uniform(sEntry[g]);
        uint32_t nS = 0, nL = 0;
        const uint32_t sEnd = (g + 1u) * PZ_GSEGS < nSeg ? (g + 1u) * PZ_GSEGS : nSeg;
        for (uint32_t seg = g * PZ_GSEGS; seg < sEnd; seg++) {
            const uint32_t p = seg * 64u + lane;
            const PzSeg s = pz_seg(R, p, n, lane, lazy);
            uint64_t path = 0;
            uint32_t c = e;
            while (c < 64u) { path |= 1ull << c; c = gc_readlane(s.nxt, c); }
            e = c - 64u;
            const uint64_t takeMask = __ballot(s.take), inMask = __ballot(p < n);
            const uint64_t mS = path & takeMask, mL = path & ~takeMask & inMask;
            if (lane == 0) { sMaskSeq[seg] = mS; sMaskLit[seg] = mL; }
            nS += (uint32_t)__popcll(mS); nL += (uint32_t)__popcll(mL);
        }
        if (lane == 0) { sGSeq[g] = nS; sGLit[g] = nL; }
    }
    __syncthreads();
    if (wave == 0) {
        const uint32_t cs = lane < nGroups ? sGSeq[lane] : 0u, cl = lane < nGroups ? sGLit[lane] : 0u;
        const uint32_t is = gc_wave_incl_sum(cs), il = gc_wave_incl_sum(cl);
        sGSeq[lane] = is - cs; sGLit[lane] = il - cl;
        if (lane == 63u) { GcBlockMeta m; m.nSeqRaw = is; m.nLit = il; meta[b] = m; }
    }
    __syncthreads();
    // ---- phase 3b: emit
    const uint64_t lt = gc_lanemask_lt();
    for (uint32_t g = wave; g < nGroups; g += PZ_WAVES) {
        uint32_t seqRun = sGSeq[g], litRun = sGLit[g];
        const uint32_t sEnd = (g + 1u) * PZ_GSEGS < nSeg ? (g + 1u) * PZ_GSEGS : nSeg;
        for (uint32_t seg = g * PZ_GSEGS; seg < sEnd; seg++) {
            const uint64_t mS = sMaskSeq[seg], mL = sMaskLit[seg];
            const uint32_t p = seg * 64u + lane;
            const uint32_t myLitRank = litRun + (uint32_t)__popcll(mL & lt);
            if ((mS >> lane) & 1ull) {
                GcSeqRaw r; r.litRank = myLitRank; r.offml = R[p];
                mySeq[seqRun + (uint32_t)__popcll(mS & lt)] = r;
                if (stats) { atomicAdd(&sStat[GC_PRICE_LEN + (r.offml & 0xFFu)], 1u); atomicAdd(&sStat[GC_PRICE_SLOT + gc_dist_slot((r.offml >> 8) - 1u)], 1u); }
            }
            if ((mL >> lane) & 1ull) {
                const uint32_t byte = bsrc[p];
                myLit[myLitRank] = (uint8_t)byte;
                if (stats) atomicAdd(&sStat[GC_PRICE_LIT + ((((base + p + hasPrev) ? (uint32_t)bsrc[(int64_t)p - 1] >> 5 : 0u) & litCtxMask) << 8) + byte], 1u);
            }
            seqRun += (uint32_t)__popcll(mS); litRun += (uint32_t)__popcll(mL);
        }
    }
    if (!stats) return;
    // ---- statistics -> static prices of this block for the price-based parse W7 (units of 1/16 bit):
    //      literal given the top 3 bits of the byte in front of it, piece length, distance slot, literal / match flag
    __shared__ uint32_t sSum[12];
    __syncthreads();
    if (wave < 10u) {                                             // row sums: 8 literal contexts, lengths, slots
        uint32_t a = 0;
        if (wave < 8u) for (uint32_t i = lane; i < 256u; i += 64u) a += sStat[GC_PRICE_LIT + wave * 256u + i];
        else if (wave == 8u) for (uint32_t i = lane; i < GC_PRICE_NLEN; i += 64u) a += sStat[GC_PRICE_LEN + i];
        else a = sStat[GC_PRICE_SLOT + lane];
        a = gc_wave_sum(a);
        if (lane == 0) sSum[wave] = a;
    }
    __syncthreads();
    uint32_t nLit = 0; for (uint32_t c = 0; c < 8u; c++) nLit += sSum[c];
    const uint32_t nMat = sSum[8];
    uint16_t* T = priceTab + (uint64_t)b * GC_PRICE_WORDS;
    for (uint32_t i = t; i < GC_PRICE_WORDS; i += PZ_T) {
        uint32_t pr;
        if (i < GC_PRICE_LEN) pr = pz_price(10u * sStat[i] + 3u, 10u * sSum[i >> 8] + 768u);
        else if (i < GC_PRICE_LEN + GC_MIN_MATCH) {
            // lengths below the finder's own minimum never occur in the greedy parse; W7 is offered them (W5s, short pass).  Priced as
            // if each were as frequent as the average of the five lengths above them: what they cost once the coder has adapted
            const uint32_t avg = (sStat[GC_PRICE_LEN + 5u] + sStat[GC_PRICE_LEN + 6u] + sStat[GC_PRICE_LEN + 7u] + sStat[GC_PRICE_LEN + 8u] + sStat[GC_PRICE_LEN + 9u]) / 5u;
            pr = pz_price(2u * avg + 1u, 2u * (nMat + 3u * avg) + 63u);
        }
        else if (i < GC_PRICE_SLOT) pr = pz_price(2u * sStat[i] + 1u, 2u * nMat + 63u);
        else if (i < GC_PRICE_FLAGS) pr = pz_price(2u * sStat[i] + 1u, 2u * nMat + 44u);
        else if (i == GC_PRICE_FLAGS) pr = pz_price(nLit + 1u, nLit + nMat + 2u);          // "this symbol is a literal"
        else if (i == GC_PRICE_FLAGS + 1u) { pr = pz_price(nMat + 1u, nLit + nMat + 2u); if (pr > 64u) pr = 64u; }   // "this symbol is a match": at most 4 bits.
                                                                  // The greedy parse only knows matches of >= 5 bytes; where it finds none (16-bit samples,
                                                                  // tables of small records) the flag would be priced at 15 bits and the shortest path would never
                                                                  // try the 2-4 byte matches the reference codes such data with (its adaptive flag settles near 1 bit)
        else pr = 0;
        T[i] = (uint16_t)pr;
    }
}
#endif

}   // namespace
