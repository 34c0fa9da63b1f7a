// gc_lz_window.hip -- the windowed match finder W1..W5 (geometry and overview: gc_mf.h).
//
// Serves the higher levels of all three codecs: matches reach back to the start of their frame (<= 8 MiB) instead of the start
// of their 128 KiB block.  Replaces, per position, the table lookups of ZSTD_compressBlock_doubleFast
// (C/zstd/zstd_double_fast.c:105-323: long + short table, most recent position wins), and stands in for RMF_buildTable /
// RMF_getMatch (C/fast-lzma2/radix_engine.h:920, radix_get.h:60) and brotli's H6 FindLongestMatch
// (C/brotli/enc/hash_longest_match64_inc.h:157-290) as the source of candidates.
//
// Why partition: "most recent earlier position with the same key" is a sequential notion.  Inside one LDS table it can be had
// deterministically with ds_max (K1); for a multi-MiB window the table does not fit LDS, and a table in HBM would be hammered by
// random atomics from all CUs (one 128-byte line per 4-byte update).  Splitting the key space 128 ways by the top hash bits
// turns the problem into 128 independent position-ordered lists per frame and kind, each small enough for one workgroup's LDS
// table; every byte the passes move through HBM is a coalesced run (>= 512 B on average).
//
// Determinism: W3's scatter is stable (ranks from ballots, no atomics on addresses), W4 uses order-independent ds_max / ds_min
// updates between barriers, so the candidate lists -- and with them the compressed bytes -- do not depend on wave timing.
#include "gc_mf.h"
#include "gc_lz_parse.h"

#define MF_WG        256u             // W1 / W3: four waves, one tile each
#define MF_WAVES     (MF_WG / 64u)
#define LINK_T       1024u            // W4: entries per step
#define LINK_LOG     14u              // W4: LDS table slots per partition (x 128 partitions = 2^21 slots per frame and kind)
#define LINK_LOG_C   11u

struct MfKeys { bool ok; uint32_t kL, kS; };

// keys of absolute position P, or ok = false when P is not listed:
//   - no full compare window (GC_MATCH_CAP + 16 bytes) left in the FRAME: such positions never match.  The limit is the frame
//     end, not the input end, so that a frame's bytes do not depend on what follows it (a frame-aligned range shard produces
//     exactly the frames the whole input would)
//   - inside a run of one byte value (the 8 bytes at P equal the 8 bytes at P-1): all those positions share one key and would
//     pile into one partition; W5 gives them the candidate P-1 instead, which is what the table would have returned
__device__ __forceinline__ MfKeys mf_keys(const uint8_t* src, uint64_t srcSize, uint64_t P, uint64_t frameStart, uint64_t frameEnd)
{
    MfKeys r; r.ok = false; r.kL = 0; r.kS = 0;
    (void)srcSize;
    if (P + GC_MATCH_CAP + 16u <= frameEnd) {
        const uint64_t x = gc_ld64(src + P);
        bool run = false;
        if (P > frameStart) run = ((x << 8) | (uint64_t)src[P - 1u]) == x;
        if (!run) {
            const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
            r.ok = true; r.kL = lz_hash_long(lo, hi); r.kS = lz_hash_short(lo, hi);
        }
    }
    return r;
}

// ------------------------------------------------------------------------------------------------ W1 count
extern "C" __global__ void __launch_bounds__(MF_WG)
gc_mf_count_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nFrames, uint32_t* __restrict__ cnt)
{
    __shared__ uint32_t sHist[MF_WAVES][GC_MF_KINDS][GC_MF_PARTS];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t TPF = frameBlocks * GC_MF_TILES_PER_BLOCK;
    const uint32_t tile = blockIdx.x * MF_WAVES + wave;
    if (tile >= nFrames * TPF) return;
    const uint32_t frame = tile / TPF, tif = tile % TPF;
    const uint64_t frameBytes = (uint64_t)frameBlocks * GC_ZSTD_BLOCK_MAX;
    const uint64_t frameStart = (uint64_t)frame * frameBytes;
    const uint64_t frameEnd = (frameStart + frameBytes) < srcSize ? (frameStart + frameBytes) : srcSize;
    const uint64_t tileStart = frameStart + (uint64_t)tif * GC_MF_TILE;
    for (uint32_t i = lane; i < GC_MF_KINDS * GC_MF_PARTS; i += 64u) sHist[wave][i >> GC_MF_PART_LOG][i & (GC_MF_PARTS - 1u)] = 0;
    gc_wave_sync();
    if (tileStart < frameEnd) {
        const uint32_t len = (uint32_t)((frameEnd - tileStart) < GC_MF_TILE ? (frameEnd - tileStart) : GC_MF_TILE);
        for (uint32_t r = 0; r < len; r += 64u) {
            const MfKeys k = mf_keys(src, srcSize, tileStart + r + lane, frameStart, frameEnd);
            if (k.ok) {
                atomicAdd(&sHist[wave][0][k.kL >> (32u - GC_MF_PART_LOG)], 1u);
                atomicAdd(&sHist[wave][1][k.kS >> (32u - GC_MF_PART_LOG)], 1u);
            }
        }
    }
    gc_wave_sync();
    for (uint32_t i = lane; i < GC_MF_KINDS * GC_MF_PARTS; i += 64u) {
        const uint32_t k = i >> GC_MF_PART_LOG, g = i & (GC_MF_PARTS - 1u);
        cnt[(((uint64_t)frame * GC_MF_KINDS + k) * GC_MF_PARTS + g) * TPF + tif] = sHist[wave][k][g];
    }
}

// ------------------------------------------------------------------------------------------------ W2 scan
// one workgroup per (frame, kind): counts -> exclusive offsets in (partition, tile) order, in place
extern "C" __global__ void __launch_bounds__(1024)
gc_mf_scan_kernel(uint32_t* __restrict__ cnt, uint32_t* __restrict__ partStart, uint32_t tilesPerFrame)
{
    __shared__ uint32_t sWave[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, fk = blockIdx.x;
    uint32_t* A = cnt + (uint64_t)fk * GC_MF_PARTS * tilesPerFrame;
    const uint32_t C = (GC_MF_PARTS * tilesPerFrame) / 1024u;      // elements per thread (tilesPerFrame is a multiple of 16)
    uint32_t sum = 0;
    for (uint32_t i = 0; i < C; i++) sum += A[t * C + i];
    const uint32_t incl = gc_wave_incl_sum(sum);
    if (lane == 63u) sWave[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < 16u; w++) if (w < wave) before += sWave[w];
    const uint32_t excl = before + incl - sum;
    uint32_t run = excl;
    for (uint32_t i = 0; i < C; i++) { const uint32_t v = A[t * C + i]; A[t * C + i] = run; run += v; }
    // partition g starts at element g * tilesPerFrame = first element of thread 8g
    if ((t & 7u) == 0u) partStart[fk * (GC_MF_PARTS + 1u) + (t >> 3)] = excl;
    if (t == 1023u) partStart[fk * (GC_MF_PARTS + 1u) + GC_MF_PARTS] = excl + sum;
}

// ------------------------------------------------------------------------------------------------ W3 scatter
extern "C" __global__ void __launch_bounds__(MF_WG)
gc_mf_scatter_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nFrames,
                     const uint32_t* __restrict__ offs, GcMfEntry* __restrict__ ent, uint64_t entStride)
{
    __shared__ uint32_t sRun[MF_WAVES][GC_MF_KINDS][GC_MF_PARTS];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t TPF = frameBlocks * GC_MF_TILES_PER_BLOCK;
    const uint32_t tile = blockIdx.x * MF_WAVES + wave;
    if (tile >= nFrames * TPF) return;
    const uint32_t frame = tile / TPF, tif = tile % TPF;
    const uint64_t frameBytes = (uint64_t)frameBlocks * GC_ZSTD_BLOCK_MAX;
    const uint64_t frameStart = (uint64_t)frame * frameBytes;
    const uint64_t frameEnd = (frameStart + frameBytes) < srcSize ? (frameStart + frameBytes) : srcSize;
    const uint64_t tileStart = frameStart + (uint64_t)tif * GC_MF_TILE;
    if (tileStart >= frameEnd) return;
    for (uint32_t i = lane; i < GC_MF_KINDS * GC_MF_PARTS; i += 64u) {
        const uint32_t k = i >> GC_MF_PART_LOG, g = i & (GC_MF_PARTS - 1u);
        sRun[wave][k][g] = offs[(((uint64_t)frame * GC_MF_KINDS + k) * GC_MF_PARTS + g) * TPF + tif];
    }
    gc_wave_sync();
    const uint32_t len = (uint32_t)((frameEnd - tileStart) < GC_MF_TILE ? (frameEnd - tileStart) : GC_MF_TILE);
    const uint64_t lt = gc_lanemask_lt();
    for (uint32_t r = 0; r < len; r += 64u) {
        const uint64_t P = tileStart + r + lane;
        const MfKeys k = mf_keys(src, srcSize, P, frameStart, frameEnd);
#pragma unroll
        for (uint32_t kind = 0; kind < GC_MF_KINDS; kind++) {
            const uint32_t key = kind ? k.kS : k.kL;
            const uint32_t g = key >> (32u - GC_MF_PART_LOG);
            // lanes of this round with the same partition (position order = lane order): stable rank without atomics
            uint64_t peers = __ballot(k.ok);
#pragma unroll
            for (uint32_t b = 0; b < GC_MF_PART_LOG; b++) {
                const bool bit = ((g >> b) & 1u) != 0u;
                const uint64_t bal = __ballot(k.ok && bit);
                peers &= bit ? bal : ~bal;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & lt);
            const uint32_t base = k.ok ? sRun[wave][kind][g] : 0u;
            gc_wave_sync();
            if (k.ok && rank == 0u) sRun[wave][kind][g] = base + (uint32_t)__popcll(peers);
            gc_wave_sync();
            if (k.ok) {
                GcMfEntry e; e.pos = (uint32_t)(P - frameStart); e.key = key;
                ent[(uint64_t)kind * entStride + frameStart + base + rank] = e;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ W4 link
// One workgroup per (frame, kind, partition) streams the partition's position-ordered list in steps of LINK_T entries:
//   probe   tab[slot] = most recent entry of all EARLIER steps with this slot (entry = (pos+1)<<8 | tag8)
//   insert  ds_max -> most recent position wins, as sequential insertion would leave it
//   in-step tabC (generation-stamped ds_min) = first entry of THIS step with exactly this key; it is nearer than anything in
//           tab, so it is preferred.  (First, not nearest: a free choice, every candidate is verified against the input in W5.)
// The entry's key is then replaced by candidate position + 1 (0 = none).
extern "C" __global__ void __launch_bounds__(LINK_T)
gc_mf_link_kernel(const uint32_t* __restrict__ partStart, GcMfEntry* __restrict__ ent, uint64_t entStride, uint64_t frameBytes)
{
    __shared__ uint32_t tab[1u << LINK_LOG];
    __shared__ uint32_t tabC[1u << LINK_LOG_C];
    __shared__ uint32_t sPos[2][LINK_T];
    const uint32_t t = threadIdx.x;
    const uint32_t fk = blockIdx.x >> GC_MF_PART_LOG, g = blockIdx.x & (GC_MF_PARTS - 1u);
    const uint32_t frame = fk >> 1, kind = fk & 1u;
    const uint32_t start = partStart[fk * (GC_MF_PARTS + 1u) + g], end = partStart[fk * (GC_MF_PARTS + 1u) + g + 1u];
    GcMfEntry* E = ent + (uint64_t)kind * entStride + (uint64_t)frame * frameBytes;
    for (uint32_t i = t; i < (1u << LINK_LOG); i += LINK_T) tab[i] = 0;
    for (uint32_t i = t; i < (1u << LINK_LOG_C); i += LINK_T) tabC[i] = 0xFFFFFFFFu;
    __syncthreads();
    const uint32_t nSteps = (end - start + LINK_T - 1u) / LINK_T;
    GcMfEntry nxt; nxt.pos = 0; nxt.key = 0;
    if (start + t < end) nxt = E[start + t];
    for (uint32_t k = 0; k < nSteps; k++) {
        const uint32_t i = start + k * LINK_T + t;
        const bool valid = i < end;
        const GcMfEntry cur = nxt;
        if (i + LINK_T < end) nxt = E[i + LINK_T];               // next step's entry: its latency hides under this step
        if (k != 0u && (k & 0xFFu) == 0u) {                      // generation stamps wrap every 256 steps
            __syncthreads();
            for (uint32_t j = t; j < (1u << LINK_LOG_C); j += LINK_T) tabC[j] = 0xFFFFFFFFu;
        }
        const uint32_t slot = (cur.key >> 11) & ((1u << LINK_LOG) - 1u), tag = (cur.key >> 3) & 0xFFu;
        const uint32_t slotC = (cur.key >> 14) & ((1u << LINK_LOG_C) - 1u), tagC = cur.key & 0x3FFFu;
        const uint32_t gen = (~k) & 0xFFu;
        const uint32_t old = valid ? tab[slot] : 0u;
        sPos[k & 1u][t] = cur.pos;
        __syncthreads();
        if (valid) {
            atomicMax(&tab[slot], ((cur.pos + 1u) << 8) | tag);
            atomicMin(&tabC[slotC], (gen << 24) | (t << 14) | tagC);
        }
        __syncthreads();
        if (valid) {
            uint32_t cand = 0;
            const uint32_t eC = tabC[slotC], tc = (eC >> 14) & 0x3FFu;
            if ((eC >> 24) == gen && (eC & 0x3FFFu) == tagC && tc < t) cand = sPos[k & 1u][tc] + 1u;
            else if (old != 0u && (old & 0xFFu) == tag) cand = old >> 8;
            E[i].key = cand;
        }
    }
}

// ------------------------------------------------------------------------------------------------ W5 parse
// One workgroup per block.  Per 8 KiB tile: every wave copies 16 of the tile's 256 (kind, partition) runs into LDS in position
// order (sCand[kind][position in tile] = candidate + 1), then eight steps of LZ_T positions go through verify / parse / emit.
__device__ __forceinline__ void
lzw_parse_body(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, const uint32_t* __restrict__ offs,
               const uint32_t* __restrict__ partStart, const GcMfEntry* __restrict__ ent, uint64_t entStride,
               GcSeqRaw* __restrict__ seqRaw, uint8_t* __restrict__ lit, GcBlockMeta* __restrict__ meta,
               unsigned long long* __restrict__ prof)
{
    __shared__ uint32_t sCand[GC_MF_KINDS][GC_MF_TILE];
    __shared__ uint32_t sRunOff[GC_MF_KINDS * GC_MF_PARTS], sRunCnt[GC_MF_KINDS * GC_MF_PARTS];
    __shared__ LzParseLds S;

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t frame = b / frameBlocks, bif = b % frameBlocks;
    const uint32_t TPF = frameBlocks * GC_MF_TILES_PER_BLOCK;
    const uint64_t frameBytes = (uint64_t)frameBlocks * GC_ZSTD_BLOCK_MAX;
    const uint64_t frameStart = (uint64_t)frame * frameBytes;
    const uint64_t frameEnd = (frameStart + frameBytes) < srcSize ? (frameStart + frameBytes) : srcSize;
    const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t n = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
    const uint32_t wbase = bif * GC_ZSTD_BLOCK_MAX;            // block start relative to the frame
    const uint8_t* wsrc = src + frameStart;
    GcSeqRaw* mySeq = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* myLit = lit + (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const GcMfEntry* E0 = ent + frameStart;
    const GcMfEntry* E1 = E0 + entStride;

    if (t == 0) S.sCursor = 0;
    LzProf P; P.on = prof != nullptr; for (int i = 0; i < GC_LZ_PHASES; i++) P.pc[i] = 0;
    P.tprev = P.on ? gc_clock() : 0ull;
    uint32_t totalSeq = 0, totalLit = 0;
    LzW16 own; own.a = 0; own.b = 0;
    uint32_t prevByte = 0x100u;                                  // byte before the own position; 0x100 = none (frame start)
    if (base + t + GC_MATCH_CAP + 16u <= srcSize) own = lz_ld16(src, base + t);
    if (base + t > frameStart) prevByte = src[base + t - 1u];

    const uint32_t nTiles = (n + GC_MF_TILE - 1u) >> GC_MF_TILE_LOG;
    for (uint32_t ti = 0; ti < nTiles; ti++) {
        const uint32_t tif = bif * GC_MF_TILES_PER_BLOCK + ti;
        __syncthreads();                                         // the previous tile's steps are done with sCand
        for (uint32_t i = t; i < GC_MF_KINDS * GC_MF_TILE; i += LZ_T) sCand[i >> GC_MF_TILE_LOG][i & (GC_MF_TILE - 1u)] = 0;
        if (t < GC_MF_KINDS * GC_MF_PARTS) {
            const uint32_t k = t >> GC_MF_PART_LOG, g = t & (GC_MF_PARTS - 1u);
            const uint64_t row = (((uint64_t)frame * GC_MF_KINDS + k) * GC_MF_PARTS + g) * TPF;
            const uint32_t o = offs[row + tif];
            const uint32_t nx = (tif + 1u < TPF) ? offs[row + tif + 1u] : partStart[(frame * GC_MF_KINDS + k) * (GC_MF_PARTS + 1u) + g + 1u];
            sRunOff[t] = o; sRunCnt[t] = nx - o;
        }
        __syncthreads();
        for (uint32_t r = wave; r < GC_MF_KINDS * GC_MF_PARTS; r += LZ_WAVES) {
            const GcMfEntry* E = (r >> GC_MF_PART_LOG) ? E1 : E0;
            const uint32_t o = sRunOff[r], c = sRunCnt[r];
            for (uint32_t i = lane; i < c; i += 64u) {
                const GcMfEntry e = E[o + i];
                sCand[r >> GC_MF_PART_LOG][e.pos & (GC_MF_TILE - 1u)] = e.key;
            }
        }
        __syncthreads();
        LZ_PHASE(P, 0);    // gather
        const uint32_t tileLen = (n - ti * GC_MF_TILE) < GC_MF_TILE ? (n - ti * GC_MF_TILE) : GC_MF_TILE;
        for (uint32_t s = 0; s * LZ_T < tileLen; s++) {
            const uint32_t cbase = ti * GC_MF_TILE + s * LZ_T;
            const uint32_t p = cbase + t;                       // block-relative
            const uint32_t pw = wbase + p;                      // frame-relative
            const bool inBlock = p < n;
            const bool canMatch = p + 8u <= n && base + p + GC_MATCH_CAP + 16u <= frameEnd;
            const LzW16 me = own;
            const uint32_t pb = prevByte;
            uint32_t bestLen = 0, bestOff = 0;
            if (canMatch) {
                const uint32_t maxLen = (n - p) < GC_MATCH_CAP ? (n - p) : GC_MATCH_CAP;
                uint32_t cand[3]; int nc = 0;
                const uint32_t cL = sCand[0][s * LZ_T + t], cS = sCand[1][s * LZ_T + t];
                if (cL != 0u) cand[nc++] = cL - 1u;
                if (cS != 0u && cS != cL) cand[nc++] = cS - 1u;
                if (nc == 0 && pb == (uint32_t)(me.a & 0xFFu)) cand[nc++] = pw - 1u;     // inside a byte run (not listed, see mf_keys)
                lz_verify(wsrc, pw, me, cand, nc, maxLen, bestLen, bestOff);
            }
            if (base + p + LZ_T + GC_MATCH_CAP + 16u <= srcSize) own = lz_ld16(src, base + p + LZ_T);
            if (base + p + LZ_T < srcSize) prevByte = src[base + p + LZ_T - 1u];
            S.sM[t] = (bestOff << 8) | bestLen;
            __syncthreads();
            LZ_PHASE(P, 2);    // verify
            lz_parse_emit(S, P, cbase, inBlock, bestLen, bestOff, src + base, mySeq, myLit, totalSeq, totalLit);
        }
    }
    if (prof && t == 0) for (int i = 0; i < GC_LZ_PHASES; i++) atomicAdd(&prof[i], P.pc[i]);
    if (t == 0) { GcBlockMeta m; m.nSeqRaw = totalSeq; m.nLit = totalLit; meta[b] = m; }
}

// Two builds of the same body: 77 KiB of LDS allow two workgroups per CU, which needs <= 64 VGPRs (a few spills); the host
// picks by block count (few blocks: one workgroup per CU at full register budget is enough to cover the chip).
extern "C" __global__ void __launch_bounds__(LZ_T)
gc_lzw_parse_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, const uint32_t* __restrict__ offs,
                    const uint32_t* __restrict__ partStart, const GcMfEntry* __restrict__ ent, uint64_t entStride,
                    GcSeqRaw* __restrict__ seqRaw, uint8_t* __restrict__ lit, GcBlockMeta* __restrict__ meta,
                    unsigned long long* __restrict__ prof)
{
    lzw_parse_body(src, srcSize, frameBlocks, offs, partStart, ent, entStride, seqRaw, lit, meta, prof);
}
#ifndef HIPEMU
extern "C" __global__ void __launch_bounds__(LZ_T, 8)
gc_lzw_parse_kernel_occ2(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, const uint32_t* __restrict__ offs,
                         const uint32_t* __restrict__ partStart, const GcMfEntry* __restrict__ ent, uint64_t entStride,
                         GcSeqRaw* __restrict__ seqRaw, uint8_t* __restrict__ lit, GcBlockMeta* __restrict__ meta,
                         unsigned long long* __restrict__ prof)
{
    lzw_parse_body(src, srcSize, frameBlocks, offs, partStart, ent, entStride, seqRaw, lit, meta, prof);
}
#endif
