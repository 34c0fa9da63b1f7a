// gc_zstd_lz.hip -- K1: all-positions-parallel match finder + deterministic greedy/lazy parse.
//
// Replaces, for one <=128 KiB zstd block per workgroup, the serial per-position loop of
// ZSTD_compressBlock_doubleFast_noDict_generic (C/zstd/zstd_double_fast.c:105-323) and the sequence store
// (ZSTD_storeSeq, zstd_compress_internal.h:776).  The CPU loop visits one position at a time and skips the
// inside of matches; a CDNA4 workgroup instead evaluates a chunk of LZ_T consecutive positions at once:
//
//   P0  every lane loads the 8 bytes at its position and hashes them twice: a 5-byte "short" hash and an
//       8-byte "long" hash (the reference's two-table idea, zstd_double_fast.c:132-141; the hash functions
//       themselves are free choices and use 32-bit multiplies, which are full rate on the VALU)
//   P1  probe both LDS tables (entry = position<<15 | 15-bit tag, so hash collisions are rejected from the
//       tag alone without touching memory)
//   P2  insert: ds_max_u32 -> "most recent position wins", identical to sequential insertion order, hence
//       the result does not depend on wave timing (the encoder is deterministic)
//   P3  a small generation-stamped table (ds_min_u32 -> first occurrence inside the chunk) supplies the
//       short-distance candidates that P1 cannot see because the whole chunk was probed before it was inserted
//   P4  candidates are verified against the immutable input: all candidate windows (16 B each) are requested
//       together so only one memory latency is exposed; a saturated best candidate is extended 16 B per round
//       up to GC_MATCH_CAP; longer matches appear as chains of capped matches with equal offset, merged in K3
//   P5  parse: next(t) = t+len if a match is taken at t, else t+1.  Each wave resolves its 64-position
//       segment for EVERY possible entry lane by pointer doubling through ds_bpermute (6 rounds), one lane
//       chains the 16 wave exits, then each wave marks its real path by binary lifting over the same jump tables.
//   P6  emit: wave ballots + popcounts place literals and sequences; no atomics, order = position order.
//
// LDS: 64 KiB long table + 64 KiB short table + 8 KiB chunk table + 8 KiB parse scratch (1 workgroup / CU).
#include "gc_lz_parse.h"

#define LZ_LOG_L    14u              // long-hash table: 2^14 entries
#define LZ_LOG_S    14u              // short-hash table
#define LZ_LOG_C    11u              // chunk-local table
#define LZ_TAG_BITS 15u

extern "C" __global__ void __launch_bounds__(LZ_T)
gc_zstd_lz_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, GcSeqRaw* __restrict__ seqRaw,
                  uint8_t* __restrict__ lit, GcBlockMeta* __restrict__ meta,
                  unsigned long long* __restrict__ prof /* optional: per-phase cycle sums (thread 0 of every block) */)
{
    __shared__ uint32_t tabL[1u << LZ_LOG_L];
    __shared__ uint32_t tabS[1u << LZ_LOG_S];
    __shared__ uint32_t tabC[1u << LZ_LOG_C];
    __shared__ LzParseLds S;

    const uint32_t t = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t n = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
    GcSeqRaw* mySeq = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* myLit = lit + (uint64_t)b * GC_ZSTD_BLOCK_MAX;

    for (uint32_t i = t; i < (1u << LZ_LOG_L); i += LZ_T) tabL[i] = 0;
    for (uint32_t i = t; i < (1u << LZ_LOG_S); i += LZ_T) tabS[i] = 0;
    for (uint32_t i = t; i < (1u << LZ_LOG_C); i += LZ_T) tabC[i] = 0xFFFFFFFFu;
    if (t == 0) S.sCursor = 0;
    __syncthreads();

    LzProf P; P.on = prof != nullptr; for (int i = 0; i < GC_LZ_PHASES; i++) P.pc[i] = 0;
    P.tprev = P.on ? gc_clock() : 0ull;
    uint32_t totalSeq = 0, totalLit = 0;   // uniform running totals
    const uint32_t nChunks = (n + LZ_T - 1) / LZ_T;
    LzW16 own; own.a = 0; own.b = 0;                    // own 16 bytes, always loaded one chunk ahead
    if (base + t + GC_MATCH_CAP + 16u <= srcSize) own = lz_ld16(src, base + t);

    for (uint32_t k = 0; k < nChunks; k++) {
        const uint32_t cbase = k * LZ_T;
        const uint32_t p = cbase + t;
        const bool inBlock = p < n;
        // positions closer than 8 bytes to the block end, or without a readable compare window, stay literals
        const bool canHash = p + 8u <= n && base + p + GC_MATCH_CAP + 16u <= srcSize;

        // ---- P0/P1: load, hash, probe
        uint32_t lo = 0, hi = 0, hL = 0, hS = 0, eL = 0, eS = 0;
        const LzW16 me = own;
        if (canHash) {
            lo = (uint32_t)me.a; hi = (uint32_t)(me.a >> 32);
            hL = lz_hash_long(lo, hi); hS = lz_hash_short(lo, hi);
            eL = tabL[hL >> (32u - LZ_LOG_L)];
            eS = tabS[hS >> (32u - LZ_LOG_S)];
        }
        const uint32_t tagL = (hL >> (32u - LZ_LOG_L - LZ_TAG_BITS)) & ((1u << LZ_TAG_BITS) - 1u);
        const uint32_t tagS = (hS >> (32u - LZ_LOG_S - LZ_TAG_BITS)) & ((1u << LZ_TAG_BITS) - 1u);
        const uint32_t gen = (~k) & 0xFFu;
        const uint32_t tagC = (hS >> 4) & 0x3FFFu;
        const uint32_t slotC = hS >> (32u - LZ_LOG_C);
        __syncthreads();
        LZ_PHASE(P, 0);    // load + hash + probe
        // ---- P2: insert (most recent position wins; first-in-chunk wins for the chunk table)
        if (canHash) {
            atomicMax(&tabL[hL >> (32u - LZ_LOG_L)], (p << LZ_TAG_BITS) | tagL);
            atomicMax(&tabS[hS >> (32u - LZ_LOG_S)], (p << LZ_TAG_BITS) | tagS);
            atomicMin(&tabC[slotC], (gen << 24) | (t << 14) | tagC);
        }
        __syncthreads();
        LZ_PHASE(P, 1);    // insert
        // ---- P3/P4: near probe + verification
        uint32_t bestLen = 0, bestOff = 0;
        if (canHash) {
            const uint32_t maxLen = (n - p) < GC_MATCH_CAP ? (n - p) : GC_MATCH_CAP;
            uint32_t cand[3]; int nc = 0;
            if (eL != 0 && (eL & ((1u << LZ_TAG_BITS) - 1u)) == tagL) cand[nc++] = eL >> LZ_TAG_BITS;
            if (eS != 0 && (eS & ((1u << LZ_TAG_BITS) - 1u)) == tagS) { uint32_t c = eS >> LZ_TAG_BITS; if (nc == 0 || cand[0] != c) cand[nc++] = c; }
            {
                uint32_t eC = tabC[slotC];
                if ((eC >> 24) == gen && (eC & 0x3FFFu) == tagC) {
                    uint32_t tc = (eC >> 14) & 0x3FFu;
                    if (tc < t) { uint32_t c = cbase + tc; bool dup = false; for (int i = 0; i < nc; i++) dup |= cand[i] == c; if (!dup) cand[nc++] = c; }
                }
            }
            lz_verify(src + base, p, me, cand, nc, maxLen, bestLen, bestOff);
        }
        // prefetch the next chunk's own bytes; the latency hides under the parse below
        if (base + p + LZ_T + GC_MATCH_CAP + 16u <= srcSize) own = lz_ld16(src, base + p + LZ_T);
        S.sM[t] = (bestOff << 8) | bestLen;
        __syncthreads();
        LZ_PHASE(P, 2);    // verify
        lz_parse_emit(S, P, cbase, inBlock, bestLen, bestOff, src + base, mySeq, myLit, totalSeq, totalLit);
    }
    if (prof && t == 0) for (int i = 0; i < GC_LZ_PHASES; i++) atomicAdd(&prof[i], P.pc[i]);
    if (t == 0) { GcBlockMeta m; m.nSeqRaw = totalSeq; m.nLit = totalLit; meta[b] = m; }
}
