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
#include "gc_common.h"
#include "gc_device.h"

#define LZ_T        1024u            // threads per workgroup = positions per chunk
#define LZ_WAVES    (LZ_T / 64u)
#define LZ_LOG_L    14u              // long-hash table: 2^14 entries
#define LZ_LOG_S    14u              // short-hash table
#define LZ_LOG_C    11u              // chunk-local table
#define LZ_TAG_BITS 15u

__device__ __forceinline__ uint32_t lz_hash_long(uint32_t lo, uint32_t hi)  { return lo * 0x9E3779B1u + hi * 0x85EBCA77u; }
__device__ __forceinline__ uint32_t lz_hash_short(uint32_t lo, uint32_t hi) { return lo * 0x9E3779B1u + (hi & 0xFFu) * 0xC2B2AE3Du; }

// 16 bytes at src[pos..] as two little-endian words.  Callers only load windows that lie inside the input:
// a position takes part in matching only if GC_MATCH_CAP + 16 bytes are readable behind it (the last ~80 bytes
// of the whole input are therefore always literals), and every candidate lies before its position.
struct LzW16 { uint64_t a, b; };
__device__ __forceinline__ LzW16 lz_ld16(const uint8_t* src, uint64_t pos)
{
    LzW16 w; __builtin_memcpy(&w, src + pos, 16); return w;
}
// common prefix length (0..16) of two 16-byte windows
__device__ __forceinline__ uint32_t lz_cmp16(LzW16 x, LzW16 y)
{
    uint64_t d0 = x.a ^ y.a, d1 = x.b ^ y.b;
    if (d0) return gc_ctz64(d0) >> 3;
    if (d1) return 8u + (gc_ctz64(d1) >> 3);
    return 16u;
}

// cost-ish score used to compare candidates and for the lazy check: 4 bits per matched byte minus offset bits
__device__ __forceinline__ int lz_gain(uint32_t len, uint32_t off) { return (int)(len * 4u) - (int)gc_hibit32(off + 1u); }

extern "C" __global__ void __launch_bounds__(LZ_T)
gc_zstd_lz_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, GcSeqRaw* __restrict__ seqRaw,
                  uint8_t* __restrict__ lit, GcBlockMeta* __restrict__ meta,
                  unsigned long long* __restrict__ prof /* optional: per-phase cycle sums (thread 0 of every block) */)
{
    __shared__ uint32_t tabL[1u << LZ_LOG_L];
    __shared__ uint32_t tabS[1u << LZ_LOG_S];
    __shared__ uint32_t tabC[1u << LZ_LOG_C];
    __shared__ uint32_t sM[LZ_T];          // per-position match record (offset<<8 | len)
    __shared__ uint32_t sE[LZ_T];          // per-position exit of its wave segment (chunk-relative)
    __shared__ uint32_t sEntry[LZ_WAVES];  // real entry lane of each wave (64 = wave not entered)
    __shared__ uint32_t sCnt[LZ_WAVES];    // per wave: nSeq<<16 | nLit
    __shared__ uint32_t sCursor;
    __shared__ uint8_t  sMark[LZ_T];       // path marks (P5d)

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t b = blockIdx.x;
    const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t n = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
    GcSeqRaw* mySeq = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* myLit = lit + (uint64_t)b * GC_ZSTD_BLOCK_MAX;

    for (uint32_t i = t; i < (1u << LZ_LOG_L); i += LZ_T) tabL[i] = 0;
    for (uint32_t i = t; i < (1u << LZ_LOG_S); i += LZ_T) tabS[i] = 0;
    for (uint32_t i = t; i < (1u << LZ_LOG_C); i += LZ_T) tabC[i] = 0xFFFFFFFFu;
    if (t == 0) sCursor = 0;
    __syncthreads();

    unsigned long long pc[GC_LZ_PHASES]; for (int i = 0; i < GC_LZ_PHASES; i++) pc[i] = 0;
    unsigned long long tprev = prof ? gc_clock() : 0ull;
#define LZ_PHASE(i) do { if (prof && t == 0) { unsigned long long now_ = gc_clock(); pc[i] += now_ - tprev; tprev = now_; } } while (0)
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
        LZ_PHASE(0);    // load + hash + probe
        // ---- P2: insert (most recent position wins; first-in-chunk wins for the chunk table)
        if (canHash) {
            atomicMax(&tabL[hL >> (32u - LZ_LOG_L)], (p << LZ_TAG_BITS) | tagL);
            atomicMax(&tabS[hS >> (32u - LZ_LOG_S)], (p << LZ_TAG_BITS) | tagS);
            atomicMin(&tabC[slotC], (gen << 24) | (t << 14) | tagC);
        }
        __syncthreads();
        LZ_PHASE(1);    // insert
        // ---- P3/P4: near probe + verification
        uint32_t bestLen = 0, bestOff = 0;
        if (canHash) {
            const uint32_t maxLen = (n - p) < GC_MATCH_CAP ? (n - p) : GC_MATCH_CAP;
            int bestGain = -1000;
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
            // level 1: all candidate windows are requested together (one exposed memory latency), 16 bytes each
            LzW16 cw[3];
            for (int i = 0; i < 3; i++) if (i < nc) cw[i] = lz_ld16(src, base + cand[i]);
            uint32_t bestC = 0;
            for (int i = 0; i < 3; i++) {
                if (i < nc) {
                    uint32_t len = lz_cmp16(me, cw[i]);
                    if (len > maxLen) len = maxLen;
                    if (len >= GC_MIN_MATCH) {
                        int g = lz_gain(len, p - cand[i]);
                        if (g > bestGain) { bestGain = g; bestLen = len; bestOff = p - cand[i]; bestC = cand[i]; }
                    }
                }
            }
            // level 2: only a saturated best candidate is extended, 16 bytes per round, up to GC_MATCH_CAP
            while (bestLen >= 16u && (bestLen & 15u) == 0u && bestLen < maxLen) {
                LzW16 x = lz_ld16(src, base + p + bestLen);
                LzW16 y = lz_ld16(src, base + bestC + bestLen);
                uint32_t more = lz_cmp16(x, y);
                bestLen += more;
                if (bestLen > maxLen) bestLen = maxLen;
                if (more < 16u) break;
            }
        }
        // prefetch the next chunk's own bytes; the latency hides under the parse below
        if (base + p + LZ_T + GC_MATCH_CAP + 16u <= srcSize) own = lz_ld16(src, base + p + LZ_T);
        sM[t] = (bestOff << 8) | bestLen;
        __syncthreads();
        LZ_PHASE(2);    // verify
        // ---- P5a: lazy decision and next pointer
        bool take = bestLen != 0;
        if (take && t + 1u < LZ_T) {
            uint32_t m1 = sM[t + 1u];
            uint32_t l1 = m1 & 0xFFu;
            if (l1 > bestLen && lz_gain(l1, m1 >> 8) > lz_gain(bestLen, bestOff) + 4) take = false;
        }
        const uint32_t wbase = wave * 64u;
        uint32_t cur = (take ? lane + bestLen : lane + 1u);      // wave-relative; >= 64 means "left the wave"
        // ---- P5b: pointer doubling inside the wave: exit reached from every lane
        uint32_t jump[6];                                          // jump[r] = position after 2^r hops
#pragma unroll
        for (int r = 0; r < 6; r++) {
            jump[r] = cur;
            uint32_t o = __shfl(cur, (int)(cur & 63u));
            if (cur < 64u) cur = o;
        }
        sMark[t] = 0;
        sE[t] = wbase + cur;
        __syncthreads();
        LZ_PHASE(3);    // lazy + wave pointer doubling
        // ---- P5c: chain the wave exits from the carried cursor.  Wave 0 pulls the 16 exit tables into registers
        //      (lane l holds the exit for entry lane l of every wave) and hops with v_readlane: 16 short steps.
        if (wave == 0) {
            uint32_t ex[LZ_WAVES];
#pragma unroll
            for (uint32_t w = 0; w < LZ_WAVES; w++) ex[w] = sE[w * 64u + lane];
            const uint32_t cursor = sCursor;                      // absolute position in block (uniform)
            uint32_t c = gc_uniform(cursor > cbase ? cursor - cbase : 0u);   // chunk-relative entry
            uint32_t myEntry = 64u;                               // lane w < 16 keeps wave w's entry
#pragma unroll
            for (uint32_t w = 0; w < LZ_WAVES; w++) {
                if (c < (w + 1u) * 64u) {                         // path enters wave w (c >= w*64 by monotonicity)
                    const uint32_t e = c - w * 64u;
                    if (lane == w) myEntry = e;
                    c = gc_readlane(ex[w], e);
                }
            }
            if (lane < LZ_WAVES) sEntry[lane] = myEntry;
            if (lane == 0) sCursor = cbase + c;
        }
        __syncthreads();
        LZ_PHASE(4);    // exit chain
        // ---- P5d: mark the real path of this wave by binary lifting over the saved jump tables:
        //      after round r every node within 2^(r+1)-1 hops of the entry lane is marked (LDS byte scatter, wave-local)
        const uint32_t entry = gc_uniform(sEntry[wave]);
        bool marked = lane == entry;
#pragma unroll
        for (int r = 0; r < 6; r++) {
            if (marked && jump[r] < 64u) sMark[wbase + jump[r]] = 1;
            gc_wave_sync();
            marked = marked || sMark[t] != 0;
        }
        const uint64_t seqMask = __ballot(marked && take);          // path nodes that start a match
        const uint64_t litMask = __ballot(marked && !take && inBlock);   // all other path nodes are literals
        // ---- P6: emit
        if (lane == 0) sCnt[wave] = ((uint32_t)__popcll(seqMask) << 16) | (uint32_t)__popcll(litMask);
        __syncthreads();
        LZ_PHASE(5);    // path walk
        uint32_t seqBefore = 0, litBefore = 0, seqAll = 0, litAll = 0;
        for (uint32_t w = 0; w < LZ_WAVES; w++) {
            uint32_t c = sCnt[w];
            if (w < wave) { seqBefore += c >> 16; litBefore += c & 0xFFFFu; }
            seqAll += c >> 16; litAll += c & 0xFFFFu;
        }
        const uint64_t lt = gc_lanemask_lt();
        const uint32_t myLitRank = totalLit + litBefore + (uint32_t)__popcll(litMask & lt);
        if ((seqMask >> lane) & 1ull) {
            uint32_t idx = totalSeq + seqBefore + (uint32_t)__popcll(seqMask & lt);
            GcSeqRaw r; r.litRank = myLitRank; r.offml = (bestOff << 8) | bestLen;
            mySeq[idx] = r;
        }
        if ((litMask >> lane) & 1ull) myLit[myLitRank] = src[base + p];
        totalSeq += seqAll; totalLit += litAll;
        LZ_PHASE(6);    // emit
    }
    if (prof && t == 0) for (int i = 0; i < GC_LZ_PHASES; i++) atomicAdd(&prof[i], pc[i]);
    if (t == 0) { GcBlockMeta m; m.nSeqRaw = totalSeq; m.nLit = totalLit; meta[b] = m; }
}
