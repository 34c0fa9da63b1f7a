// gc_lz_parse.h -- the position-parallel verification, parse and emit steps shared by the two match finders:
//   K1 gc_zstd_lz_kernel   (gc_zstd_lz.hip)   block-local finder, candidates from LDS hash tables
//   (the windowed finder W1..W6, gc_lz_window.hip, shares lz_verify and lz_gain; its parse W6 is hierarchical instead)
// Both evaluate LZ_T consecutive positions per step; what differs is only where the candidates come from and how
// far back they may lie.  See gc_zstd_lz.hip for the description of the phases (P4 verify, P5 parse, P6 emit).
#pragma once
#include "gc_common.h"
#include "gc_device.h"

#define LZ_T        1024u            // threads per workgroup = positions per step
#define LZ_WAVES    (LZ_T / 64u)

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

#include "gc_mf.h"
// price = 16 * log2(den / num), clamped to [1, GC_PRICE_MAX]: integer arithmetic only, so that the emulator build and the GPU
// produce the same tables (and with them the same parse, byte for byte)
__device__ __forceinline__ uint32_t pz_log2_q8(uint32_t x)       // 256 * log2(x), x >= 1; error < 0.01 bit
{
    const uint32_t e = gc_hibit32(x);
    const uint32_t f = ((x << (31u - e)) >> 15) & 0xFFFFu;         // mantissa - 1 in Q16
    const uint32_t t = (f * (65536u - f)) >> 16;
    const uint32_t frac = f + ((t * 22713u) >> 16);                // log2(1 + f) ~ f + 0.3466 f (1 - f)
    return (e << 8) + (frac >> 8);
}
__device__ __forceinline__ uint32_t pz_price(uint32_t num, uint32_t den)
{
    const uint32_t a = pz_log2_q8(den), b = pz_log2_q8(num);
    uint32_t pr = a > b ? (a - b + 8u) >> 4 : 0u;
    if (pr < 1u) pr = 1u;
    return pr > GC_PRICE_MAX ? GC_PRICE_MAX : pr;
}

// optional in-kernel phase profile (thread 0's shader-clock deltas)
struct LzProf {
    unsigned long long pc[GC_LZ_PHASES];
    unsigned long long tprev;
    bool on;
};
#define LZ_PHASE(P, i) do { if ((P).on && threadIdx.x == 0) { unsigned long long now_ = gc_clock(); (P).pc[i] += now_ - (P).tprev; (P).tprev = now_; } } while (0)

// LDS scratch of the parse
struct LzParseLds {
    uint32_t sM[LZ_T];          // per-position match record (offset<<8 | len)
    uint32_t sE[LZ_T];          // per-position exit of its wave segment (step-relative)
    uint32_t sEntry[LZ_WAVES];  // real entry lane of each wave (64 = wave not entered)
    uint32_t sCnt[LZ_WAVES];    // per wave: nSeq<<16 | nLit
    uint32_t sCursor;           // block-relative position where the parse continues
    uint8_t  sMark[LZ_T];       // path marks (P5d)
};

// ---- P4: verification of up to three candidates against the immutable input.
//   wsrc     start of the search window (block base for K1, frame base for W5)
//   pw       own position relative to wsrc;  cand[i] < pw, relative to wsrc
//   level 1: all candidate windows (16 B each) are requested together so that only one memory latency is exposed;
//   level 2: only a saturated best candidate is extended, 16 bytes per round, up to maxLen (<= GC_MATCH_CAP)
__device__ __forceinline__ void lz_verify(const uint8_t* wsrc, uint32_t pw, const LzW16& me, const uint32_t cand[3], int nc,
                                          uint32_t maxLen, uint32_t& bestLen, uint32_t& bestOff)
{
    int bestGain = -1000;
    LzW16 cw[3];
    for (int i = 0; i < 3; i++) if (i < nc) cw[i] = lz_ld16(wsrc, cand[i]);
    uint32_t bestC = 0;
    for (int i = 0; i < 3; i++) {
        if (i < nc) {
            uint32_t len = lz_cmp16(me, cw[i]);
            if (len > maxLen) len = maxLen;
            if (len >= GC_MIN_MATCH) {
                int g = lz_gain(len, pw - cand[i]);
                if (g > bestGain) { bestGain = g; bestLen = len; bestOff = pw - cand[i]; bestC = cand[i]; }
            }
        }
    }
    while (bestLen >= 16u && (bestLen & 15u) == 0u && bestLen < maxLen) {
        LzW16 x = lz_ld16(wsrc, (uint64_t)pw + bestLen);
        LzW16 y = lz_ld16(wsrc, (uint64_t)bestC + bestLen);
        uint32_t more = lz_cmp16(x, y);
        bestLen += more;
        if (bestLen > maxLen) bestLen = maxLen;
        if (more < 16u) break;
    }
}

// ---- P5/P6: parse and emit one step of LZ_T positions.  Call with every thread of the workgroup; sM[t] must already hold
//      this thread's match record and a workgroup barrier must separate that store from this call.
//   cbase    block-relative position of the step's first position;  p = cbase + t
//   bsrc     start of the block in the input (literal bytes are copied from it)
__device__ __forceinline__ void lz_parse_emit(LzParseLds& S, LzProf& prof, uint32_t cbase, bool inBlock, uint32_t bestLen, uint32_t bestOff,
                                              const uint8_t* bsrc, GcSeqRaw* mySeq, uint8_t* myLit, uint32_t& totalSeq, uint32_t& totalLit)
{
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t p = cbase + t;
    // ---- P5a: lazy decision and next pointer
    bool take = bestLen != 0;
    if (take && t + 1u < LZ_T) {
        uint32_t m1 = S.sM[t + 1u];
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
    S.sMark[t] = 0;
    S.sE[t] = wbase + cur;
    __syncthreads();
    LZ_PHASE(prof, 3);    // lazy + wave pointer doubling
    // ---- P5c: chain the wave exits from the carried cursor.  Wave 0 pulls the 16 exit tables into registers
    //      (lane l holds the exit for entry lane l of every wave) and hops with v_readlane: 16 short steps.
    if (wave == 0) {
        uint32_t ex[LZ_WAVES];
#pragma unroll
        for (uint32_t w = 0; w < LZ_WAVES; w++) ex[w] = S.sE[w * 64u + lane];
        const uint32_t cursor = S.sCursor;                    // absolute position in block (uniform)
        uint32_t c = gc_uniform(cursor > cbase ? cursor - cbase : 0u);   // step-relative entry
        uint32_t myEntry = 64u;                               // lane w < 16 keeps wave w's entry
#pragma unroll
        for (uint32_t w = 0; w < LZ_WAVES; w++) {
            if (c < (w + 1u) * 64u) {                         // path enters wave w (c >= w*64 by monotonicity)
                const uint32_t e = c - w * 64u;
                if (lane == w) myEntry = e;
                c = gc_readlane(ex[w], e);
            }
        }
        if (lane < LZ_WAVES) S.sEntry[lane] = myEntry;
        if (lane == 0) S.sCursor = cbase + c;
    }
    __syncthreads();
    LZ_PHASE(prof, 4);    // exit chain
    // ---- P5d: mark the real path of this wave by binary lifting over the saved jump tables:
    //      after round r every node within 2^(r+1)-1 hops of the entry lane is marked (LDS byte scatter, wave-local)
    const uint32_t entry = gc_uniform(S.sEntry[wave]);
    bool marked = lane == entry;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        if (marked && jump[r] < 64u) S.sMark[wbase + jump[r]] = 1;
        gc_wave_sync();
        marked = marked || S.sMark[t] != 0;
    }
    const uint64_t seqMask = __ballot(marked && take);          // path nodes that start a match
    const uint64_t litMask = __ballot(marked && !take && inBlock);   // all other path nodes are literals
    // ---- P6: emit
    if (lane == 0) S.sCnt[wave] = ((uint32_t)__popcll(seqMask) << 16) | (uint32_t)__popcll(litMask);
    __syncthreads();
    LZ_PHASE(prof, 5);    // path walk
    uint32_t seqBefore = 0, litBefore = 0, seqAll = 0, litAll = 0;
    for (uint32_t w = 0; w < LZ_WAVES; w++) {
        uint32_t c = S.sCnt[w];
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
    if ((litMask >> lane) & 1ull) myLit[myLitRank] = bsrc[p];
    totalSeq += seqAll; totalLit += litAll;
    LZ_PHASE(prof, 6);    // emit
}
