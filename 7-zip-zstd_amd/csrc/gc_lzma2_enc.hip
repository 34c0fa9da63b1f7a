// gc_lzma2_enc.hip -- FLZMA2 path: LZMA2 encoder for the matches found by the match finders (gc_zstd_lz.hip, gc_lz_window.hip).
//
// Replaces, for the 7-Zip method id 0x21 ("FLZMA2", CPP/7zip/Compress/FastLzma2Register.cpp:13-18), the slice encoder of
// Fast-LZMA2: LZMA2_encode (C/fast-lzma2/lzma2_enc.c:1937-2099), LZMA_encodeChunkFast (:579), the length / distance /
// literal coders (:382-441, :116-122) and the range coder (C/fast-lzma2/range_enc.h:62-152, range_enc.c:123-211).
// What is format-normative is what the stock decoder re-derives (C/LzmaDec.c, C/Lzma2Dec.c:97): the symbol grammar
// (isMatch / isRep / isRepG0.. / isRep0Long, length coder choice+low/mid/high trees, 6-bit position slot selected by
// min(len-2,3), reverse bit trees for slots 4..13, direct bits + 4 reverse "align" bits above), the state machine
// (12 states, 7 literal states), matched-literal coding after a match, 11-bit probabilities with shift-5 adaptation, the
// 32-bit range coder with carry propagation, and the LZMA2 chunk header.  The parse, the chunk size and where the coder
// state is reset are free choices.
//
// GPU structure (units: gc_lzma2.h).  What is serial in LZMA is (a) the adaptive probabilities, per probability, and (b) the
// range recurrence, per range-coder run.  The LZMA2 container restarts (b) at every chunk and lets the encoder choose where
// (a) restarts, the device Fast-LZMA2 itself uses to run slices on several threads (lzma2_enc.h:22, lzma2_enc.c:2040-2075):
//   L1 gc_lzma2_prep_kernel  one workgroup per 128 KiB block: the finder's capped records -> position-ordered item list
//                            (matches cut at 4 KiB boundaries and at length 273, literal runs cut every 16 bytes)
//   L2 gc_lzma2_model_kernel one WAVE per model segment.  Per 64 items the lanes derive everything that is local -- coder
//                            state (12-state machine), repeat distances (scans), length / slot / tree nodes, literal nodes
//                            incl. matched literals -- and write the coded bits as events (probability index, bit) into an
//                            LDS buffer in coding order.  The events are then applied 64 at a time: a returning ds_add on a
//                            ticket byte per probability gives every lane its rank among the lower lanes that touch the same
//                            probability (the LDS unit serves them in lane order), and the ranks are played in rounds, so each
//                            event sees exactly the probability sequential coding would see.  Output: one 16-bit word per
//                            coded bit (probability before the update + the bit) in HBM.
//   L3 gc_lzma2_rc_kernel    one LANE per 4 KiB rc chunk: bound = (range >> 11) * p, renormalise, carry -- register-only
//                            work; output bytes are collected eight at a time
//   L4/L5 plan + emit        chunk headers and concatenation (gc_lzma2_frame.hip)
#include "gc_common.h"
#include "gc_device.h"
#include "gc_lzma2.h"
#include "gc_mf.h"
#include "gc_lz_parse.h"      // pz_price
#ifdef HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif

#define LZP_T 256u

// ---------------------------------------------------------------------------------------------- L1: item list
__device__ __forceinline__ uint32_t lzp_excl_scan(uint32_t v, uint32_t* sWave, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t incl = gc_wave_incl_sum(v);
    if (lane == 63u) sWave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < LZP_T / 64u; w++) { const uint32_t c = sWave[w]; if (w < wave) before += c; all += c; }
    __syncthreads();
    *total = all;
    return before + incl - v;
}
// a match [pos, pos + ml) as it is coded: a piece of one byte at either end (left over by a 4 KiB boundary) is given up
__device__ __forceinline__ void lzp_trim(uint32_t pos, uint32_t ml, uint32_t& mStart, uint32_t& mEnd)
{
    mStart = pos; mEnd = pos + ml;
    if (ml == 0u) return;
    const uint32_t b1 = (mStart | (GC_LZMA_RC_SIZE - 1u)) + 1u;            // first boundary behind the start
    if (b1 - mStart == 1u && mEnd > b1) mStart += 1u;
    const uint32_t b2 = (mEnd - 1u) & ~(GC_LZMA_RC_SIZE - 1u);             // last boundary in front of the end
    if (mEnd - b2 == 1u && mStart < b2) mEnd -= 1u;
}
// items of one head: cuts of the literal run [a, mStart) then the pieces of the match [mStart, mEnd); out == nullptr: count only
__device__ __forceinline__ uint32_t lzp_items(uint32_t a, uint32_t mStart, uint32_t mEnd, uint32_t off, uint32_t prevOff, bool isTail, uint64_t* out)
{
    uint32_t n = 0;
    if (a < mStart) {
        for (uint32_t m = (a / GC_LZMA_LIT_CUT + 1u) * GC_LZMA_LIT_CUT; m < mStart; m += GC_LZMA_LIT_CUT) { if (out) out[n] = (uint64_t)m | ((uint64_t)prevOff << 34); n++; }
        if (isTail || (mStart & (GC_LZMA_RC_SIZE - 1u)) == 0u) { if (out) out[n] = (uint64_t)mStart | ((uint64_t)prevOff << 34); n++; }
    }
    uint32_t s = mStart;
    while (s < mEnd) {
        uint32_t e = (s | (GC_LZMA_RC_SIZE - 1u)) + 1u; if (e > mEnd) e = mEnd;       // piece inside one rc chunk
        uint32_t rem = e - s;
        while (rem) {
            uint32_t take = rem < 273u ? rem : 273u;
            if (rem - take == 1u) take--;
            if (out) out[n] = (uint64_t)s | ((uint64_t)take << 18) | ((uint64_t)off << 34);
            n++; s += take; rem -= take;
        }
    }
    return n;
}

extern "C" __global__ void __launch_bounds__(LZP_T)
gc_lzma2_prep_kernel(const GcSeqRaw* __restrict__ seqRaw, const GcBlockMeta* __restrict__ meta, uint64_t srcSize,
                     uint64_t* __restrict__ M /* GC_LZMA_MAX_ITEMS per block */, uint32_t* __restrict__ nM)
{
    __shared__ uint32_t sWave[LZP_T / 64u];
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const uint32_t nRaw = meta[b].nSeqRaw;
    const uint64_t blockBase = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const GcSeqRaw* R = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint64_t* out = M + (uint64_t)b * GC_LZMA_MAX_ITEMS;
    uint32_t nOut = 0, lenBefore = 0;        // uniform running totals
#ifdef HIPEMU
    if (getenv("GC_TRACE_LZ") && t == 0) fprintf(stderr, "prep start block %u nRaw %u len %u\n", b, nRaw, blockLen);
#endif
    // one virtual record behind the last one stands for the block end (closes the trailing literal run)
    for (uint32_t tb = 0; tb <= nRaw; tb += LZP_T) {
        const uint32_t i = tb + t;
        uint32_t head = 0, off = 0, ml = 0, myLen = 0, rank = 0;
        if (i < nRaw) {
            const GcSeqRaw r = R[i];
            uint32_t prevRank = 0xFFFFFFFFu, prevOff = 0;
            if (i) { const GcSeqRaw q = R[i - 1u]; prevRank = q.litRank; prevOff = q.offml >> 8; }
            rank = r.litRank; off = r.offml >> 8; myLen = r.offml & 0xFFu; ml = myLen;
            head = (i == 0u || rank != prevRank || off != prevOff) ? 1u : 0u;
            if (head) for (uint32_t k = i + 1u; k < nRaw; k++) {          // absorb the chain of capped records
                const GcSeqRaw c = R[k];
                if (c.litRank != rank || (c.offml >> 8) != off) break;
                ml += c.offml & 0xFFu;
            }
        } else if (i == nRaw) head = 1u;
        // start position of a record = literals before it + lengths of all records before it
        uint32_t lTot;
        const uint32_t lBefore = lzp_excl_scan(myLen, sWave, &lTot);
        uint32_t pos = rank + lenBefore + lBefore;
        if (i == nRaw) pos = blockLen;
        // the previous head (walk back over the continuation records): where its coded match ends, and its distance
        uint32_t a = 0, prevOff = 1u, cnt = 0, mStart = 0, mEnd = 0;
        if (head) {
            if (i) {
                uint32_t j = i - 1u, plen = 0;
                const GcSeqRaw last = R[j];
                plen = last.offml & 0xFFu;
                while (j > 0u) { const GcSeqRaw q = R[j - 1u]; if (q.litRank != last.litRank || (q.offml >> 8) != (last.offml >> 8)) break; plen += q.offml & 0xFFu; j--; }
                const uint32_t pEnd = last.litRank + lenBefore + lBefore;     // literals in front of it + all record lengths up to it
                uint32_t ps, pe; lzp_trim(pEnd - plen, plen, ps, pe);
                a = pe; prevOff = last.offml >> 8;
            }
            lzp_trim(pos, ml, mStart, mEnd);
            cnt = lzp_items(a, mStart, mEnd, off, prevOff, i == nRaw, nullptr);
        }
#ifdef HIPEMU
        if (getenv("GC_TRACE_LZ") && head && cnt > 2000) fprintf(stderr, "  head i=%u pos=%u ml=%u a=%u mS=%u mE=%u cnt=%u\n", i, pos, ml, a, mStart, mEnd, cnt);
#endif
        uint32_t cTot;
        const uint32_t at = nOut + lzp_excl_scan(cnt, sWave, &cTot);
        if (head && at + cnt <= GC_LZMA_MAX_ITEMS) lzp_items(a, mStart, mEnd, off, prevOff, i == nRaw, out + at);
        nOut += cTot; lenBefore += lTot;
    }
#ifdef HIPEMU
    if (getenv("GC_TRACE_LZ") && t == 0) fprintf(stderr, "prep block %u nRaw %u items %u\n", b, nRaw, nOut);
#endif
    if (t == 0) nM[b] = nOut <= GC_LZMA_MAX_ITEMS ? nOut : 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------- L2: model
// (p, bit) stream between L2 and L3: one 16-bit word per coded bit.
//   adaptive bit : bit11 = bit, bits 0..10 = probability of a 0 BEFORE the update
//   direct bit   : bit14 set, bit11 = bit   (range halving, no probability)
#define LZW_BIT    0x0800u
#define LZW_DIRECT 0x4000u
// event in LDS: adaptive bit = index << 1 | bit (index < 2^13), direct bit = 0x8000 | bit
#define LZE_DIRECT 0x8000u
#define LZ2_EVCAP  1600u              // events buffered per round
#define LZ2_TICKS  2048u              // ticket bytes: a probability's ticket is the byte (index mod LZ2_TICKS).  Probabilities that share a
                                      // byte are merely ranked together (an extra round now and then); the point is LDS: 19.8 KiB per wave
                                      // = eight waves per CU instead of five, and the kernel's throughput is waves in flight
#define LZ2_EVMAX  (9u * GC_LZMA_LIT_CUT + 48u)                   // most events of one item

struct LzEv { uint16_t* p; uint32_t n; bool store; };
__device__ __forceinline__ void ev_put(LzEv& o, uint32_t idx, uint32_t bit) { if (o.store) o.p[o.n] = (uint16_t)((idx << 1) | (bit & 1u)); o.n++; }
__device__ __forceinline__ void ev_direct(LzEv& o, uint32_t bit) { if (o.store) o.p[o.n] = (uint16_t)(LZE_DIRECT | (bit & 1u)); o.n++; }

// state after `k` literals starting from state s (LzmaDec.c: state < 4 -> 0, < 10 -> s-3, else s-6)
__device__ __forceinline__ uint32_t lz_lit_advance(uint32_t s, uint32_t k)
{
    for (uint32_t i = 0; i < 3u && i < k; i++) s = s < 4u ? 0u : (s < 10u ? s - 3u : s - 6u);
    return k >= 3u ? 0u : s;
}

// length coder (len >= 2).  base = LZP_LEN or LZP_REPLEN
__device__ __forceinline__ void lz_gen_len(LzEv& o, uint32_t base, uint32_t len, uint32_t posState)
{
    uint32_t v = len - 2u;
    if (v < 8u) {
        ev_put(o, base + LZL_CHOICE, 0);
        const uint32_t tb = base + LZL_LOW + posState * 8u; uint32_t m = 1;
        for (int i = 2; i >= 0; i--) { const uint32_t bit = (v >> i) & 1u; ev_put(o, tb + m, bit); m = (m << 1) | bit; }
    } else if (v < 16u) {
        ev_put(o, base + LZL_CHOICE, 1); ev_put(o, base + LZL_CHOICE2, 0);
        v -= 8u;
        const uint32_t tb = base + LZL_MID + posState * 8u; uint32_t m = 1;
        for (int i = 2; i >= 0; i--) { const uint32_t bit = (v >> i) & 1u; ev_put(o, tb + m, bit); m = (m << 1) | bit; }
    } else {
        ev_put(o, base + LZL_CHOICE, 1); ev_put(o, base + LZL_CHOICE2, 1);
        v -= 16u;
        const uint32_t tb = base + LZL_HIGH; uint32_t m = 1;
        for (int i = 7; i >= 0; i--) { const uint32_t bit = (v >> i) & 1u; ev_put(o, tb + m, bit); m = (m << 1) | bit; }
    }
}

// one match-type symbol.  kind: 0 normal match, 1 rep0 (long), 2 rep1, 3 short rep (one byte at rep0: LzmaDec.c "IsRep0Long = 0"), 4 rep2, 5 rep3.
// dist = distance-1.  At most 48 events.
__device__ __forceinline__ void lz_gen_match(LzEv& o, uint32_t kind, uint32_t len, uint32_t dist, uint32_t state, uint32_t posState)
{
    ev_put(o, LZP_ISMATCH + state * 4u + posState, 1);
    if (kind == 3u) { ev_put(o, LZP_ISREP + state, 1); ev_put(o, LZP_ISREPG0 + state, 0); ev_put(o, LZP_ISREP0LONG + state * 4u + posState, 0); return; }
    if (kind == 0u) {
        ev_put(o, LZP_ISREP + state, 0);
        lz_gen_len(o, LZP_LEN, len, posState);
        const uint32_t lenState = len - 2u < 3u ? len - 2u : 3u;
        uint32_t slot;
        if (dist < 4u) slot = dist;
        else { const uint32_t hb = gc_hibit32(dist); slot = 2u * hb + ((dist >> (hb - 1u)) & 1u); }
        { const uint32_t tb = LZP_POSSLOT + lenState * 64u; uint32_t m = 1;
          for (int i = 5; i >= 0; i--) { const uint32_t bit = (slot >> i) & 1u; ev_put(o, tb + m, bit); m = (m << 1) | bit; } }
        if (slot >= 4u) {
            const uint32_t footer = (slot >> 1) - 1u, base = (2u | (slot & 1u)) << footer, red = dist - base;
            if (slot < 14u) {
                const uint32_t tb = LZP_SPECPOS + (slot - 4u) * 32u; uint32_t m = 1;
                for (uint32_t i = 0; i < footer; i++) { const uint32_t bit = (red >> i) & 1u; ev_put(o, tb + m, bit); m = (m << 1) | bit; }
            } else {
                for (uint32_t i = footer - 4u; i-- > 0u; ) ev_direct(o, (red >> (4u + i)) & 1u);          // high bits first
                uint32_t m = 1;
                for (uint32_t i = 0; i < 4u; i++) { const uint32_t bit = (red >> i) & 1u; ev_put(o, LZP_ALIGN + m, bit); m = (m << 1) | bit; }
            }
        }
    } else {
        ev_put(o, LZP_ISREP + state, 1);
        if (kind == 1u) { ev_put(o, LZP_ISREPG0 + state, 0); ev_put(o, LZP_ISREP0LONG + state * 4u + posState, 1); }
        else if (kind == 2u) { ev_put(o, LZP_ISREPG0 + state, 1); ev_put(o, LZP_ISREPG1 + state, 0); }
        else { ev_put(o, LZP_ISREPG0 + state, 1); ev_put(o, LZP_ISREPG1 + state, 1); ev_put(o, LZP_ISREPG2 + state, kind == 5u ? 1u : 0u); }       // LzmaDec.c: IsRepG2 picks rep2 / rep3
        lz_gen_len(o, LZP_REPLEN, len, posState);
    }
}

// one literal: isMatch=0 + 8 tree bits (plain, or "matched" after a match: LzmaDec.c MATCHED_LITER_DEC)
__device__ __forceinline__ void lz_gen_literal(LzEv& o, uint32_t cur, uint32_t prev, uint32_t matchByte, uint32_t state, uint32_t posState, uint32_t lc, uint32_t lpMask)
{
    ev_put(o, LZP_ISMATCH + state * 4u + posState, 0);
    const uint32_t pb = LZP_LITERAL + 0x300u * (((posState & lpMask) << lc) + (prev >> (8u - lc)));      // LzmaDec.c: ((processedPos & lpMask) << lc) + (prevByte >> (8 - lc)); lc + lp <= 3 here
    if (state < 7u) {
        uint32_t m = 1;
        for (int i = 7; i >= 0; i--) { const uint32_t bit = (cur >> i) & 1u; ev_put(o, pb + m, bit); m = (m << 1) | bit; }
    } else {
        uint32_t offs = 0x100u, sym = cur | 0x100u, mb = matchByte;
        for (int i = 0; i < 8; i++) {
            mb <<= 1;
            ev_put(o, pb + offs + (mb & offs) + (sym >> 8), (sym >> 7) & 1u);
            sym <<= 1;
            offs &= ~(mb ^ sym);
        }
    }
}

struct LzItem { uint32_t pos, len, off; };
__device__ __forceinline__ LzItem lz_item(const uint64_t* M, uint32_t idx)
{
    const uint64_t m = M[idx];
    LzItem it; it.pos = (uint32_t)(m & 0x3FFFFu); it.len = (uint32_t)((m >> 18) & 0xFFFFu); it.off = (uint32_t)(m >> 34);
    return it;
}
// events of one item: its literals [lp, pos) then its match.  st0 = coder state at the first literal, rep0 = current repeat
// distance (matched literal), S = block base (S[-1] exists iff blockBase > 0).  The literal bytes (<= GC_LZMA_LIT_CUT of them,
// plus the byte in front) are fetched with three 8-byte loads up front instead of one dependent byte load per literal.
__device__ __forceinline__ void lz_gen_item(LzEv& o, const uint8_t* S, uint64_t blockBase, uint64_t srcSize, uint32_t hasPrev, uint32_t lp, const LzItem& it,
                                            uint32_t st0, uint32_t rep0, uint32_t kind, uint32_t lc, uint32_t lpMask)
{
    uint32_t st = st0;
    const uint32_t ll = it.pos - lp;
    if (ll) {
        const uint64_t absLp = blockBase + lp;
        uint64_t x0 = 0, x1 = 0, x2 = 0;                            // bytes absLp - 1 .. absLp + 22
        const bool fast = absLp + hasPrev >= 1u && absLp + 23u <= srcSize;           // hasPrev: the buffer is a later part of a stream
        if (fast) { x0 = gc_ld64(S + lp - 1); x1 = gc_ld64(S + lp + 7); x2 = gc_ld64(S + lp + 15); }
        const uint32_t mb0 = st >= 7u ? S[(int64_t)lp - (int64_t)rep0] : 0u;      // only a literal right behind a match is "matched"
        uint32_t prev = fast ? (uint32_t)(x0 & 0xFFu) : (absLp + hasPrev ? S[(int64_t)lp - 1] : 0u);
        for (uint32_t i = 0; i < ll; i++) {
            uint32_t cur;
            if (fast) { const uint32_t j = i + 1u; const uint64_t x = j < 8u ? x0 : (j < 16u ? x1 : x2); cur = (uint32_t)(x >> ((j & 7u) * 8u)) & 0xFFu; }
            else cur = S[lp + i];
            lz_gen_literal(o, cur, prev, i == 0u ? mb0 : 0u, st, (lp + i) & 3u, lc, lpMask);
            st = st < 4u ? 0u : (st < 10u ? st - 3u : st - 6u);
            prev = cur;
        }
    }
    if (it.len) lz_gen_match(o, kind, it.len, it.off - 1u, st, it.pos & 3u);
}

// The decoder's four repeat distances as an LRU list (rep4 mode, round 3: written and checked on the emulator, NOT enabled in the shipped library yet).
// LZMA moves the distance a match uses to the front of (rep0..rep3) -- a new one pushes the last out -- so as long as the encoder codes EVERY match
// whose distance is in the list as that repeat (first index), the list is "the four most recently used distances, most recent first".  A run of
// accesses is summarised by its distinct distances in recency order (at most four), and two runs compose: S(AB) = S(B) ++ (S(A) without S(B)) -- an
// associative operation, so a wave scan gives every item the summary of the items in front of it; the list in front of the tile (wave-uniform; it may
// hold equal entries while the initial 1, 1, 1, 1 is still in it) follows with the first occurrence of every summarised distance taken out.
struct LzLru { uint32_t v0, v1, v2, v3; };                         // 0 = empty (distances are >= 1)
__device__ __forceinline__ void lru_put(LzLru& a, uint32_t x, bool dedupe)
{
    if (x == 0u || (dedupe && (x == a.v0 || x == a.v1 || x == a.v2 || x == a.v3))) return;
    if (!a.v0) a.v0 = x; else if (!a.v1) a.v1 = x; else if (!a.v2) a.v2 = x; else if (!a.v3) a.v3 = x;
}
// later ++ (earlier without later's entries): both hold distinct entries
__device__ __forceinline__ LzLru lru_join(const LzLru& later, const LzLru& earlier)
{
    LzLru r = later;
    lru_put(r, earlier.v0, true); lru_put(r, earlier.v1, true); lru_put(r, earlier.v2, true); lru_put(r, earlier.v3, true);
    return r;
}
// summary ++ (the decoder's list without the FIRST occurrence of every summarised distance)
__device__ __forceinline__ LzLru lru_over(const LzLru& sum, LzLru dec)
{
    const uint32_t e[4] = { sum.v0, sum.v1, sum.v2, sum.v3 };
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t x = e[j];
        if (x == 0u) continue;
        if (dec.v0 == x) dec.v0 = 0; else if (dec.v1 == x) dec.v1 = 0; else if (dec.v2 == x) dec.v2 = 0; else if (dec.v3 == x) dec.v3 = 0;
    }
    LzLru r = sum;
    lru_put(r, dec.v0, false); lru_put(r, dec.v1, false); lru_put(r, dec.v2, false); lru_put(r, dec.v3, false);
    return r;
}

extern "C" __global__ void __launch_bounds__(64)
gc_lzma2_model_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, const uint64_t* __restrict__ Mall,
                      const uint32_t* __restrict__ nM, uint32_t segLog, uint32_t hasPrev /* the byte in front of src exists (src is a later part of a stream) */,
                      uint16_t* __restrict__ stream, GcLzmaChunkInfo* __restrict__ cinfo,
                      const uint32_t* __restrict__ winCost /* W7's estimate per 4 KiB window of each block (1/16 bit), or nullptr */,
                      unsigned long long* __restrict__ prof /* optional phase profile (shader-clock sums over all segments): [0] tile header
                         (item loads + scans), [1] event generation, [2] event application, [3] application steps (64 events each),
                         [4] rounds played, [5] segments; nullptr = off */,
                      uint32_t mergeWords /* neighbouring rc chunks of a segment are coded as ONE LZMA2 chunk while their coded bits stay
                         within this many words (0 = never): a chunk costs 10 bytes (5 of header, 5 of range-coder start / flush), which
                         is 1 % of a well-compressed 4 KiB; the bound keeps the longest range-coder chain what it is for 4 KiB of
                         incompressible data */,
                      uint32_t wordCap /* words the segment may produce (its reserved place, GC_LZMA_STREAM_WORDS); beyond that it is stored */,
                      uint32_t rep4 /* 1: matches whose distance is rep2 / rep3 of the decoder are coded as such (LzLru above); 0: rep0 / rep1 only */,
                      uint8_t* __restrict__ segProps /* out, per segment: the LZMA props byte its first chunk carries (lc / lp chosen per segment) */,
                      uint32_t litSel /* 1: choose the literal context bits per segment; 0: the reference's lc = 3, lp = 0 */,
                      uint32_t segMerge /* round 5: 2 / 4 / 8 = a model segment may span that many 128 KiB blocks (aligned groups, power of two) where the parse prices
                         them cheaply enough; 0 / 1 = off.  Needs segLog = 17, winCost and rep4 */,
                      uint32_t mergeBudget /* ... while the blocks' estimated cost (winCost units, 1/16 bit) stays within this: the model chain of a merged
                         segment is then no longer than that of ONE block of poorly compressible data, which is what the launch takes anyway */)
{
    __shared__ __attribute__((aligned(16))) uint16_t P[LZP_TOTAL];
    __shared__ uint32_t sTick[LZ2_TICKS / 4u];                    // ticket bytes (see LZ2_TICKS)
    __shared__ uint16_t sEv[LZ2_EVCAP];
    __shared__ uint32_t sWordEnd[GC_LZMA_RC_PER_BLOCK];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t segSize = 1u << segLog, perBlock = GC_ZSTD_BLOCK_MAX >> segLog, rcPerSeg = segSize >> GC_LZMA_RC_LOG;
    // ---- model segments over several blocks (round 5).  A state reset costs what the model needs to learn the data again, and that does not shrink with the
    // data's size: on 8 MiB of ROCm shared objects that compress 9 : 1 an exact LZMA model prices the same parse 1.9 % / 3.1 % / 3.5 % smaller with resets every
    // 256 KiB / 512 KiB / 1 MiB than every 128 KiB (tools/lzma_parse_lab.c), and the reference's own slices are >= 218 KiB at level 5 with 64 threads
    // (fl2_compress.c:272-292: 14 MiB of new data per 16 MiB dictionary block / threads).  What a wave of this kernel costs is its coded bits, not its input bytes,
    // so an aligned group of 2 / 4 / 8 blocks whose estimated cost (the parse's winCost) stays within what ONE poorly compressible block costs is modelled by ONE
    // wave as ONE segment: the group's first block is the leader (state reset + props byte), the others continue its probabilities, coder state and repeat
    // distances (props entry 0xFF; their first chunk is an ordinary 0x80 chunk).  Everything else stays per block: item list, word stream place, rc chunks.
    uint32_t seg = blockIdx.x, nMember = 1u;
    if (segMerge >= 2u && perBlock == 1u && winCost != nullptr && rep4 != 0u) {
        const uint32_t nBlocksAll = (uint32_t)((srcSize + GC_ZSTD_BLOCK_MAX - 1u) / GC_ZSTD_BLOCK_MAX);
        const uint32_t g0 = seg & ~(segMerge - 1u);
        // lane j * 8 + k: (k-th group of four windows) of block g0 + j; est[j] = the block's estimate, 0xFFFFFFFF = absent or hopeless (never merged)
        uint32_t e = 0;
        { const uint32_t j = lane >> 3, bb = g0 + j;
          if (j < segMerge && bb < nBlocksAll) {
              const uint64_t bs = (uint64_t)bb * GC_ZSTD_BLOCK_MAX; const uint32_t bl = (uint32_t)((srcSize - bs) < GC_ZSTD_BLOCK_MAX ? (srcSize - bs) : GC_ZSTD_BLOCK_MAX);
              const uint32_t w0 = (lane & 7u) * 4u; const uint32_t* wc = winCost + (uint64_t)bb * GC_LZMA_RC_PER_BLOCK + w0;
#pragma unroll
              for (uint32_t q = 0; q < 4u; q++) if (((w0 + q) << GC_LZMA_RC_LOG) < bl) e += wc[q];      // (windows behind the end of a short last block were never written)
          }
          e += __shfl_xor(e, 1); e += __shfl_xor(e, 2); e += __shfl_xor(e, 4); }
        uint32_t est[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; j++) {
            const uint32_t bb = g0 + j;
            uint32_t v = __shfl(e, (int)(j * 8u));
            if (j >= segMerge || bb >= nBlocksAll) v = 0xFFFFFFFFu;
            else { const uint64_t bs = (uint64_t)bb * GC_ZSTD_BLOCK_MAX; const uint32_t bl = (uint32_t)((srcSize - bs) < GC_ZSTD_BLOCK_MAX ? (srcSize - bs) : GC_ZSTD_BLOCK_MAX);
                   if (v >= bl * 128u || nM[bb] == 0xFFFFFFFFu) v = 0xFFFFFFFFu; }
            est[j] = v;
        }
        const uint32_t me = seg - g0;
        uint32_t lead = me; nMember = 1u;
        for (uint32_t sz = segMerge; sz >= 2u; sz >>= 1) {           // the largest aligned group around this block that fits the budget
            const uint32_t a = me & ~(sz - 1u);
            uint64_t sum = 0; bool ok = true;
#pragma unroll
            for (uint32_t j = 0; j < 8u; j++) if (j >= a && j < a + sz) { if (est[j] == 0xFFFFFFFFu) ok = false; else sum += est[j]; }
            if (ok && sum <= (uint64_t)mergeBudget) { lead = a; nMember = sz; break; }
        }
        if (lead != me) return;                                     // a member: its leader's wave does the work
    }
    bool overflow = false;            // the words of a block outgrew their reserved place: the whole segment is stored
    uint32_t cOff = 1u;               // effective distance of the previous item (1 until the segment's first match)
    uint32_t cExit = 0;               // coder state after the last match item (0: state at segment start)
    uint32_t cLits = 0;               // literals coded since the last match item (in cut items)
    bool segHasMatch = false;         // a match in an earlier block of this segment
    bool anyDemoted = false;          // (wave-uniform) a one-byte item of this segment was coded as a literal: the rep0 / rep1 scans (a cross-check under the emulator) no longer apply
    LzLru cLru; cLru.v0 = cLru.v1 = cLru.v2 = cLru.v3 = 1u;                 // rep4: the decoder's repeat distances after a state reset (LzmaDec.c: reps = 1)
    uint32_t litLc = GC_LZMA_LC, litLpMask = 0u;
    const uint32_t segLead = seg;
    unsigned long long tprev = prof ? gc_clock() : 0ull, pc0 = 0, pc1 = 0, pc2 = 0; uint32_t pSteps = 0, pRounds = 0;
  for (uint32_t bi = 0; bi < nMember; bi++) {
    seg = segLead + bi;
    const uint32_t b = seg / perBlock, sInB = seg % perBlock;
    const uint64_t blockBase = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const uint32_t ss = sInB << segLog;
    GcLzmaChunkInfo* CI = cinfo + (uint64_t)b * GC_LZMA_RC_PER_BLOCK + (ss >> GC_LZMA_RC_LOG);
    const uint32_t nItems = nM[b];
    // A segment that the price-based parse already prices at or above its raw size (random or encrypted data) is stored without
    // being modelled: nothing is lost -- it would be stored after coding anyway -- and such segments are the longest chains of
    // this kernel and of the range coder (9 coded bits per byte), i.e. the tail of both launches.
    bool hopeless = false;
    if (winCost != nullptr && ss < blockLen) {
        uint32_t est = 0;
        for (uint32_t w = lane; w < rcPerSeg; w += 64u) { const uint32_t cs = ss + (w << GC_LZMA_RC_LOG); if (cs < blockLen) est += winCost[(uint64_t)b * GC_LZMA_RC_PER_BLOCK + (cs >> GC_LZMA_RC_LOG)]; }
        est = gc_wave_sum(est);
        const uint32_t segLen = (ss + segSize < blockLen ? ss + segSize : blockLen) - ss;
        hopeless = est >= segLen * 128u;                          // 8 bits per byte in 1/16 bit units
    }
    if (lane == 0u) segProps[seg] = bi == 0u ? (uint8_t)GC_LZMA_PROPS : (uint8_t)0xFFu;
    if (ss >= blockLen || nItems == 0xFFFFFFFFu || hopeless) {                // (never a block of a merged segment: those exist, have a list and are not hopeless)
        // no such segment -- or the item list overflowed (cannot happen for lists built by L1; kept as a guard): store it
        for (uint32_t c = lane; c < rcPerSeg; c += 64u) {
            const uint32_t cs = ss + (c << GC_LZMA_RC_LOG);
            GcLzmaChunkInfo ci; ci.usize = cs < blockLen ? ((blockLen - cs) < GC_LZMA_RC_SIZE ? (blockLen - cs) : GC_LZMA_RC_SIZE) : 0u;
            ci.csize = 0xFFFFFFFFu; ci.wordStart = 0; ci.wordEnd = 0; CI[c] = ci;
        }
        return;
    }
    const uint32_t se = ss + segSize < blockLen ? ss + segSize : blockLen;
    const uint8_t* S = src + blockBase;                   // block-relative addressing; S[-1] exists iff blockBase > 0
    const uint64_t* M = Mall + (uint64_t)b * GC_LZMA_MAX_ITEMS;

    // ---- literal context bits of this segment.  Every model segment starts with a state reset and carries a props byte (gc_lzma2_frame.hip), so lc / lp
    // are free per segment.  The reference codes everything with lc 3 / lp 0 (fl2_compress.c:116-138); machine code and tables of 2- / 4-byte fields code their
    // literals better by POSITION (liblzma on ROCm shared objects: lc 0 / lp 2 -2.1 %, lc 2 / lp 1 -1.8 %; text +0.3-0.5 %).  Chosen by the order-0 cost of a
    // quarter of the segment's bytes under each of three context functions (all with <= 8 contexts: the literal coder keeps its size), the reference's unless
    // another is 1/64 cheaper.  Counters: two per 32-bit word in the (not yet initialised) probability array.
    if (bi == 0u) {
    if (litSel && se - ss >= 4096u) {
        uint32_t* W = (uint32_t*)P;
        constexpr uint32_t NCELL = 2048u + 2048u + 1024u;
        for (uint32_t i = lane; i < NCELL / 2u; i += 64u) W[i] = 0u;
        gc_wave_sync();
        const uint32_t nGroups = (se - ss) >> 6;                  // one 16-byte piece out of every 64 bytes: <= 32 Ki samples, counters stay below 2^16
        for (uint32_t g = lane; g < nGroups; g += 64u) {
            const uint32_t o = ss + (g << 6);
            uint8_t by[16]; __builtin_memcpy(by, S + o, 16);
            uint32_t prev = (blockBase + o + hasPrev) != 0u ? S[(int64_t)o - 1] : 0u;
#pragma unroll
            for (uint32_t j = 0; j < 16u; j++) {
                const uint32_t cur = by[j];
                const uint32_t cA = ((prev >> 5) << 8) + cur, cB = 2048u + (((((j & 1u) << 2) + (prev >> 6))) << 8) + cur, cC = 4096u + ((j & 3u) << 8) + cur;
                atomicAdd(&W[cA >> 1], 1u << ((cA & 1u) << 4)); atomicAdd(&W[cB >> 1], 1u << ((cB & 1u) << 4)); atomicAdd(&W[cC >> 1], 1u << ((cC & 1u) << 4));
                prev = cur;
            }
        }
        gc_wave_sync();
        uint32_t cost[3] = { 0u, 0u, 0u };
        for (uint32_t ctx = 0; ctx < 20u; ctx++) {                // 8 + 8 + 4 contexts of 256 cells
            uint32_t n[4], tot = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) { const uint32_t cell = (ctx << 8) + (q << 6) + lane; n[q] = (W[cell >> 1] >> ((cell & 1u) << 4)) & 0xFFFFu; tot += n[q]; }
            tot = gc_wave_sum(tot);
            uint32_t part = 0;
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) if (n[q]) part += n[q] * pz_price(4u * n[q] + 1u, 4u * tot + 256u);
            part = gc_wave_sum(part);
            cost[ctx < 8u ? 0u : (ctx < 16u ? 1u : 2u)] += part >> 4;
        }
        uint32_t best = 0u;
        if ((unsigned long long)cost[1] * 64u < (unsigned long long)cost[0] * 63u) best = 1u;
        if ((unsigned long long)cost[2] * 64u < (unsigned long long)cost[0] * 63u && cost[2] < cost[best]) best = 2u;
        best = gc_uniform(best);
#ifdef HIPEMU
        { static const int xf = getenv("GC_X_LITFORCE") ? atoi(getenv("GC_X_LITFORCE")) : -1; if (xf >= 0) best = (uint32_t)xf;
          if (getenv("GC_X_LITSHOW") && lane == 0u) fprintf(stderr, "seg %u cost %u %u %u best %u\n", seg, cost[0], cost[1], cost[2], best); }
#endif
        if (best == 1u) { litLc = 2u; litLpMask = 1u; } else if (best == 2u) { litLc = 0u; litLpMask = 3u; }
        if (lane == 0u) segProps[seg] = (uint8_t)((GC_LZMA_PB * 5u + (best == 1u ? 1u : (best == 2u ? 2u : 0u))) * 9u + litLc);
        gc_wave_sync();
    }
    for (uint32_t i = lane; i < LZP_TOTAL; i += 64u) P[i] = 1024u;
    for (uint32_t i = lane; i < LZ2_TICKS / 4u; i += 64u) sTick[i] = 0;
    }
    if (lane < GC_LZMA_RC_PER_BLOCK) sWordEnd[lane] = 0;

    // items of this segment = items that END in (ss, se]  (a cut at position ss closes the previous segment)
    uint32_t first, last;
    { uint32_t lo = 0, hi = nItems;
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; const LzItem x = lz_item(M, mid); if (x.pos + x.len > ss) hi = mid; else lo = mid + 1u; }
      first = lo; hi = nItems;
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; const LzItem x = lz_item(M, mid); if (x.pos + x.len > se) hi = mid; else lo = mid + 1u; }
      last = lo; }
    first = gc_uniform(first); last = gc_uniform(last);
    uint16_t* W = stream + (uint64_t)seg * GC_LZMA_STREAM_WORDS(segLog);
    gc_wave_sync();

    // carried across tiles of 64 items (all wave-uniform)
    // (cOff, cExit, cLits, cLru live across the blocks of a merged segment: declared above)
    uint32_t cursor = ss;             // end of the previous item
    uint32_t cRun = 0;                // (k + 1) of the start of the run of equal distances containing the previous item; 0 = virtual
    uint32_t cMatch = 0;              // (k + 1) of the last match item OF THIS BLOCK; 0 = none yet (segHasMatch: one in an earlier block of the segment)
    uint32_t wpos = 0;                // words written so far (of this block: every block has its own place)

#define L2_PHASE(acc) do { if (prof) { const unsigned long long now_ = gc_clock(); acc += now_ - tprev; tprev = now_; } } while (0)
    for (uint32_t base = first; base < last; base += 64u) {
        const uint32_t cnt = last - base < 64u ? last - base : 64u;
#ifdef HIPEMU
        if (getenv("GC_TRACE_LZ") && lane == 0) fprintf(stderr, "seg %u items [%u,%u) tile base %u cnt %u wpos %u\n", seg, first, last, base, cnt, wpos);
#endif
        const uint32_t k = base - first + lane;                              // segment-relative item index
        LzItem it; it.pos = se; it.len = 0; it.off = 0;
        if (lane < cnt) it = lz_item(M, base + lane);
        // rep4: the decoder's repeat distances in front of every item, first (an LRU list as a wave scan, LzLru above).  One-byte items are short repeats: they name
        // rep0 and leave the list as it is, so they take no part in the scan -- and one whose distance is NOT the list's front (the parse offers the short repeat
        // from the repeat distance its window ARRIVED with, which is the path of another lane: W7L, round 5) is coded as the literal it also is: the item becomes a
        // cut one byte further on.  (Round 4 only let the parse name a short repeat behind a match of its own window: 14.5 K of them against the reference's 60 K
        // on 4 MiB of tables of records.)
        LzLru lruIncl; lruIncl.v0 = lruIncl.v1 = lruIncl.v2 = lruIncl.v3 = 0;
        LzLru L4; L4.v0 = L4.v1 = L4.v2 = L4.v3 = 0;
        {   // (always: with the hook GC_L2_REP4=0 the list is not used for rep2 / rep3, but a one-byte item that names another distance than the decoder's rep0 must still
            //  become the literal it also is -- W7L offers short repeats at distances it does not know for certain)
            LzLru a; a.v0 = (lane < cnt && it.len >= 2u) ? it.off : 0u; a.v1 = a.v2 = a.v3 = 0;
#pragma unroll
            for (uint32_t d = 1; d < 64u; d <<= 1) {
                LzLru b; b.v0 = __shfl_up(a.v0, d); b.v1 = __shfl_up(a.v1, d); b.v2 = __shfl_up(a.v2, d); b.v3 = __shfl_up(a.v3, d);
                if (lane < d) b.v0 = b.v1 = b.v2 = b.v3 = 0;
                a = lru_join(a, b);
            }
            lruIncl = a;
            LzLru ex; ex.v0 = __shfl_up(a.v0, 1); ex.v1 = __shfl_up(a.v1, 1); ex.v2 = __shfl_up(a.v2, 1); ex.v3 = __shfl_up(a.v3, 1);
            if (lane == 0u) ex.v0 = ex.v1 = ex.v2 = ex.v3 = 0;
            L4 = lru_over(ex, cLru);                                         // the decoder's list in front of this item
            const bool demote = lane < cnt && it.len == 1u && it.off != L4.v0;
            if (demote) { it.pos += 1u; it.len = 0u; it.off = 0u; }
            if (__any(demote)) anyDemoted = true;
        }
        const bool isM = lane < cnt && it.len != 0u;
        uint32_t prevEnd = __shfl_up(it.pos + it.len, 1); if (lane == 0) prevEnd = cursor;
        const uint32_t ll = lane < cnt ? it.pos - prevEnd : 0u;
        // last match item at or before / before this one
        uint32_t mIncl = gc_wave_incl_max(isM ? k + 1u : 0u); if (mIncl < cMatch) mIncl = cMatch;
        uint32_t mPrev = __shfl_up(mIncl, 1); if (lane == 0) mPrev = cMatch;
        // effective distance: a cut repeats the distance of the last match; before the segment's first match it is 1
        const bool mPrevInTile = mPrev != 0u && mPrev - 1u >= base - first;
        const uint32_t offAtMPrev = __shfl(it.off, (int)(mPrevInTile ? mPrev - 1u - (base - first) : 0u));
        const uint32_t offE = isM ? it.off : (mPrevInTile ? offAtMPrev : cOff);       // (cOff is 1 until the segment's first match)
        uint32_t pOff = __shfl_up(offE, 1); if (lane == 0) pOff = cOff;
        // repeat-distance history as scans: rep0 = previous distance, rep1 = distance before the current run of equal ones
        const uint32_t v = (lane < cnt && offE != pOff) ? k + 1u : 0u;
        uint32_t runIncl = gc_wave_incl_max(v); if (runIncl < cRun) runIncl = cRun;
        uint32_t runPrev = __shfl_up(runIncl, 1); if (lane == 0) runPrev = cRun;
        uint32_t rep1 = 1u;
        if (isM && runPrev >= 2u) {                                          // item in front of the run that contains item k-1
            const uint32_t j = runPrev - 2u;                                 // segment-relative; its effective distance:
            const LzItem x = lz_item(M, first + j);
            if (x.len) rep1 = x.off;
            else {                                                           // a cut: distance of the last match before it, if any
                uint32_t jj = j; rep1 = 1u;
                while (jj > 0u) { const LzItem y = lz_item(M, first + jj - 1u); if (y.len) { rep1 = y.off; break; } jj--; }
            }
        }
        uint32_t kind = 0;
        if (isM) kind = it.off == pOff ? (it.len == 1u ? 3u : 1u) : (it.off == rep1 ? 2u : 0u);
#ifdef HIPEMU
        if (bi == 0u && !anyDemoted && isM && it.len == 1u && kind != 3u) { fprintf(stderr, "L2: one-byte item at %u is not a repeat of the previous distance (%u vs %u)\n", it.pos, it.off, pOff); abort(); }
#endif
        if (rep4) {
            if (isM) {
                const uint32_t k4 = it.off == L4.v0 ? (it.len == 1u ? 3u : 1u) : (it.off == L4.v1 ? 2u : (it.off == L4.v2 ? 4u : (it.off == L4.v3 ? 5u : 0u)));
#ifdef HIPEMU
                if (bi == 0u && !anyDemoted && ((kind != 0u && k4 != kind) || (kind == 0u && k4 != 0u && k4 < 4u))) { fprintf(stderr, "L2 rep4: item at %u off %u: list %u %u %u %u says kind %u, the scans say %u\n", it.pos, it.off, L4.v0, L4.v1, L4.v2, L4.v3, k4, kind); abort(); }
#endif
                kind = k4;
            }
        }
        // coder state: literals of cut items since the last match, state after that match
        const uint32_t cl = (lane < cnt && !isM) ? ll : 0u;
        const uint32_t aIncl = gc_wave_incl_sum(cl), aExcl = aIncl - cl;
        const bool mInTile = mPrevInTile;
        const uint32_t mLane = mInTile ? mPrev - 1u - (base - first) : 0u;
        const uint32_t aAtM = __shfl(aExcl, (int)mLane);
        const uint32_t L = mInTile ? aExcl - aAtM : cLits + aExcl;           // literals between the last match and this item
        const bool litBefore = (L + ll) != 0u || (mPrev == 0u && !segHasMatch);      // <=> coder state before the match is a literal state
        const uint32_t stAfter = kind == 0u ? (litBefore ? 7u : 10u) : (kind == 3u ? (litBefore ? 9u : 11u) : (litBefore ? 8u : 11u));   // LzmaDec.c state updates
        const uint32_t exAtM = __shfl(stAfter, (int)mLane);
        const uint32_t exM = mInTile ? exAtM : cExit;                        // state after the last match before this item
        const uint32_t st0 = lz_lit_advance(exM, L);                         // state at this item's first literal
        // event counts -> offsets
        uint32_t nEv = 0;
        if (lane < cnt) {                                                    // 9 per literal + the match's own count (state-independent)
            LzEv o; o.p = nullptr; o.n = 9u * ll; o.store = false;
            if (it.len) lz_gen_match(o, kind, it.len, it.off - 1u, 0u, 0u);
            nEv = o.n;
        }
        const uint32_t evIncl = gc_wave_incl_sum(nEv);
        // rc chunk ends: the item that ends on a 4 KiB boundary (or at the segment end) closes its chunk
        if (lane < cnt) {
            const uint32_t end = it.pos + it.len;
            if ((end & (GC_LZMA_RC_SIZE - 1u)) == 0u || end == se) sWordEnd[(end - 1u - ss) >> GC_LZMA_RC_LOG] = wpos + evIncl;
        }
        // ---- rounds: as many items as fit the event buffer, generated by their lanes, applied 64 events at a time
        uint32_t done = 0;                                                   // items of this tile already coded
        uint32_t evDone = 0;                                                 // their events
        L2_PHASE(pc0);
        while (done < cnt) {
#ifdef HIPEMU
            if (getenv("GC_TRACE_LZ") && lane >= done && lane < cnt && nEv > LZ2_EVMAX) { fprintf(stderr, "  BAD item k=%u pos=%u len=%u off=%u prevEnd=%u ll=%u nEv=%u\n", base + lane, it.pos, it.len, it.off, prevEnd, ll, nEv); abort(); }
#endif
            const uint64_t fit = __ballot(lane >= done && lane < cnt && evIncl - evDone <= LZ2_EVCAP);
            const uint32_t upto = done + (uint32_t)__popcll(fit);            // items [done, upto) fit (prefix property of evIncl)
            const uint32_t evEnd = gc_readlane(evIncl, upto - 1u);
#ifdef HIPEMU
            if (getenv("GC_TRACE_LZ") && lane == 0) fprintf(stderr, "  round done %u upto %u evDone %u evEnd %u\n", done, upto, evDone, evEnd);
#endif
            if (lane >= done && lane < upto) {
                LzEv o; o.p = sEv + (evIncl - nEv - evDone); o.n = 0; o.store = true;
                lz_gen_item(o, S, blockBase, srcSize, hasPrev, prevEnd, it, st0, pOff, kind, litLc, litLpMask);
            }
            gc_wave_sync();
            L2_PHASE(pc1);
            const uint32_t total = evEnd - evDone;
            // The segment's words have a reserved place (9 per byte: a literal is 9 coded bits).  A match can take more words per byte than that
            // (a far 3- or 4-byte match: ~40), so a segment of almost only literals plus a few such matches can outgrow it: then it is not coded
            // at all but stored (found by the randomized round trips: the words ran into the next segment's place and its last chunk was garbage).
            if (wpos + total > wordCap) {
#ifdef HIPEMU
                if (getenv("GC_TRACE_OVF") && lane == 0) fprintf(stderr, "L2 overflow: seg %u (block %u) words %u + %u > %u, items [%u,%u) tile base %u, segment bytes %u\n", seg, b, wpos, total, wordCap, first, last, base, se - ss);
#endif
                overflow = true; break;
            }
            for (uint32_t e0 = 0; e0 < total; e0 += 64u) {
                const bool valid = e0 + lane < total;
                const uint32_t ev = valid ? sEv[e0 + lane] : LZE_DIRECT;
                const bool adaptive = (ev & LZE_DIRECT) == 0u;
                const uint32_t idx = (ev >> 1) & 0x1FFFu, bit = ev & 1u;
                const uint32_t tk = idx & (LZ2_TICKS - 1u), sh = (tk & 3u) * 8u;
                uint32_t rank = 0;
                if (adaptive) rank = (atomicAdd(&sTick[tk >> 2], 1u << sh) >> sh) & 0xFFu;   // lower lanes with the same ticket byte
                // The rounds communicate through LDS only, and the LDS unit executes the operations of one wave in program order: the
                // read of round r + 1 is queued behind the write of round r (no fence).  Measured (run 22-25, phase profile): a step
                // costs ~1400 cycles at ~3 rounds -- instruction issue as much as LDS latency -- so the loop is kept short (the
                // update is a select).  (Issuing the next step's ticket atomics and event loads ahead
                // of the rounds was tried and gained nothing.)
                const uint32_t my = adaptive ? rank : 0xFFFFFFFFu;
                const uint32_t left = adaptive ? rank + 1u : 0u;             // rounds this lane still needs
                uint32_t p = 0, nRounds = 0;
                for (uint32_t r = 0; __any(left > r); r++) {
                    if (my == r) {
                        p = P[idx];
                        const uint32_t up = p + ((2048u - p) >> 5), dn = p - (p >> 5);
                        P[idx] = (uint16_t)(bit ? dn : up);
                    }
                    gc_wave_step();
                    nRounds++;
                }
                if (prof) { pRounds += nRounds; pSteps++; }
                if (my == 0u) sTick[tk >> 2] = 0;                            // (lanes that share the word all write 0)
                if (valid) W[wpos + e0 + lane] = (uint16_t)(adaptive ? (p | (bit ? LZW_BIT : 0u)) : (LZW_DIRECT | (bit ? LZW_BIT : 0u)));
                gc_wave_step();
            }
            wpos += total; evDone = evEnd; done = upto;
            L2_PHASE(pc2);
        }
        if (overflow) break;
        // ---- carries
        const uint32_t lastLane = cnt - 1u;
        cursor = gc_readlane(it.pos + it.len, lastLane);
        cOff = gc_readlane(offE, lastLane);
        cRun = gc_readlane(runIncl, lastLane);
        const uint32_t mLast = gc_readlane(mIncl, lastLane);
        if (mLast != 0u && mLast - 1u >= base - first) {                     // the tile contains a match: restart the literal count
            const uint32_t ml_ = mLast - 1u - (base - first);
            cExit = gc_readlane(stAfter, ml_);
            cLits = gc_readlane(aIncl, lastLane) - gc_readlane(aIncl, ml_);
        } else cLits += gc_readlane(aIncl, lastLane);
        cMatch = mLast;
        if (rep4) {
            LzLru t; t.v0 = gc_readlane(lruIncl.v0, lastLane); t.v1 = gc_readlane(lruIncl.v1, lastLane); t.v2 = gc_readlane(lruIncl.v2, lastLane); t.v3 = gc_readlane(lruIncl.v3, lastLane);
            cLru = lru_over(t, cLru);
        }
        gc_wave_sync();
    }
    gc_wave_sync();
    if (overflow) break;
    segHasMatch = segHasMatch || cMatch != 0u;                               // (the next block of a merged segment starts from cOff / cExit / cLits / cLru as they are)
    // rc chunks -> LZMA2 chunks: greedy groups of neighbours (one lane; at most 32 chunks).  The leader carries the group (its
    // uncompressed size, the word range of all members), the other members are marked absent (usize 0), which is what every
    // later stage already skips; the group's range-coder output lies in the members' staging areas, which are contiguous.
    if (lane == 0u) {
        uint32_t c = 0;
        while (c < rcPerSeg) {
            const uint32_t cs = ss + (c << GC_LZMA_RC_LOG);
            GcLzmaChunkInfo ci; ci.csize = 0; ci.usize = 0; ci.wordStart = 0; ci.wordEnd = 0;
            if (cs >= se) { CI[c] = ci; c++; continue; }
            ci.usize = (se - cs) < GC_LZMA_RC_SIZE ? (se - cs) : GC_LZMA_RC_SIZE;
            ci.wordStart = c ? sWordEnd[c - 1u] : 0u; ci.wordEnd = sWordEnd[c];
            uint32_t k = 1;
            while (c + k < rcPerSeg && k < GC_LZMA_RC_GROUP_MAX) {
                const uint32_t cs2 = ss + ((c + k) << GC_LZMA_RC_LOG);
                if (cs2 >= se || sWordEnd[c + k] - ci.wordStart > mergeWords) break;
                ci.usize += (se - cs2) < GC_LZMA_RC_SIZE ? (se - cs2) : GC_LZMA_RC_SIZE;
                ci.wordEnd = sWordEnd[c + k];
                k++;
            }
            CI[c] = ci;
            GcLzmaChunkInfo none; none.usize = 0; none.csize = 0; none.wordStart = 0; none.wordEnd = 0;
            for (uint32_t j = 1; j < k; j++) CI[c + j] = none;
            c += k;
        }
    }
    gc_wave_sync();                                                          // (lane 0 has read sWordEnd: the next block may clear it)
  }
    if (prof && lane == 0u) {
        atomicAdd(&prof[0], pc0); atomicAdd(&prof[1], pc1); atomicAdd(&prof[2], pc2);
        atomicAdd(&prof[3], (unsigned long long)pSteps); atomicAdd(&prof[4], (unsigned long long)pRounds); atomicAdd(&prof[5], 1ull);
    }
    if (overflow) {                                                          // (wave-uniform) same marking as for a segment that is not modelled at all -- for EVERY block of the segment:
        for (uint32_t bi = 0; bi < nMember; bi++) {                          // the blocks behind the one that overflowed continue a model that was never coded, the ones in front go with them (the plan stores a segment whole)
            const uint32_t sg = segLead + bi, b = sg / perBlock, ss = (sg % perBlock) << segLog;
            const uint64_t blockBase = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
            const uint32_t blockLen = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
            GcLzmaChunkInfo* CI = cinfo + (uint64_t)b * GC_LZMA_RC_PER_BLOCK + (ss >> GC_LZMA_RC_LOG);
            if (lane == 0u) segProps[sg] = bi == 0u ? (uint8_t)GC_LZMA_PROPS : (uint8_t)0xFFu;
            for (uint32_t c = lane; c < rcPerSeg; c += 64u) {
                const uint32_t cs = ss + (c << GC_LZMA_RC_LOG);
                GcLzmaChunkInfo ci; ci.usize = cs < blockLen ? ((blockLen - cs) < GC_LZMA_RC_SIZE ? (blockLen - cs) : GC_LZMA_RC_SIZE) : 0u;
                ci.csize = 0xFFFFFFFFu; ci.wordStart = 0; ci.wordEnd = 0; CI[c] = ci;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- L3: range coder
// With the probabilities already resolved, what is left of LZMA's serial dependency is the range recurrence itself -- bound = (range >> 11) * p;
// range = bit ? range - bound : bound; renormalise (C/fast-lzma2/range_enc.h:62-152) -- and all chunks of an input are coded at the same time, so
// the kernel takes as long as ONE chunk's chain: what matters is the number of instructions per coded bit of a lane that has a SIMD to itself.
// Round 3: two kernels.  What RC_shiftLow (range_enc.c:123-140) does with `cache` and `cacheSize` -- hold back the last byte and the 0xFF bytes
// behind it until it is known whether a carry comes -- is a big-number addition done one digit at a time; it does not have to sit in the serial
// loop.  Every shift step takes the 9-bit digit low >> 24 (a byte and the carry into the byte in front of it) out of low:
//   L3a gc_lzma2_rc_kernel      one LANE per rc chunk (group of chunks): the recurrence, and one 16-bit store per shift step: the digit goes
//                               back into the word stream the lane is reading, which it has consumed further than it has written (a step shifts at
//                               most once: p >= 31).  No cache, no pending count, no byte window, no wave-level look at rare cases
//                               (8.4 -> 5.95 ms on the Silesia stand-in).
//   L3b gc_lzma2_rc_fin_kernel  one WAVE per chunk: output byte k = (digit[k-1] & 255) + (digit[k] >> 8) + the carry from byte k + 1; 64 bytes at
//                               a time from the chunk's end, the carries of a tile resolved by one 64-bit addition of its generate / propagate masks.
// The bytes are those of the reference's encoder for the same (probability, bit) sequence: n shift steps + 5 flush steps give n + 5 bytes.
__device__ __forceinline__ uint32_t rc_mul24(uint32_t a, uint32_t b)       // a < 2^21, b < 2^11: the full-rate 24-bit multiply is exact
{
#ifdef HIPEMU
    return a * b;
#else
    return __umul24(a, b);
#endif
}
struct LzRc {
    uint64_t low;                     // < 2^33
    uint32_t range;
    uint32_t nDig;                    // digits written so far
    uint16_t* dig;                    // where they go: the start of this chunk's words
};
// one coded bit
__device__ __forceinline__ void rc_word(LzRc& rc, uint32_t w)
{
    const bool bit = (w & LZW_BIT) != 0u, direct = (w & LZW_DIRECT) != 0u;
    const uint32_t bound = direct ? rc.range >> 1 : rc_mul24(rc.range >> 11, w & 0x7FFu);
    rc.low += bit ? bound : 0u;
    uint32_t range = (bit && !direct) ? rc.range - bound : bound;
    if (range < (1u << 24)) {                                      // RC_shiftLow; one step always suffices (p >= 31)
        rc.dig[rc.nDig++] = (uint16_t)(rc.low >> 24);              // (measured against a branch-free form that stores the digit at every bit and only counts it at a
        rc.low = (rc.low & 0xFFFFFFull) << 8;                      //  shift step: 5.95 ms with the branch, 8.3 ms without -- a store per coded bit costs more than the branch)
        range <<= 8;
    }
    rc.range = range;
}

extern "C" __global__ void __launch_bounds__(64)
gc_lzma2_rc_kernel(uint16_t* stream, uint32_t segLog, uint32_t nRc, uint8_t* __restrict__ rcOut, GcLzmaChunkInfo* __restrict__ cinfo)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t c = blockIdx.x * 64u + lane;
    GcLzmaChunkInfo ci; ci.usize = 0; ci.csize = 0; ci.wordStart = 0; ci.wordEnd = 0;
    if (c < nRc) ci = cinfo[c];
    const bool live = ci.usize != 0u && ci.csize != 0xFFFFFFFFu;
    if (!live) return;                                              // (nothing below is wave-synchronous)
    const uint32_t seg = c >> (segLog - GC_LZMA_RC_LOG);
    uint16_t* W = stream + (uint64_t)seg * GC_LZMA_STREAM_WORDS(segLog);
    LzRc rc; rc.low = 0; rc.range = 0xFFFFFFFFu; rc.nDig = 0; rc.dig = W + ci.wordStart;
    uint32_t k = ci.wordStart;
    const uint32_t end = ci.wordEnd;
    // words up to the next 16-byte boundary of the stream one by one, then eight at a time from one 16-byte load (the next load is in
    // flight while these are coded; digits land in front of word k, loads reach behind it), then the rest one by one
    while (k < end && (k & 7u) != 0u) { rc_word(rc, W[k]); k++; }
    const uint32_t nVec = k + 8u <= end ? (end - k) >> 3 : 0u;
    const GcU4* V = (const GcU4*)(W + k);
    GcU4 nxt; nxt.x = nxt.y = nxt.z = nxt.w = 0;
    if (nVec) nxt = V[0];
    for (uint32_t i = 0; i < nVec; i++) {
        const GcU4 cur = nxt;
        if (i + 1u < nVec) nxt = V[i + 1u];
        rc_word(rc, cur.x & 0xFFFFu); rc_word(rc, cur.x >> 16); rc_word(rc, cur.y & 0xFFFFu); rc_word(rc, cur.y >> 16);
        rc_word(rc, cur.z & 0xFFFFu); rc_word(rc, cur.z >> 16); rc_word(rc, cur.w & 0xFFFFu); rc_word(rc, cur.w >> 16);
    }
    k += nVec << 3;
    while (k < end) { rc_word(rc, W[k]); k++; }
    // what is left of low leaves through the five flush steps, which L3b derives from it: the value travels in the chunk's staging area
    *(uint64_t*)(rcOut + (uint64_t)c * GC_LZMA_RC_STRIDE) = rc.low;
    cinfo[c].csize = rc.nDig;
}

__device__ __forceinline__ uint64_t rc_rev64(uint64_t v) { return ((uint64_t)__brev((uint32_t)v) << 32) | (uint64_t)__brev((uint32_t)(v >> 32)); }

extern "C" __global__ void __launch_bounds__(64)
gc_lzma2_rc_fin_kernel(const uint16_t* __restrict__ stream, uint32_t segLog, uint32_t nRc, uint8_t* __restrict__ rcOut, GcLzmaChunkInfo* __restrict__ cinfo)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (c >= nRc) return;
    const GcLzmaChunkInfo ci = cinfo[c];
    if (ci.usize == 0u || ci.csize == 0xFFFFFFFFu) return;         // (uniform)
    const uint32_t seg = c >> (segLog - GC_LZMA_RC_LOG);
    const uint16_t* D = stream + (uint64_t)seg * GC_LZMA_STREAM_WORDS(segLog) + ci.wordStart;
    uint8_t* out = rcOut + (uint64_t)c * GC_LZMA_RC_STRIDE;
    const uint32_t nDig = ci.csize, n = nDig + 5u;                  // bytes of the chunk
    const uint64_t low = *(const uint64_t*)out;                     // (read by every lane before any lane writes: the barrier below)
    gc_wave_sync_global();
    const uint32_t members = (ci.usize + GC_LZMA_RC_SIZE - 1u) >> GC_LZMA_RC_LOG;       // rc chunks coded as this one LZMA2 chunk
    if (n > members * GC_LZMA_RC_STRIDE || n > 65536u) {            // did not fit the staging area / an LZMA2 chunk holds at most 64 KiB of coded bytes: the segment is stored
        if (lane == 0u) cinfo[c].csize = 0xFFFFFFFFu;
        return;
    }
    // digit k: a shift step's, or one of the five flush steps' (low >> 24 with its carry bit, then the three bytes below, then 0)
    auto digit = [&](uint32_t kk) -> uint32_t {
        if (kk < nDig) return D[kk];
        const uint32_t f = kk - nDig;
        return f == 0u ? (uint32_t)(low >> 24) & 0x1FFu : (f < 4u ? (uint32_t)(low >> (24u - 8u * f)) & 0xFFu : 0u);
    };
    uint32_t cin = 0;                                               // carry into the tile from the bytes behind it
    for (uint32_t tb = ((n - 1u) >> 6) << 6;; tb -= 64u) {
        const uint32_t kk = tb + lane;
        uint32_t v = 0;
        if (kk < n) v = (kk ? digit(kk - 1u) & 0xFFu : 0u) + (digit(kk) >> 8);          // 0 .. 256
        // bit i of the masks = the byte of significance i inside the tile (lane 63 is the least significant)
        const uint64_t G = rc_rev64(__ballot(v == 256u)), P = rc_rev64(__ballot(v == 255u));
        const uint64_t X = G | P, s1 = X + G, s2 = s1 + cin;
        const uint32_t cout = (s1 < X || s2 < s1) ? 1u : 0u;
        const uint64_t carryIn = s2 ^ P;                            // bit i: the carry that enters byte i  (X ^ G = P)
        if (kk < n) out[kk] = (uint8_t)(v + (uint32_t)((carryIn >> (63u - lane)) & 1ull));
        cin = cout;
        if (tb == 0u) break;
    }
    if (lane == 0u) cinfo[c].csize = n;
}
