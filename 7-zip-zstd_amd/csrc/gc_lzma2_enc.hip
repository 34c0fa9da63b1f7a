// gc_lzma2_enc.hip -- FLZMA2 path: LZMA2 chunk encoder for the matches found by K1 (gc_zstd_lz.hip).
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
// GPU structure.  Adaptive range coding is serial per coder state, so the unit of parallelism is the LZMA2 chunk: every
// chunk resets the coder state (control 0xC0/0xE0; the dictionary is NOT reset, so matches still reach back across
// chunks), exactly the device Fast-LZMA2 itself uses to run slices on several threads (lzma2_enc.h:22, lzma2_enc.c:2040-2075).
//   L1 gc_lzma2_prep_kernel  one workgroup per 128 KiB match-finder block: K1's capped records -> merged matches with
//                            their start positions (scan), M[j] = pos | len<<17 | off<<34
//   L2 gc_lzma2_model_kernel one WAVE per chunk.  The 64 lanes turn symbols into (probability index, bit) entries in
//                            parallel -- state machine, repeat-distance history, slot/length trees are all computed
//                            per symbol from local information (scans), nothing serial.  The probability updates of one
//                            symbol touch distinct entries, so they are applied by all lanes at once (LDS read-modify-write).
//                            Output: the chunk's stream of (probability, bit) words in HBM -- the adaptive model is now
//                            fully resolved and no longer needed.
//   L3 gc_lzma2_rc_kernel    one LANE per chunk: what remains serial is the range recurrence (bound = (range >> 11) * p,
//                            renormalise, carry), register-only work, so 64 chunks advance per wave instruction.
//   L4/L5 plan + emit        chunk headers and concatenation (gc_lzma2_frame.hip)
#include "gc_common.h"
#include "gc_device.h"
#include "gc_lzma2.h"
#ifdef HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif

#define LZP_T 256u

// ---------------------------------------------------------------------------------------------- L1: merge + positions
extern "C" __global__ void __launch_bounds__(LZP_T)
gc_lzma2_prep_kernel(const GcSeqRaw* __restrict__ seqRaw, const GcBlockMeta* __restrict__ meta,
                     uint64_t* __restrict__ M /* GC_MAX_SEQ_PER_BLOCK per block */, uint32_t* __restrict__ nM)
{
    __shared__ uint32_t sWave[LZP_T / 64u];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, b = blockIdx.x;
    const uint32_t nRaw = meta[b].nSeqRaw;
    const GcSeqRaw* R = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint64_t* out = M + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint32_t nOut = 0, lenBefore = 0;        // uniform running totals
    for (uint32_t tb = 0; tb < nRaw; tb += LZP_T) {
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
        }
        // two block-wide exclusive scans: heads (output slot) and record lengths (start position = litRank + lens before)
        uint32_t inclH = gc_wave_incl_sum(head), inclL = gc_wave_incl_sum(myLen);
        if (lane == 63u) sWave[wave] = (inclH << 20) | inclL;             // <= 256 heads, <= 256*64 length per tile
        __syncthreads();
        uint32_t hBefore = 0, lBefore = 0, hAll = 0, lAll = 0;
        for (uint32_t w = 0; w < LZP_T / 64u; w++) {
            const uint32_t c = sWave[w];
            if (w < wave) { hBefore += c >> 20; lBefore += c & 0xFFFFFu; }
            hAll += c >> 20; lAll += c & 0xFFFFFu;
        }
        __syncthreads();
        if (head) {
            const uint32_t pos = rank + lenBefore + lBefore + inclL - myLen;
            out[nOut + hBefore + inclH - 1u] = (uint64_t)pos | ((uint64_t)ml << 17) | ((uint64_t)off << 34);
        }
        nOut += hAll; lenBefore += lAll;
    }
    if (t == 0) nM[b] = nOut;
}

// ---------------------------------------------------------------------------------------------- L2: chunk encoder
// (p, bit) stream between L2 and L3: one 16-bit word per coded bit.
//   adaptive bit : bit11 = bit, bits 0..10 = probability of a 0 BEFORE the update
//   direct bit   : bit14 set, bit11 = bit   (range halving, no probability)
#define LZW_BIT    0x0800u
#define LZW_DIRECT 0x4000u
struct LzStream { uint16_t* w; uint32_t pos; };                 // pos is wave-uniform

// Apply the entries of ONE group held one per lane (lane k < cnt holds entry k, in coding order) to the probability model
// and append the resulting words to the chunk's stream.  Probability updates: all lanes at once, in `cnt/per` rounds of
// `per` lanes (round r = lanes [r*per, (r+1)*per)): inside one symbol all indices are distinct by construction (one node
// per tree depth); different symbols may share indices, so symbols are applied in order (LDS operations of a wave are in
// order).  Nothing here is serial in the number of coded bits.
__device__ __forceinline__ void lz_emit_group(LzStream& st, uint32_t e, uint32_t cnt, uint32_t per, uint16_t* P, uint32_t lane)
{
    const bool valid = lane < cnt;
    const bool isDirect = valid && (e >> 31) != 0u;
    uint32_t p = 0;
    for (uint32_t r0 = 0; r0 < cnt; r0 += per) {
        if (valid && !isDirect && lane >= r0 && lane < r0 + per) {
            const uint32_t idx = e >> 1;
            p = P[idx];
            P[idx] = (uint16_t)((e & 1u) ? p - (p >> 5) : p + ((2048u - p) >> 5));
        }
        gc_wave_sync();
    }
    const uint64_t dmask = __ballot(isDirect);
    if (dmask == 0ull) {                                             // common case: one word per lane, one coalesced store
        if (valid) st.w[st.pos + lane] = (uint16_t)(p | ((e & 1u) ? LZW_BIT : 0u));
        st.pos += cnt;
    } else {                                                         // a far match: its direct-bits entry expands to one word per bit
        const uint32_t nb = isDirect ? (e >> 26) & 31u : 0u;
        const uint32_t width = valid ? (isDirect ? nb : 1u) : 0u;
        const uint32_t incl = gc_wave_incl_sum(width);
        const uint32_t at = st.pos + incl - width;
        if (valid && !isDirect) st.w[at] = (uint16_t)(p | ((e & 1u) ? LZW_BIT : 0u));
        if (isDirect) { const uint32_t v = e & 0x03FFFFFFu; for (uint32_t i = 0; i < nb; i++) st.w[at + i] = (uint16_t)(LZW_DIRECT | (((v >> (nb - 1u - i)) & 1u) ? LZW_BIT : 0u)); }
        st.pos += gc_readlane(incl, 63u);
    }
}

#define ENT(idx, bit) ((((uint32_t)(idx)) << 1) | ((uint32_t)(bit) & 1u))

// state after `k` literals starting from state s (LzmaDec.c: state < 4 -> 0, < 10 -> s-3, else s-6)
__device__ __forceinline__ uint32_t lz_lit_advance(uint32_t s, uint32_t k)
{
    for (uint32_t i = 0; i < 3u && i < k; i++) s = s < 4u ? 0u : (s < 10u ? s - 3u : s - 6u);
    return k >= 3u ? 0u : s;
}

// length coder entries (len >= 2).  base = LZP_LEN or LZP_REPLEN
__device__ __forceinline__ uint32_t lz_gen_len(uint32_t* e, uint32_t base, uint32_t len, uint32_t posState)
{
    uint32_t n = 0, v = len - 2u;
    if (v < 8u) {
        e[n++] = ENT(base + LZL_CHOICE, 0);
        const uint32_t tb = base + LZL_LOW + posState * 8u; uint32_t m = 1;
        for (int i = 2; i >= 0; i--) { const uint32_t bit = (v >> i) & 1u; e[n++] = ENT(tb + m, bit); m = (m << 1) | bit; }
    } else if (v < 16u) {
        e[n++] = ENT(base + LZL_CHOICE, 1); e[n++] = ENT(base + LZL_CHOICE2, 0);
        v -= 8u;
        const uint32_t tb = base + LZL_MID + posState * 8u; uint32_t m = 1;
        for (int i = 2; i >= 0; i--) { const uint32_t bit = (v >> i) & 1u; e[n++] = ENT(tb + m, bit); m = (m << 1) | bit; }
    } else {
        e[n++] = ENT(base + LZL_CHOICE, 1); e[n++] = ENT(base + LZL_CHOICE2, 1);
        v -= 16u;
        const uint32_t tb = base + LZL_HIGH; uint32_t m = 1;
        for (int i = 7; i >= 0; i--) { const uint32_t bit = (v >> i) & 1u; e[n++] = ENT(tb + m, bit); m = (m << 1) | bit; }
    }
    return n;
}

// one match-type symbol.  kind: 0 normal match, 1 rep0 (long), 2 rep1.  dist = distance-1.  Returns entry count (<= 24).
__device__ __forceinline__ uint32_t lz_gen_match(uint32_t* e, uint32_t kind, uint32_t len, uint32_t dist, uint32_t state, uint32_t posState)
{
    uint32_t n = 0;
    e[n++] = ENT(LZP_ISMATCH + state * 4u + posState, 1);
    if (kind == 0u) {
        e[n++] = ENT(LZP_ISREP + state, 0);
        n += lz_gen_len(e + n, LZP_LEN, len, posState);
        const uint32_t lenState = len - 2u < 3u ? len - 2u : 3u;
        uint32_t slot;
        if (dist < 4u) slot = dist;
        else { const uint32_t hb = gc_hibit32(dist); slot = 2u * hb + ((dist >> (hb - 1u)) & 1u); }
        { const uint32_t tb = LZP_POSSLOT + lenState * 64u; uint32_t m = 1;
          for (int i = 5; i >= 0; i--) { const uint32_t bit = (slot >> i) & 1u; e[n++] = ENT(tb + m, bit); m = (m << 1) | bit; } }
        if (slot >= 4u) {
            const uint32_t footer = (slot >> 1) - 1u, base = (2u | (slot & 1u)) << footer, red = dist - base;
            if (slot < 14u) {
                const uint32_t tb = LZP_SPECPOS + (slot - 4u) * 32u; uint32_t m = 1;
                for (uint32_t i = 0; i < footer; i++) { const uint32_t bit = (red >> i) & 1u; e[n++] = ENT(tb + m, bit); m = (m << 1) | bit; }
            } else {
                e[n++] = 0x80000000u | ((footer - 4u) << 26) | (red >> 4);
                uint32_t m = 1;
                for (uint32_t i = 0; i < 4u; i++) { const uint32_t bit = (red >> i) & 1u; e[n++] = ENT(LZP_ALIGN + m, bit); m = (m << 1) | bit; }
            }
        }
    } else {
        e[n++] = ENT(LZP_ISREP + state, 1);
        if (kind == 1u) { e[n++] = ENT(LZP_ISREPG0 + state, 0); e[n++] = ENT(LZP_ISREP0LONG + state * 4u + posState, 1); }
        else { e[n++] = ENT(LZP_ISREPG0 + state, 1); e[n++] = ENT(LZP_ISREPG1 + state, 0); }
        n += lz_gen_len(e + n, LZP_REPLEN, len, posState);
    }
    return n;
}

// one literal: isMatch=0 + 8 tree bits (plain, or "matched" after a match: LzmaDec.c MATCHED_LITER_DEC)
__device__ __forceinline__ void lz_gen_literal(uint32_t* e, uint32_t cur, uint32_t prev, uint32_t matchByte, uint32_t state, uint32_t posState)
{
    e[0] = ENT(LZP_ISMATCH + state * 4u + posState, 0);
    const uint32_t pb = LZP_LITERAL + 0x300u * (prev >> (8u - GC_LZMA_LC));
    if (state < 7u) {
        uint32_t m = 1;
        for (int i = 7; i >= 0; i--) { const uint32_t bit = (cur >> i) & 1u; e[8 - i] = ENT(pb + m, bit); m = (m << 1) | bit; }
    } else {
        uint32_t offs = 0x100u, sym = cur | 0x100u, mb = matchByte;
        for (int i = 0; i < 8; i++) {
            mb <<= 1;
            e[1 + i] = ENT(pb + offs + (mb & offs) + (sym >> 8), (sym >> 7) & 1u);
            sym <<= 1;
            offs &= ~(mb ^ sym);
        }
    }
}

struct LzItem { uint32_t pos, len, off; };
// item k of the chunk = M[first + k] clipped to [cs, ce)
__device__ __forceinline__ LzItem lz_item(const uint64_t* M, uint32_t idx, uint32_t cs, uint32_t ce)
{
    const uint64_t m = M[idx];
    uint32_t pos = (uint32_t)(m & 0x1FFFFu), len = (uint32_t)((m >> 17) & 0x1FFFFu), off = (uint32_t)(m >> 34);
    uint32_t end = pos + len;
    if (pos < cs) pos = cs;
    if (end > ce) end = ce;
    LzItem it; it.pos = pos; it.len = end - pos; it.off = off;
    return it;
}

extern "C" __global__ void __launch_bounds__(64)
gc_lzma2_model_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, const uint64_t* __restrict__ Mall,
                      const uint32_t* __restrict__ nM, uint32_t chunkLog, uint16_t* __restrict__ stream,
                      GcLzmaChunkInfo* __restrict__ cinfo)
{
    __shared__ uint16_t P[LZP_TOTAL];
    __shared__ uint32_t sPiece[24]; __shared__ uint32_t sPieceN;
    __shared__ uint32_t sMat[64u * 24u];
    __shared__ uint32_t sMatN[64];

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t chunk = blockIdx.x;
    const uint32_t chunkSize = 1u << chunkLog, perBlock = GC_ZSTD_BLOCK_MAX >> chunkLog;
    const uint32_t b = chunk / perBlock, cInB = chunk % perBlock;
    const uint64_t blockBase = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const uint32_t cs = cInB * chunkSize;
    if (cs >= blockLen) { if (lane == 0) { cinfo[chunk].usize = 0; cinfo[chunk].csize = 0; cinfo[chunk].nWords = 0; } return; }
    const uint32_t ce = cs + chunkSize < blockLen ? cs + chunkSize : blockLen;
    const uint8_t* S = src + blockBase;                   // block-relative addressing; S[-1] exists iff blockBase > 0
    const uint64_t* M = Mall + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    const uint32_t n = nM[b];

    for (uint32_t i = lane; i < LZP_TOTAL; i += 64u) P[i] = 1024u;

    // items of this chunk: [first, last) in M (sorted by position), clipped to the chunk
    uint32_t first, last;
    { uint32_t lo = 0, hi = n;                            // first item that ends after cs
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; const uint64_t m = M[mid]; if ((uint32_t)(m & 0x1FFFFu) + (uint32_t)((m >> 17) & 0x1FFFFu) > cs) hi = mid; else lo = mid + 1u; }
      first = lo;
      lo = first; hi = n;                                 // first item that starts at or after ce
      while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if ((uint32_t)(M[mid] & 0x1FFFFu) >= ce) hi = mid; else lo = mid + 1u; }
      last = lo; }
    // a clipped piece shorter than 2 bytes cannot be a match: it can only be the first or the last item
    if (first < last && lz_item(M, first, cs, ce).len < 2u) first++;
    if (first < last && lz_item(M, last - 1u, cs, ce).len < 2u) last--;
    first = gc_uniform(first); last = gc_uniform(last);     // keep the walk below on the scalar unit

    LzStream strm; strm.w = stream + (uint64_t)chunk * GC_LZMA_STREAM_WORDS(chunkLog); strm.pos = 0;
    gc_wave_sync();

    uint32_t cursor = cs;             // next position to encode (uniform)
    uint32_t exitState = 0;           // coder state after the previous item (uniform); 0 at chunk start
    uint32_t prevOff = 1u;            // distance of the previous item (rep0), 1 after a state reset
    uint32_t carryRun = 0;            // (index + 1 - first) of the start of the run of equal offsets containing the previous item; 0 = virtual

    for (uint32_t base = first; base <= last; base += 64u) {
        const uint32_t cnt = last - base < 64u ? last - base : 64u;        // may be 0 on the final pass (tail literals only)
        // ---- per-item parallel part
        LzItem it; it.pos = ce; it.len = 0; it.off = 0;
        if (lane < cnt) it = lz_item(M, base + lane, cs, ce);
        // end of the previous item (cursor for lane 0)
        uint32_t prevEnd = __shfl_up(it.pos + it.len, 1); if (lane == 0) prevEnd = cursor;
        uint32_t pOff = __shfl_up(it.off, 1); if (lane == 0) pOff = prevOff;
        const uint32_t ll = it.pos - prevEnd;
        // repeat-distance history as scans: rep0 = previous distance, rep1 = distance before the current run of equal ones
        const uint32_t k = base - first + lane;                             // chunk-relative item index
        const uint32_t v = (lane < cnt && it.off != pOff) ? k + 1u : 0u;
        uint32_t runIncl = gc_wave_incl_max(v); if (runIncl < carryRun) runIncl = carryRun;   // run start (+1) of the run containing item k
        uint32_t runPrev = __shfl_up(runIncl, 1); if (lane == 0) runPrev = carryRun;          // ... containing item k-1
        uint32_t rep1 = 1u;
        if (lane < cnt && runPrev >= 2u) rep1 = (uint32_t)(M[first + runPrev - 2u] >> 34);
        uint32_t kind = 0;
        if (lane < cnt) kind = it.off == pOff ? 1u : (it.off == rep1 ? 2u : 0u);
        // coder state: the state after an item depends only on its kind and on whether a literal preceded it
        const bool split = it.len > 273u;
        const bool litBefore = ll > 0u || (k == 0u && cursor == cs && base == first);
        const uint32_t stAfterFirst = kind == 0u ? (litBefore ? 7u : 10u) : (litBefore ? 8u : 11u);
        const uint32_t myExit = split ? 11u : stAfterFirst;                  // a split match ends with a rep0 continuation piece
        uint32_t sPrev = __shfl_up(myExit, 1); if (lane == 0) sPrev = exitState;
        const uint32_t sBefore = lz_lit_advance(sPrev, ll);
        uint32_t firstLen = it.len;
        if (split) { firstLen = 273u; if ((it.len - 273u) % 273u == 1u) firstLen = 272u; }
        uint32_t nEnt = 0;
        if (lane < cnt) nEnt = lz_gen_match(&sMat[lane * 24u], kind, firstLen, it.off - 1u, sBefore, it.pos & 3u);
        sMatN[lane] = nEnt;
        gc_wave_sync();

        // ---- serial walk over the items of this batch (uniform loop), literal runs produced 64 at a time
        const uint32_t steps = cnt + ((base + cnt >= last) ? 1u : 0u);      // the final pass also flushes the tail literals
        for (uint32_t j = 0; j < steps; j++) {
            const bool isTail = j >= cnt;
            const uint32_t jj = isTail ? 0u : j;
            const uint32_t ipos = isTail ? ce : gc_readlane(it.pos, jj);
            const uint32_t ilen = isTail ? 0u : gc_readlane(it.len, jj);
            const uint32_t ioff = isTail ? 0u : gc_readlane(it.off, jj);
            const uint32_t iflen = isTail ? 0u : gc_readlane(firstLen, jj);
            const uint32_t iexit = isTail ? 0u : gc_readlane(myExit, jj);
            const uint32_t sAfterPrev = exitState;
            // literals [cursor, ipos): 7 literals x 9 entries per group, lane = (literal, tree depth), entries in closed form
            for (uint32_t lp = cursor; lp < ipos; lp += 7u) {
                const uint32_t cntL = ipos - lp < 7u ? ipos - lp : 7u;
                const uint32_t li = lane / 9u, d = lane - li * 9u;          // literal in group, entry 0 = isMatch, 1..8 = tree depth
                uint32_t e = 0;
                if (li < cntL) {
                    const uint32_t p = lp + li, i = p - cursor;
                    const uint32_t st = lz_lit_advance(sAfterPrev, i);
                    const uint32_t cur = S[p];
                    if (d == 0u) e = ENT(LZP_ISMATCH + st * 4u + (p & 3u), 0);
                    else {
                        const uint32_t prev = (blockBase + p) ? S[(int64_t)p - 1] : 0u;
                        const uint32_t pb = LZP_LITERAL + 0x300u * (prev >> (8u - GC_LZMA_LC));
                        const uint32_t m = (0x100u | cur) >> (9u - d), bit = (cur >> (8u - d)) & 1u;
                        uint32_t idx = pb + m;
                        if (st >= 7u) {                                      // matched literal (LzmaDec.c MATCHED_LITER_DEC)
                            const uint32_t mb = S[(int64_t)p - (int64_t)prevOff];
                            // still "matching" at depth d iff the d-1 bits above agree; then the node is selected by the match bit too
                            const bool on = ((cur ^ mb) >> (9u - d)) == 0u;
                            if (on) idx += 0x100u + (((mb >> (8u - d)) & 1u) << 8);
                        }
                        e = ENT(idx, bit);
                    }
                }
                lz_emit_group(strm, e, cntL * 9u, 9u, P, lane);
            }
            if (isTail) { cursor = ce; break; }
            // the match itself (first piece pre-generated by its lane), then continuation pieces of very long matches
            { const uint32_t ne = gc_uniform(sMatN[jj]); lz_emit_group(strm, lane < ne ? sMat[jj * 24u + lane] : 0u, ne, 24u, P, lane); }
            if (ilen > iflen) {
                // state after the first piece (match 7/10, rep 8/11); every further piece is a rep0 coded from a state >= 7
                uint32_t done = iflen, st = gc_readlane(stAfterFirst, jj);
                while (done < ilen) {
                    uint32_t piece = ilen - done < 273u ? ilen - done : 273u;
                    if (ilen - done - piece == 1u) piece--;
                    if (lane == 0) sPieceN = lz_gen_match(sPiece, 1u, piece, ioff - 1u, st, (ipos + done) & 3u);
                    gc_wave_sync();
                    { const uint32_t ne = gc_uniform(sPieceN); lz_emit_group(strm, lane < ne ? sPiece[lane] : 0u, ne, 24u, P, lane); }
                    gc_wave_sync();
                    st = 11u; done += piece;
                }
            }
            cursor = ipos + ilen; exitState = iexit; prevOff = ioff;
        }
        { const uint32_t r = gc_readlane(runIncl, 63u); if (cnt) carryRun = r; }
        gc_wave_sync();
    }

    if (lane == 0) { GcLzmaChunkInfo ci; ci.usize = ce - cs; ci.csize = 0; ci.nWords = strm.pos; cinfo[chunk] = ci; }
}

// ---------------------------------------------------------------------------------------------- L3: range coder
// One chunk per LANE.  With the probabilities already resolved, what is left of LZMA's serial dependency is the range
// recurrence itself -- bound = (range >> 11) * p; range = bit ? range - bound : bound; renormalise -- about two dozen
// register-only instructions per coded bit (C/fast-lzma2/range_enc.h:62-152, RC_shiftLow range_enc.c:123-140), so 64
// chunks advance per wave instruction and the latency of one chunk's chain is shared 64 ways.
struct LzRc { uint64_t low; uint32_t range; uint32_t cache; uint32_t cacheSize; uint32_t outPos; uint32_t outCap; uint8_t* out; };

__device__ __forceinline__ void rc_shift_low(LzRc& rc)
{
    if ((uint32_t)rc.low < 0xFF000000u || (rc.low >> 32) != 0) {
        const uint32_t carry = (uint32_t)(rc.low >> 32);
        uint32_t c = rc.cache;
        do {
            if (rc.outPos < rc.outCap) rc.out[rc.outPos] = (uint8_t)(c + carry);
            rc.outPos++;
            c = 0xFFu;
        } while (--rc.cacheSize != 0);
        rc.cache = ((uint32_t)rc.low >> 24) & 0xFFu;
    }
    rc.cacheSize++;
    rc.low = (rc.low & 0x00FFFFFFull) << 8;
}

__device__ __forceinline__ void rc_word(LzRc& rc, uint32_t w)
{
    const bool bit = (w & LZW_BIT) != 0u;
    if (w & LZW_DIRECT) { rc.range >>= 1; if (bit) rc.low += rc.range; }
    else {
        const uint32_t bound = (rc.range >> 11) * (w & 0x7FFu);
        if (bit) { rc.low += bound; rc.range -= bound; } else rc.range = bound;
    }
    if (rc.range < (1u << 24)) { rc.range <<= 8; rc_shift_low(rc); }
}

extern "C" __global__ void __launch_bounds__(64)
gc_lzma2_rc_kernel(const uint16_t* __restrict__ stream, uint32_t chunkLog, uint32_t nChunks, uint8_t* __restrict__ chunkOut,
                   GcLzmaChunkInfo* __restrict__ cinfo)
{
    const uint32_t chunk = blockIdx.x * 64u + threadIdx.x;
    if (chunk >= nChunks) return;
    const GcLzmaChunkInfo ci = cinfo[chunk];
    if (ci.usize == 0u) return;
    const uint16_t* W = stream + (uint64_t)chunk * GC_LZMA_STREAM_WORDS(chunkLog);     // 16-byte aligned
    LzRc rc; rc.low = 0; rc.range = 0xFFFFFFFFu; rc.cache = 0; rc.cacheSize = 1; rc.outPos = 0;
    rc.outCap = ci.usize;                                  // a chunk that does not shrink is stored raw anyway
    rc.out = chunkOut + ((uint64_t)chunk << chunkLog);
    const uint32_t n = ci.nWords, nVec = n >> 3;
    const GcU4* V = (const GcU4*)W;
    GcU4 nxt; nxt.x = nxt.y = nxt.z = nxt.w = 0; if (nVec) nxt = V[0];
    for (uint32_t i = 0; i < nVec; i++) {                  // 8 words per 16-byte load, next load in flight while these are coded
        const GcU4 cur = nxt;
        if (i + 1u < nVec) nxt = V[i + 1u];
        rc_word(rc, cur.x & 0xFFFFu); rc_word(rc, cur.x >> 16); rc_word(rc, cur.y & 0xFFFFu); rc_word(rc, cur.y >> 16);
        rc_word(rc, cur.z & 0xFFFFu); rc_word(rc, cur.z >> 16); rc_word(rc, cur.w & 0xFFFFu); rc_word(rc, cur.w >> 16);
    }
    for (uint32_t k = nVec << 3; k < n; k++) rc_word(rc, W[k]);
    for (int i = 0; i < 5; i++) rc_shift_low(rc);          // RC_flush
    cinfo[chunk].csize = rc.outPos < rc.outCap ? rc.outPos : 0xFFFFFFFFu;      // 0xFFFFFFFF: store raw
}
