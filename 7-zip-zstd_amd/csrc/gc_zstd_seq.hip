// gc_zstd_seq.hip -- K3: sequences section of one zstd block per workgroup (256 threads).
//
// Replaces ZSTD_seqToCodes (C/zstd/zstd_compress.c:2693), ZSTD_updateRep (zstd_compress_internal.h:818),
// ZSTD_buildSequencesStatistics (zstd_compress.c:2763: histograms, ZSTD_selectEncodingType, ZSTD_buildCTable ->
// FSE_normalizeCount / FSE_writeNCount / FSE_buildCTable_wksp) and ZSTD_encodeSequences_body
// (zstd_compress_sequences.c:291-383).
//
// Stages (S = serial per lane, P = parallel over sequences):
//   P  merge   chains of capped matches (same offset, litLength 0) collapse into one sequence
//   P  repcode repeat-offset history as a scan: rep1 = previous offset, rep2 = the offset before the current run
//              of equal offsets.  Only codes whose meaning depends on (rep1, rep2) are emitted, so the decoder's
//              third history slot never matters and the whole assignment is order-independent
//   P  codes   LL/ML/OF code + extra bits, three LDS histograms
//   S  tables  one wave per table: mode (predefined / RLE / FSE), normalisation, NCount header, CTable
//   S  chains  one wave per table walks its FSE state backwards over all sequences: 64 segments at once, each lane finding
//              its start state by running a few symbols ahead of its segment (FSE states forget their origin), checked and
//              repaired against the predecessor's final state, so the result is exactly the serial walk
//   P  pack    per-sequence bit counts -> block prefix sums -> fields OR-ed into an LDS tile -> bytes stream out
//
// Bit order is normative (zstd_compress_sequences.c:311-376): last sequence first; per sequence OF-state,
// ML-state, LL-state, LL extra, ML extra, OF extra; then final states ML, OF, LL and a closing 1 bit.
#include "gc_common.h"
#include "gc_device.h"
#include "gc_fse.h"
#ifdef HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif

#define SEQ_T GC_SEQ_T
#define SEQ_TILE_WORDS ((SEQ_T * 80u) / 32u + 8u)
#define SEQ_CHAIN_TILE 4096u   // sequences per state-chain tile
#define SEQ_CHAIN_SEG  64u     // sequences per lane and tile
#define SEQ_TC(u) ((u) + ((u) >> 6))                     // index of tile element u in tCode (row stride 65 bytes)
#define SEQ_TO(u) ((u) + 2u * ((u) >> 6))                // ... in tOut (row stride 66 half-words = 33 banks)
#ifndef SEQ_WARM_SEGS
#define SEQ_WARM_SEGS  4u      // a lane looks this many segments back for a point where all state walks meet
#endif

__constant__ uint8_t kLLCode[64] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
                                     22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 };
__constant__ uint8_t kMLCode[128] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
                                      32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
                                      40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
                                      42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42 };
__constant__ uint8_t kLLBits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
__constant__ uint8_t kMLBits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
__constant__ int16_t kLLDefNorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
__constant__ int16_t kMLDefNorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
__constant__ int16_t kOFDefNorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

__device__ __forceinline__ uint32_t seq_ll_code(uint32_t ll) { return ll > 63u ? gc_hibit32(ll) + 19u : kLLCode[ll]; }
__device__ __forceinline__ uint32_t seq_ml_code(uint32_t mlBase) { return mlBase > 127u ? gc_hibit32(mlBase) + 36u : kMLCode[mlBase]; }

struct SeqTab {              // one FSE table in LDS
    uint16_t state[512];
    GcFseSym tt[64];
    int16_t  norm[64];
    uint32_t count[64];
    uint8_t  spread[512];
    uint16_t cumul[66];
    uint8_t  desc[96];       // table description bytes for the section header
    uint32_t descSize, mode, tableLog, maxSym, finalState;
    uint32_t tabMaxSym;      // last symbol described by norm[] (predefined: whole default table)
    // state-chain tile scratch (see "chains" in the kernel)
    // a lane walks ITS segment, i.e. the lanes of a wave touch elements SEQ_CHAIN_SEG apart: with a row length of exactly 64 all of them would
    // fall into one or two LDS banks (a 32-way conflict per access); one element of padding per segment spreads them over all banks
    uint8_t  tCode[SEQ_CHAIN_TILE + SEQ_CHAIN_TILE / SEQ_CHAIN_SEG + 3u];
    uint16_t tOut[SEQ_CHAIN_TILE + 2u * (SEQ_CHAIN_TILE / SEQ_CHAIN_SEG)];
    uint64_t meet[SEQ_CHAIN_TILE / 64u];
};

// block-wide exclusive sum scan (SEQ_T threads)
__device__ __forceinline__ uint32_t seq_excl_scan(uint32_t v, uint32_t* sWave, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = gc_wave_incl_sum(v);
    if (lane == 63u) sWave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < SEQ_T / 64u; w++) { uint32_t c = sWave[w]; if (w < wave) before += c; all += c; }
    __syncthreads();
    *total = all;
    return before + incl - v;
}
// block-wide inclusive max scan
__device__ __forceinline__ uint32_t seq_incl_maxscan(uint32_t v, uint32_t* sWave, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = gc_wave_incl_max(v);
    if (lane == 63u) sWave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < SEQ_T / 64u; w++) { uint32_t c = sWave[w]; if (w < wave) before = max(before, c); all = max(all, c); }
    __syncthreads();
    *total = all;
    return max(before, incl);
}

__device__ __forceinline__ void seq_or_bits(uint32_t* buf, uint32_t bitoff, uint64_t v, uint32_t nbits)
{
    if (nbits == 0) return;
    uint32_t word = bitoff >> 5, sh = bitoff & 31u;
    uint32_t w0 = (uint32_t)(v << sh);
    uint64_t rest = sh ? (v >> (32u - sh)) : (v >> 32);
    if (w0) atomicOr(&buf[word], w0);
    if ((uint32_t)rest) atomicOr(&buf[word + 1], (uint32_t)rest);
    if ((uint32_t)(rest >> 32)) atomicOr(&buf[word + 2], (uint32_t)(rest >> 32));
}

// Build one table (single lane).  which: 0 LL, 1 OF, 2 ML.
__device__ inline void seq_build_table(SeqTab& T, int which, uint32_t nbSeq)
{
    const uint32_t maxLog = which == 1 ? 8u : 9u, defLog = which == 1 ? 5u : 6u;
    const int16_t* defNorm = which == 0 ? kLLDefNorm : (which == 1 ? kOFDefNorm : kMLDefNorm);
    const uint32_t defMax = which == 0 ? 35u : (which == 1 ? 28u : 52u);
    const uint32_t alpha = which == 0 ? 36u : (which == 1 ? 32u : 53u);
    uint32_t maxSym = 0, maxCount = 0, present = 0;
    for (uint32_t s = 0; s < alpha; s++) if (T.count[s]) { maxSym = s; present++; if (T.count[s] > maxCount) maxCount = T.count[s]; }
    T.maxSym = maxSym;
    if (maxCount == nbSeq) {                                   // one code only -> RLE table (1 byte)
        uint32_t sym = maxSym;
        T.mode = 1; T.tableLog = 0; T.descSize = 1; T.desc[0] = (uint8_t)sym; T.tabMaxSym = sym;
        T.state[0] = 0; T.tt[sym].deltaNbBits = 0; T.tt[sym].deltaFindState = 0;
        return;
    }
    // candidate: FSE-compressed table
    uint32_t tl = gc_hibit32(nbSeq - 1u) >= 2u ? gc_hibit32(nbSeq - 1u) - 2u : 0u;     // FSE_optimalTableLog flavour
    { uint32_t minBits = min(gc_hibit32(nbSeq) + 1u, gc_hibit32(maxSym) + 2u); if (tl < minBits) tl = minBits; }
    while ((1u << tl) < present) tl++;
    if (tl < 5u) tl = 5u;
    if (tl > maxLog) tl = maxLog;
    gc_fse_normalize(T.count, maxSym, nbSeq, tl, T.norm);
    // Rare symbols are demoted to a single cell: whatever the state, a count-1 symbol sends it to the same successor, and these
    // are the only points where two walks that started from different states are guaranteed to meet (an FSE step is a monotone
    // map of the state, so walks otherwise keep their distance) -- they are what lets the state chain below be walked in
    // parallel segments.  Cost: < 0.1 bit per sequence on the corpora measured.
    {
        uint32_t freed = 0, big = 0; int bigv = 0;
        for (uint32_t s = 0; s <= maxSym; s++) {
            if (T.norm[s] == 2 || T.norm[s] == 3) { freed += (uint32_t)T.norm[s] - 1u; T.norm[s] = 1; }
            if (T.norm[s] > bigv) { bigv = T.norm[s]; big = s; }
        }
        T.norm[big] = (int16_t)(T.norm[big] + (int)freed);
    }
    // cost comparison in 1/256 bits (free choice; reference: ZSTD_selectEncodingType, zstd_compress_sequences.c:157)
    uint64_t costFse = 0, costDef = 0; bool defOk = maxSym <= defMax;
    for (uint32_t s = 0; s <= maxSym; s++) {
        uint32_t c = T.count[s];
        if (!c) continue;
        costFse += (uint64_t)c * ((tl << 8) - gc_log2_q8((uint32_t)T.norm[s]));
        if (defOk) { int dn = defNorm[s]; costDef += (uint64_t)c * ((defLog << 8) - gc_log2_q8(dn < 1 ? 1u : (uint32_t)dn)); }
    }
    uint32_t descBytes = gc_fse_write_ncount(T.desc, T.norm, maxSym, tl);
    costFse += (uint64_t)descBytes * 8u * 256u;
    if (defOk && costDef <= costFse) {
        T.mode = 0; T.tableLog = defLog; T.descSize = 0; T.tabMaxSym = defMax;
        for (uint32_t s = 0; s <= defMax; s++) T.norm[s] = defNorm[s];
        gc_fse_build_ctable(T.norm, defMax, defLog, T.state, T.tt, T.spread, T.cumul);
    } else {
        T.mode = 2; T.tableLog = tl; T.descSize = descBytes; T.tabMaxSym = maxSym;
        gc_fse_build_ctable(T.norm, maxSym, tl, T.state, T.tt, T.spread, T.cumul);
    }
}

extern "C" __global__ void __launch_bounds__(SEQ_T)
gc_zstd_seq_kernel(const GcSeqRaw* __restrict__ seqRaw, const GcBlockMeta* __restrict__ meta,
                   uint64_t* __restrict__ seqPacked,      // scratch: GC_MAX_SEQ_PER_BLOCK per block
                   uint32_t* __restrict__ seqOff,         // scratch: GC_MAX_SEQ_PER_BLOCK per block (real offsets)
                   uint8_t* __restrict__ codes,           // scratch: 3 * GC_MAX_SEQ_PER_BLOCK per block (LL, OF, ML)
                   uint16_t* __restrict__ stOut,          // scratch: 3 * GC_MAX_SEQ_PER_BLOCK per block
                   uint8_t* __restrict__ seqSec, GcSectionInfo* __restrict__ info, uint64_t srcSize,
                   uint32_t frameBlocks,                  // blocks per zstd frame: repeat offsets carry over inside a frame
                   unsigned long long* __restrict__ prof /* optional per-phase cycle sums */)
{
    __shared__ SeqTab sTab[3];
    __shared__ uint32_t sWave[SEQ_T / 64u];
    __shared__ uint32_t sRun[SEQ_T];
    __shared__ uint32_t sTile[SEQ_TILE_WORDS];

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, b = blockIdx.x;
    const uint32_t nRaw = meta[b].nSeqRaw;
    const uint64_t blockBase = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const GcSeqRaw* R = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint64_t* P = seqPacked + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint32_t* O = seqOff + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* cLL = codes + (uint64_t)b * 3u * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* cOF = cLL + GC_MAX_SEQ_PER_BLOCK;
    uint8_t* cML = cOF + GC_MAX_SEQ_PER_BLOCK;
    uint16_t* stLL = stOut + (uint64_t)b * 3u * GC_MAX_SEQ_PER_BLOCK;
    uint16_t* stOF = stLL + GC_MAX_SEQ_PER_BLOCK;
    uint16_t* stML = stOF + GC_MAX_SEQ_PER_BLOCK;
    uint8_t* out = seqSec + (uint64_t)b * GC_SEQSEC_STRIDE;

    if (nRaw == 0) { if (t == 0) { out[0] = 0; info[b].seqSecSize = 1; info[b].nSeq = 0; } return; }

    for (uint32_t i = t; i < 3u * 64u; i += SEQ_T) sTab[i >> 6].count[i & 63u] = 0;
    unsigned long long tprev = prof ? gc_clock() : 0ull;
#define SEQ_PHASE(i) do { if (prof && t == 0) { unsigned long long now_ = gc_clock(); atomicAdd(&prof[i], now_ - tprev); tprev = now_; } } while (0)
    unsigned long long tsub = 0;
#define SEQ_SUB(i) do { if (prof && t == 0) { unsigned long long now_ = gc_clock(); if (tsub) atomicAdd(&prof[i], now_ - tsub); tsub = now_; } } while (0)

    // ---- S1: merge chains of capped matches.  Head = first record of a run with equal offset and litLength 0.
    //      P[j] = ll(17) | ml(18)<<17 for merged sequence j, O[j] = its offset
    uint32_t nSeq = 0;
    for (uint32_t tb = 0; tb < nRaw; tb += SEQ_T) {
        const uint32_t i = tb + t;
        uint32_t head = 0, ll = 0, off = 0, ml = 0;
        if (i < nRaw) {
            GcSeqRaw r = R[i];
            uint32_t prevRank = 0, prevOff = 0;
            if (i) { GcSeqRaw q = R[i - 1u]; prevRank = q.litRank; prevOff = q.offml >> 8; }
            ll = r.litRank - prevRank; off = r.offml >> 8; ml = r.offml & 0xFFu;
            head = (i == 0u || ll != 0u || off != prevOff) ? 1u : 0u;
            if (head) {
                const uint32_t rank = r.litRank;
                for (uint32_t k = i + 1u; k < nRaw; k++) {            // absorb the continuation records
                    GcSeqRaw c = R[k];
                    if (c.litRank != rank || (c.offml >> 8) != off) break;
                    ml += c.offml & 0xFFu;
                }
            }
        }
        uint32_t tot;
        const uint32_t j = nSeq + seq_excl_scan(head, sWave, &tot);
        if (head) { P[j] = (uint64_t)ll | ((uint64_t)ml << 17); O[j] = off; }
        nSeq += tot;
    }
    __syncthreads();      // P/O written by this workgroup are read back by other lanes below
    SEQ_PHASE(0);         // merge

    // ---- S2: repeat offsets as scans, then codes + histograms
    //      virtual history: index -1 -> offset 1, -2 -> 4 (start state {1,4,8}, zstd_internal.h:65) in the first block of a
    //      frame.  In later blocks the decoder arrives with the history the previous block left behind; this block is coded
    //      without knowing it (blocks stay independent units of work): the virtual history is "unknown" (0, never equal to
    //      an offset), so no repeat code refers to it.  rep1 = previous offset and rep2 = offset before the current run hold
    //      for the decoder whatever the unknown part was, because every code emitted below moves exactly those two slots.
    const bool firstInFrame = (b % frameBlocks) == 0u;
    const uint32_t virt1 = firstInFrame ? 1u : 0u, virt2 = firstInFrame ? 4u : 0u;
    uint32_t carryRun = 1u;      // (run start index + 2) of the run containing the last sequence of the previous tile
    for (uint32_t tb = 0; tb < nSeq; tb += SEQ_T) {
        const uint32_t j = tb + t;
        uint32_t off = 0, prevOff = virt1;
        if (j < nSeq) { off = O[j]; prevOff = j ? O[j - 1u] : virt1; }
        const uint32_t v = (j < nSeq && off != prevOff) ? j + 2u : 0u;
        uint32_t tot;
        const uint32_t runIncl = max(seq_incl_maxscan(v, sWave, &tot), carryRun);  // run start (+2) of the run containing j
        sRun[t] = runIncl;
        __syncthreads();
        const uint32_t runPrev = t ? sRun[t - 1u] : carryRun;                      // ... of the run containing j-1
        if (j < nSeq) {
            const uint64_t pk = P[j];
            const uint32_t ll = (uint32_t)(pk & 0x1FFFFu), ml = (uint32_t)((pk >> 17) & 0x3FFFFu);
            const uint32_t rep1 = prevOff;
            const int32_t before = (int32_t)runPrev - 3;             // index whose offset is rep2 (>= -2)
            const uint32_t rep2 = before >= 0 ? O[before] : (before == -1 ? virt1 : virt2);
            uint32_t offBase;
            if (ll != 0u) offBase = (off == rep1) ? 1u : ((off == rep2) ? 2u : off + 3u);
            else offBase = (off == rep2) ? 1u : ((rep1 > 1u && off == rep1 - 1u) ? 3u : off + 3u);
            const uint32_t mlBase = ml - 3u;
            const uint32_t llc = seq_ll_code(ll), mlc = seq_ml_code(mlBase), ofc = gc_hibit32(offBase);
            cLL[j] = (uint8_t)llc; cOF[j] = (uint8_t)ofc; cML[j] = (uint8_t)mlc;
            atomicAdd(&sTab[0].count[llc], 1u); atomicAdd(&sTab[1].count[ofc], 1u); atomicAdd(&sTab[2].count[mlc], 1u);
            P[j] = (uint64_t)ll | ((uint64_t)mlBase << 17) | ((uint64_t)offBase << 35);   // ll, mlBase, offBase
#ifdef HIPEMU
            if (getenv("GC_TRACE")) fprintf(stderr, "E %u ofv=%u ml=%u ll=%u off=%u rep1=%u rep2=%u\n", j, offBase, ml, ll, off, rep1, rep2);
#endif
        }
        carryRun = max(carryRun, tot);
        __syncthreads();
    }

    SEQ_PHASE(1);         // repcodes + codes + histograms
    // ---- tables: one lane per table
    if (wave < 3u && lane == 0u) seq_build_table(sTab[wave], (int)wave, nSeq);
    __syncthreads();
    SEQ_PHASE(2);         // tables

    // ---- chains: wave w walks table w's FSE state backwards (last sequence first = processing index v 0).
    //      The walk is serial through the state, but walks that started from different states meet at the next symbol whose
    //      normalised count is 1 (every state has the same successor there; the table builder above makes sure such symbols
    //      exist).  So the 64 lanes of the wave walk 64 consecutive segments of 64 sequences at once: a lane first runs
    //      ahead of its segment from the nearest such meeting point before it (no output), then its own segment.  Afterwards every
    //      segment's start state is compared with its predecessor's final state; a segment that started wrong is walked again
    //      from the right state, only until it rejoins the states already stored, and this repeats until nothing changes
    //      (each round settles at least one more segment, so the result is exactly the serial walk).
    if (wave < 3u) {
        SeqTab& T = sTab[wave];
        const uint8_t* C = wave == 0u ? cLL : (wave == 1u ? cOF : cML);
        uint16_t* S = wave == 0u ? stLL : (wave == 1u ? stOF : stML);
        if (T.mode == 1u) {                                   // RLE table: no state bits at all
            for (uint32_t j = lane; j < nSeq; j += 64u) S[j] = 0;
            if (lane == 0u) T.finalState = 0;
        } else {
            const uint32_t L = T.tableLog;
            int nv = 0;
            if (lane <= T.tabMaxSym) nv = T.norm[lane];
            const uint64_t resetMask = __ballot(nv == 1 || nv == -1);    // symbols that send every state to the same successor
            uint32_t carry = 0;                               // state after the last sequence of the previous tile
            for (uint32_t tb = 0; tb < nSeq; tb += SEQ_CHAIN_TILE) {
                const uint32_t tileLen = min(SEQ_CHAIN_TILE, nSeq - tb);
                if (prof && t == 0) tsub = gc_clock();
                // stage the codes (independent loads, eight in flight per lane), then per 64 sequences one mask of the positions
                // that hold a count-1 symbol ("meeting points")
#pragma unroll 8
                for (uint32_t u = lane; u < tileLen; u += 64u) T.tCode[SEQ_TC(u)] = C[nSeq - 1u - (tb + u)];
                gc_wave_sync();
                SEQ_SUB(5);
                for (uint32_t k = 0; k < tileLen; k += 64u) {
                    const uint32_t u = k + lane;
                    const bool isR = u < tileLen && ((resetMask >> T.tCode[SEQ_TC(u)]) & 1ull) != 0ull;
                    const uint64_t bal = __ballot(isR);
                    if (lane == 0u) T.meet[k >> 6] = bal;
                }
                gc_wave_sync();
                const uint32_t nSegs = (tileLen + SEQ_CHAIN_SEG - 1u) / SEQ_CHAIN_SEG;      // <= 64: lane i walks segment i
                const uint32_t u0 = lane * SEQ_CHAIN_SEG, u1 = min(u0 + SEQ_CHAIN_SEG, tileLen);
                uint32_t first = u0;                          // first index of the segment that emits bits
                uint32_t st0 = carry;
                if (lane < nSegs) {
                    if (tb + u0 == 0u) { st0 = gc_fse_init_state(T.state, T.tt[T.tCode[SEQ_TC(0)]]); T.tOut[SEQ_TO(0)] = 0; first = 1u; }   // FSE_initCState2: no bits
                    else if (lane != 0u) {
                        // run ahead from the nearest meeting point in the SEQ_WARM_SEGS segments before this one (from there the
                        // state is exact whatever it was before); if there is none, from SEQ_WARM_SEGS segments back (a guess)
                        uint32_t w = lane > SEQ_WARM_SEGS ? (lane - SEQ_WARM_SEGS) * SEQ_CHAIN_SEG : 0u;
                        for (uint32_t c = lane; c-- > 0u && c + SEQ_WARM_SEGS >= lane; ) {
                            const uint64_t m = T.meet[c];
                            if (m) { w = c * 64u + 63u - (uint32_t)__clzll((long long)m); break; }
                        }
                        if (tb + w == 0u) { st0 = gc_fse_init_state(T.state, T.tt[T.tCode[SEQ_TC(0)]]); w = 1u; }                   // exact, not a guess
                        else st0 = 1u << L;
                        if (w < u0) {                              // the symbol's table entry does not depend on the state: fetched one
                            GcFseSym sy = T.tt[T.tCode[SEQ_TC(w)]];          // step ahead, so a step costs one dependent LDS read, not three
                            for (; w < u0; w++) {
                                const GcFseSym nx = T.tt[T.tCode[SEQ_TC(w + 1u < u0 ? w + 1u : w)]];
                                const uint32_t nb = (st0 + sy.deltaNbBits) >> 16;
                                st0 = T.state[(st0 >> nb) + (uint32_t)sy.deltaFindState];
                                sy = nx;
                            }
                        }
                    }
                }
                SEQ_SUB(6);
                // tOut[u] = nbBits << 10 | state BEFORE symbol u (10 bits): the bits to emit are its low nbBits, and a repair walk
                // can tell when it has rejoined the trajectory already stored (same state at the same u: the rest is unchanged)
                uint32_t fin = st0;
                bool redo = lane < nSegs;
                bool firstPass = true;
#ifdef HIPEMU
                uint32_t dbgRounds = 0, dbgRedo = 0;
#endif
                for (;;) {
                    if (redo) {
                        uint32_t state = st0, u = first;
                        GcFseSym sy = T.tt[T.tCode[SEQ_TC(u < u1 ? u : u1 - 1u)]];
                        for (; u < u1; u++) {
                            if (!firstPass && (T.tOut[SEQ_TO(u)] & 0x3FFu) == (state & 0x3FFu)) break;
                            const GcFseSym nx = T.tt[T.tCode[SEQ_TC(u + 1u < u1 ? u + 1u : u)]];
                            const uint32_t nb = (state + sy.deltaNbBits) >> 16;
                            T.tOut[SEQ_TO(u)] = (uint16_t)((nb << 10) | (state & 0x3FFu));
                            state = T.state[(state >> nb) + (uint32_t)sy.deltaFindState];
                            sy = nx;
                        }
                        if (u == u1) fin = state;                 // walked to the end: the final state may have changed
                    }
                    firstPass = false;
                    const uint32_t prevFin = __shfl_up(fin, 1);
                    redo = lane != 0u && lane < nSegs && prevFin != st0;
                    if (redo) st0 = prevFin;
#ifdef HIPEMU
                    dbgRounds++; dbgRedo += (uint32_t)__popcll(__ballot(redo));
#endif
                    if (!__any(redo)) break;
                }
#ifdef HIPEMU
                if (getenv("GC_TRACE_CHAIN") && lane == 0) fprintf(stderr, "chain table=%u tile=%u len=%u L=%u rounds=%u redone=%u\n", wave, tb, tileLen, L, dbgRounds, dbgRedo);
#endif
                carry = __shfl(fin, (int)(nSegs - 1u));
                gc_wave_sync();
                SEQ_SUB(7);
                for (uint32_t u = lane; u < tileLen; u += 64u) S[nSeq - 1u - (tb + u)] = T.tOut[SEQ_TO(u)];
                gc_wave_sync();
                SEQ_SUB(8);
            }
            if (lane == 0u) T.finalState = carry;
        }
    }
    __syncthreads();
    SEQ_PHASE(3);         // state chains

    // ---- section header (zstd_compress.c:2939-2953 nbSeq; modes byte; table descriptions LL, OF, ML)
    uint32_t hdrLen;
    {
        uint32_t h = nSeq < 128u ? 1u : (nSeq < 0x7F00u ? 2u : 3u);
        hdrLen = h + 1u + sTab[0].descSize + sTab[1].descSize + sTab[2].descSize;
        if (t == 0) {
            if (h == 1u) out[0] = (uint8_t)nSeq;
            else if (h == 2u) { out[0] = (uint8_t)((nSeq >> 8) + 0x80u); out[1] = (uint8_t)nSeq; }
            else { out[0] = 0xFF; out[1] = (uint8_t)(nSeq - 0x7F00u); out[2] = (uint8_t)((nSeq - 0x7F00u) >> 8); }
            out[h] = (uint8_t)((sTab[0].mode << 6) | (sTab[1].mode << 4) | (sTab[2].mode << 2));
            uint32_t p = h + 1u;
            for (int k = 0; k < 3; k++) for (uint32_t i = 0; i < sTab[k].descSize; i++) out[p++] = sTab[k].desc[i];
        }
    }

    // ---- pack the bitstream, last sequence first
    const uint32_t limit = min(blockLen, (uint32_t)GC_SEQSEC_STRIDE - 64u);
    uint8_t* so = out + hdrLen;
    uint32_t carryBits = 0, carryVal = 0, outBytes = 0; bool overflow = false;
    for (uint32_t tb = 0; tb < nSeq && !overflow; tb += SEQ_T) {
        for (uint32_t i = t; i < SEQ_TILE_WORDS; i += SEQ_T) sTile[i] = 0;
        __syncthreads();
        const uint32_t u = tb + t;
        uint64_t a = 0, c = 0; uint32_t na = 0, nc = 0;
        if (u < nSeq) {
            const uint32_t j = nSeq - 1u - u;
            const uint64_t pk = P[j];
            const uint32_t ll = (uint32_t)(pk & 0x1FFFFu), mlBase = (uint32_t)((pk >> 17) & 0x3FFFFu), offBase = (uint32_t)(pk >> 35);
            const uint32_t llc = cLL[j], mlc = cML[j], ofc = cOF[j];
            if (u != 0u) {
                uint32_t so_ = stOF[j], sm = stML[j], sl = stLL[j];
                a |= (uint64_t)(so_ & ((1u << (so_ >> 10)) - 1u)) << na; na += so_ >> 10;     // low nbBits of the state (stored whole)
                a |= (uint64_t)(sm & ((1u << (sm >> 10)) - 1u)) << na; na += sm >> 10;
                a |= (uint64_t)(sl & ((1u << (sl >> 10)) - 1u)) << na; na += sl >> 10;
            }
            { uint32_t nb = kLLBits[llc]; a |= (uint64_t)(ll & ((1u << nb) - 1u)) << na; na += nb; }
            { uint32_t nb = kMLBits[mlc]; c |= (uint64_t)(mlBase & ((1u << nb) - 1u)); nc += nb; }
            { c |= (uint64_t)(offBase & ((1u << ofc) - 1u)) << nc; nc += ofc; }
        }
        uint32_t tileBits;
        const uint32_t off = seq_excl_scan(na + nc, sWave, &tileBits) + carryBits;
        if (t == 0 && carryBits) atomicOr(&sTile[0], carryVal);
        seq_or_bits(sTile, off, a, na);
        seq_or_bits(sTile, off + na, c, nc);
        uint32_t endBits = carryBits + tileBits;
        const bool lastTile = tb + SEQ_T >= nSeq;
        if (lastTile) {
            if (t == 0) {      // final states ML, OF, LL then the closing bit (zstd_compress_sequences.c:372-376)
                uint32_t e = endBits;
                seq_or_bits(sTile, e, sTab[2].finalState & ((1u << sTab[2].tableLog) - 1u), sTab[2].tableLog); e += sTab[2].tableLog;
                seq_or_bits(sTile, e, sTab[1].finalState & ((1u << sTab[1].tableLog) - 1u), sTab[1].tableLog); e += sTab[1].tableLog;
                seq_or_bits(sTile, e, sTab[0].finalState & ((1u << sTab[0].tableLog) - 1u), sTab[0].tableLog); e += sTab[0].tableLog;
                seq_or_bits(sTile, e, 1u, 1u);
            }
            endBits += sTab[0].tableLog + sTab[1].tableLog + sTab[2].tableLog + 1u;
        }
        __syncthreads();
        const uint32_t flush = lastTile ? (endBits + 7u) >> 3 : endBits >> 3;
        if (hdrLen + outBytes + flush > limit) overflow = true;
        else {
            for (uint32_t i = t; i < flush; i += SEQ_T) so[outBytes + i] = (uint8_t)(sTile[i >> 2] >> ((i & 3u) * 8u));
            carryBits = endBits & 7u;
            carryVal = lastTile ? 0u : ((sTile[flush >> 2] >> ((flush & 3u) * 8u)) & 0xFFu);
            outBytes += flush;
        }
        __syncthreads();
    }
    SEQ_PHASE(4);         // pack
    if (t == 0) { info[b].seqSecSize = overflow ? 0xFFFFFFFFu : hdrLen + outBytes; info[b].nSeq = nSeq; }
}
