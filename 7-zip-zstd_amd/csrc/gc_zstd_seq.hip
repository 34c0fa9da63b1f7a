// gc_zstd_seq.hip -- K3: the sequences section of every zstd block, as FOUR kernels of different shape.
//
// Replaces ZSTD_seqToCodes (C/zstd/zstd_compress.c:2693), ZSTD_updateRep (zstd_compress_internal.h:818),
// ZSTD_buildSequencesStatistics (zstd_compress.c:2763: histograms, ZSTD_selectEncodingType, ZSTD_buildCTable ->
// FSE_normalizeCount / FSE_writeNCount / FSE_buildCTable_wksp) and ZSTD_encodeSequences_body
// (zstd_compress_sequences.c:291-383).
//
//   K3a codes   workgroup of 256 per block, parallel over sequences:
//                 merge    chains of capped matches (same offset, litLength 0) collapse into one sequence
//                 repcode  repeat-offset history as a scan: rep1 = previous offset, rep2 = the offset before the current run
//                          of equal offsets.  Only codes whose meaning depends on (rep1, rep2) are emitted, so the decoder's
//                          third history slot never matters and the whole assignment is order-independent
//                 codes    LL/ML/OF code + extra bits, three histograms
//   K3b tables  ONE WAVE per (block, table): mode (predefined / RLE / FSE), normalisation, NCount header, CTable -- serial work of one lane
//   K3c chains  ONE WAVE per (block, table) walks the table's FSE state backwards over all sequences: 64 segments at once, each lane
//               finding its start state by running a few symbols ahead of its segment (FSE states forget their origin), checked and
//               repaired against the predecessor's final state, so the result is exactly the serial walk
//   K3d pack    workgroup of 256 per block: per-sequence bit counts -> prefix sums -> fields OR-ed into an LDS tile -> bytes stream out
//
// Why four: as one kernel (rounds 1-2) a block held 8 waves and 53 KB of LDS for the whole time while its serial middle (tables: three
// lanes, chains: three waves; 70 % of the block's time) ran -- three blocks per CU, a machine mostly waiting.  Apart, the serial stages are
// one-wave workgroups with 3-7 KB of LDS, a few thousand of them in flight, and the parallel stages are small workgroups that come and go.
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
#define SEQ_CHAIN_TILE GC_SEQ_CHAIN_TILE   // sequences per state-chain tile
#define SEQ_CHAIN_SEG  64u     // sequences per lane and tile
#define SEQ_TC(u) ((u) + ((u) >> 6))                     // index of tile element u in tCode: row stride 65 bytes (a lane walks ITS segment, i.e.
                                                         // the lanes of a wave touch elements 64 apart: unpadded they would share two LDS banks)
#ifndef SEQ_WARM_SEGS
#define SEQ_WARM_SEGS  4u      // a lane looks this many segments back for a point where all state walks meet
#endif
#define SEQ_CHK        8u      // the walk keeps its state at every 8th symbol of a segment: where a repair walk can tell that it has rejoined

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

struct SeqTab {              // one FSE table in LDS (K3b builds it)
    uint16_t state[512];
    GcFseSym tt[64];
    int16_t  norm[64];
    uint32_t count[64];
    uint8_t  spread[512];
    uint16_t cumul[66];
    uint8_t  desc[96];       // table description bytes for the section header
    uint32_t descSize, mode, tableLog, maxSym, finalState;
    uint32_t tabMaxSym;      // last symbol described by norm[] (predefined: whole default table)
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

#define SEQ_PHASE(i) do { if (prof && t == 0) { unsigned long long now_ = gc_clock(); atomicAdd(&prof[i], now_ - tprev); tprev = now_; } } while (0)

// ------------------------------------------------------------------------------------------------ K3a codes
// block-wide inclusive max scan of 64-bit keys (SEQ_T threads)
__device__ __forceinline__ uint64_t seq_incl_maxscan64(uint64_t v, uint64_t* sWave64, uint64_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        if (lane >= (uint32_t)d && o > v) v = o;
    }
    if (lane == 63u) sWave64[wave] = v;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (uint32_t w = 0; w < SEQ_T / 64u; w++) { const uint64_t c = sWave64[w]; if (w < wave && c > before) before = c; if (c > all) all = c; }
    __syncthreads();
    *total = all;
    return before > v ? before : v;
}

// The raw records are taken SEQ_T at a time; merged sequences collect in an LDS ring and are coded SEQ_T at a time from there, so nothing this
// kernel computes is written to memory and read back (rounds 1-2 wrote the merged sequences out, then read them and their neighbours again).
#define SEQ_RING 1024u       // merged sequences the ring holds (>= 2 * SEQ_T + 1)
extern "C" __global__ void __launch_bounds__(SEQ_T)
gc_zstd_seq_codes_kernel(const GcSeqRaw* __restrict__ seqRaw, const GcBlockMeta* __restrict__ meta,
                   uint64_t* __restrict__ seqPacked,      // out: ll | mlBase << 17 | offBase << 35 per merged sequence
                   uint32_t* __restrict__ seqOff,         // (unused since the ring; kept in the signature for the workspace's sake)
                   uint8_t* __restrict__ codes,           // out: 3 * GC_MAX_SEQ_PER_BLOCK per block (LL, OF, ML)
                   GcSeqHist* __restrict__ hist, uint8_t* __restrict__ seqSec, GcSectionInfo* __restrict__ info,
                   uint32_t frameBlocks,                  // blocks per zstd frame: repeat offsets carry over inside a frame
                   unsigned long long* __restrict__ prof /* optional per-phase cycle sums */)
{
    __shared__ uint32_t sCount[3][64];
    __shared__ uint32_t sWave[SEQ_T / 64u];
    __shared__ uint64_t sWave64[SEQ_T / 64u];
    __shared__ uint64_t sPair[SEQ_T];
    __shared__ uint32_t rLL[SEQ_RING], rML[SEQ_RING], rOFF[SEQ_RING];
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const uint32_t nRaw = meta[b].nSeqRaw;
    const GcSeqRaw* R = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint64_t* P = seqPacked + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* cLL = codes + (uint64_t)b * 3u * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* cOF = cLL + GC_MAX_SEQ_PER_BLOCK;
    uint8_t* cML = cOF + GC_MAX_SEQ_PER_BLOCK;
    (void)seqOff;
    if (nRaw == 0) { if (t == 0) { seqSec[(uint64_t)b * GC_SEQSEC_STRIDE] = 0; info[b].seqSecSize = 1; info[b].nSeq = 0; } return; }
    for (uint32_t i = t; i < 3u * 64u; i += SEQ_T) sCount[i >> 6][i & 63u] = 0;
    unsigned long long tprev = prof ? gc_clock() : 0ull;

    // virtual history of the repeat offsets: index -1 -> offset 1, -2 -> 4 (start state {1,4,8}, zstd_internal.h:65) in the first block of a
    // frame.  In later blocks the decoder arrives with the history the previous block left behind; this block is coded
    // without knowing it (blocks stay independent units of work): the virtual history is "unknown" (0, never equal to
    // an offset), so no repeat code refers to it.  rep1 = previous offset and rep2 = offset before the current run hold
    // for the decoder whatever the unknown part was, because every code emitted below moves exactly those two slots.
    const bool firstInFrame = (b % frameBlocks) == 0u;
    const uint32_t virt1 = firstInFrame ? 1u : 0u, virt2 = firstInFrame ? 4u : 0u;
    uint32_t nMerged = 0, nCoded = 0;        // merged sequences in the ring so far / coded so far (uniform)
    uint32_t carryOff = virt1;               // offset of the last coded sequence
    uint64_t carryPair = (1ull << 32) | virt2;   // (start index + 2 of the run of equal offsets that holds the last coded sequence, offset in front of that run)

    // S2 on the SEQ_T sequences from nCoded on (or the last few): repeat offsets as scans, codes, histograms
    auto code_batch = [&](uint32_t count) {
        const uint32_t j = nCoded + t;
        const bool on = t < count;
        uint32_t off = 0, prevOff = carryOff, ll = 0, ml = 0;
        if (on) {
            off = rOFF[j & (SEQ_RING - 1u)]; ll = rLL[j & (SEQ_RING - 1u)]; ml = rML[j & (SEQ_RING - 1u)];
            if (t) prevOff = rOFF[(j - 1u) & (SEQ_RING - 1u)];
        }
        // a run of equal offsets starts at j: key = (j + 2, the offset in front of the run); the scan hands every j the key of its run
        const uint64_t key = (on && off != prevOff) ? (((uint64_t)(j + 2u)) << 32) | prevOff : 0ull;
        uint64_t tot;
        uint64_t pair = seq_incl_maxscan64(key, sWave64, &tot);
        if (carryPair > pair) pair = carryPair;
        sPair[t] = pair;
        __syncthreads();
        const uint64_t pairPrev = t ? sPair[t - 1u] : carryPair;                   // ... of the run that holds j - 1
        if (on) {
            const uint32_t rep1 = prevOff, rep2 = (uint32_t)pairPrev;
            uint32_t offBase;
            if (ll != 0u) offBase = (off == rep1) ? 1u : ((off == rep2) ? 2u : off + 3u);
            else offBase = (off == rep2) ? 1u : ((rep1 > 1u && off == rep1 - 1u) ? 3u : off + 3u);
            const uint32_t mlBase = ml - 3u;
            const uint32_t llc = seq_ll_code(ll), mlc = seq_ml_code(mlBase), ofc = gc_hibit32(offBase);
            cLL[j] = (uint8_t)llc; cOF[j] = (uint8_t)ofc; cML[j] = (uint8_t)mlc;
            atomicAdd(&sCount[0][llc], 1u); atomicAdd(&sCount[1][ofc], 1u); atomicAdd(&sCount[2][mlc], 1u);
            P[j] = (uint64_t)ll | ((uint64_t)mlBase << 17) | ((uint64_t)offBase << 35);   // ll, mlBase, offBase
#ifdef HIPEMU
            if (getenv("GC_TRACE")) fprintf(stderr, "E %u ofv=%u ml=%u ll=%u off=%u rep1=%u rep2=%u\n", j, offBase, ml, ll, off, rep1, rep2);
#endif
        }
        if (tot > carryPair) carryPair = tot;
        carryOff = rOFF[(nCoded + count - 1u) & (SEQ_RING - 1u)];
        nCoded += count;
        __syncthreads();
    };

    GcSeqRaw rNext, qNext;                   // this thread's record of the next round and the record in front of it (requested a round ahead)
    rNext.litRank = rNext.offml = qNext.litRank = qNext.offml = 0;
    if (t < nRaw) { rNext = R[t]; if (t) qNext = R[t - 1u]; }
    for (uint32_t tb = 0; tb < nRaw; tb += SEQ_T) {
        // ---- S1: merge chains of capped matches.  Head = first record of a run with equal offset and litLength 0; the records behind it
        //      add their lengths to it (a chain may run on into the next SEQ_T records: the ring's last sequence stays open until then)
        const uint32_t i = tb + t;
        const GcSeqRaw r = rNext, q = qNext;
        if (i + SEQ_T < nRaw) { rNext = R[i + SEQ_T]; qNext = R[i + SEQ_T - 1u]; }
        uint32_t head = 0, ll = 0, off = 0, ml = 0;
        if (i < nRaw) {
            uint32_t prevRank = 0, prevOff = 0;
            if (i) { prevRank = q.litRank; prevOff = q.offml >> 8; }
            ll = r.litRank - prevRank; off = r.offml >> 8; ml = r.offml & 0xFFu;
            head = (i == 0u || ll != 0u || off != prevOff) ? 1u : 0u;
        }
        uint32_t tot;
        const uint32_t before = seq_excl_scan(head, sWave, &tot);
        const uint32_t slot = (nMerged + before + head - 1u) & (SEQ_RING - 1u);      // (a record that is no head: the sequence of the head in front of it)
        if (head) { rLL[slot] = ll; rML[slot] = ml; rOFF[slot] = off; }
        __syncthreads();
        if (i < nRaw && !head) atomicAdd(&rML[slot], ml);
        nMerged += tot;
        __syncthreads();
        // ---- S2 whenever SEQ_T sequences are complete (the last one in the ring may still grow)
        while (nMerged - nCoded > SEQ_T) code_batch(SEQ_T);
    }
    SEQ_PHASE(0);         // merge (and the batches coded on the way)
    while (nCoded < nMerged) code_batch(nMerged - nCoded < SEQ_T ? nMerged - nCoded : SEQ_T);
    for (uint32_t i = t; i < 3u * 64u; i += SEQ_T) hist[b].count[i >> 6][i & 63u] = sCount[i >> 6][i & 63u];
    if (t == 0) info[b].nSeq = nMerged;
    SEQ_PHASE(1);         // repcodes + codes + histograms
}

// ------------------------------------------------------------------------------------------------ K3b tables
// workgroup (one wave) bt = 3 * block + table; table 0 LL, 1 OF, 2 ML
extern "C" __global__ void __launch_bounds__(64)
gc_zstd_seq_tables_kernel(const GcSeqHist* __restrict__ hist, const GcSectionInfo* __restrict__ info, GcSeqTabG* __restrict__ tabs,
                          unsigned long long* __restrict__ prof)
{
    __shared__ SeqTab T;
    const uint32_t t = threadIdx.x, b = blockIdx.x / 3u, which = blockIdx.x % 3u;
    const uint32_t nSeq = info[b].nSeq;
    if (nSeq == 0u) return;
    unsigned long long tprev = prof ? gc_clock() : 0ull;
    T.count[t] = hist[b].count[which][t];
    T.norm[t] = 0;
    if (t == 0) T.finalState = 0;
    gc_wave_sync();
    if (t == 0) seq_build_table(T, (int)which, nSeq);
    gc_wave_sync();
    GcSeqTabG& G = tabs[blockIdx.x];
    for (uint32_t i = t; i < 512u; i += 64u) G.state[i] = T.state[i];
    G.tt[t] = T.tt[t];
    G.norm[t] = T.norm[t];
    for (uint32_t i = t; i < 96u; i += 64u) G.desc[i] = T.desc[i];
    if (t == 0) { G.descSize = T.descSize; G.mode = T.mode; G.tableLog = T.tableLog; G.maxSym = T.maxSym; G.tabMaxSym = T.tabMaxSym; G.finalState = 0; }
    SEQ_PHASE(2);         // tables
}

// ------------------------------------------------------------------------------------------------ K3c chains
// One wave walks table `which`'s FSE state backwards over the block's sequences (last sequence first = processing index 0).
//      The walk is serial through the state, but walks that started from different states meet at the next symbol whose
//      normalised count is 1 (every state has the same successor there; the table builder makes sure such symbols
//      exist).  So the 64 lanes of the wave walk 64 consecutive segments of 64 sequences at once: a lane first runs
//      ahead of its segment from the nearest such meeting point before it (no output), then its own segment.  Afterwards every
//      segment's start state is compared with its predecessor's final state; a segment that started wrong is walked again
//      from the right state, only until it rejoins the states it went through before (kept at every 8th symbol), and this repeats until
//      nothing changes (each round settles at least one more segment, so the result is exactly the serial walk).
// Output: nbBits << 10 | state BEFORE the symbol (10 bits): the bits to emit are its low nbBits.  Symbol k of segment `lane` of a tile is processing
// index u = 64 * lane + k; it is stored at SEQ_ST_IDX(u) = 256 * (lane / 4) + 4 * k + lane % 4, i.e. the 256 symbols of four neighbouring segments
// share 512 bytes: a step of the walk writes 8 bytes per group of four lanes, sixteen steps fill a line, and the 256 sequences a pack
// workgroup takes per round are four whole lines (in segment-major order the same round touched 64 lines: 3.7 GB fetched for 0.4 GB of states).
#define SEQ_ST_IDX(u) (((u) & ~255u) + (((u) & 63u) << 2) + (((u) >> 6) & 3u))
extern "C" __global__ void __launch_bounds__(64)
gc_zstd_seq_chain_kernel(const uint8_t* __restrict__ codes, const GcSectionInfo* __restrict__ info, GcSeqTabG* __restrict__ tabs,
                         uint16_t* __restrict__ states,   // out: 3 * GC_SEQ_ST_STRIDE per block (LL, OF, ML)
                         unsigned long long* __restrict__ prof)
{
    __shared__ uint16_t sState[512];
    __shared__ GcFseSym sTT[64];
    __shared__ uint8_t  tCode[SEQ_CHAIN_TILE + SEQ_CHAIN_TILE / SEQ_CHAIN_SEG + 3u];
    __shared__ uint64_t sMeet[SEQ_CHAIN_TILE / 64u];
    __shared__ uint16_t sChk[64u * (SEQ_CHAIN_SEG / SEQ_CHK + 1u)];        // [lane][checkpoint], row stride 9 half-words
    const uint32_t t = threadIdx.x, lane = t, b = blockIdx.x / 3u, which = blockIdx.x % 3u;
    const uint32_t nSeq = info[b].nSeq;
    if (nSeq == 0u) return;
    GcSeqTabG& G = tabs[blockIdx.x];
    if (G.mode == 1u) return;                                 // RLE table: no state bits at all (the pack kernel knows)
    unsigned long long tprev = prof ? gc_clock() : 0ull;
    unsigned long long tsub = 0;
#define SEQ_SUB(i) do { if (prof && t == 0) { unsigned long long now_ = gc_clock(); if (tsub) atomicAdd(&prof[i], now_ - tsub); tsub = now_; } } while (0)
    const uint8_t* C = codes + ((uint64_t)b * 3u + which) * GC_MAX_SEQ_PER_BLOCK;
    uint16_t* S = states + ((uint64_t)b * 3u + which) * GC_SEQ_ST_STRIDE;
    for (uint32_t i = lane; i < 512u; i += 64u) sState[i] = G.state[i];
    sTT[lane] = G.tt[lane];
    const uint32_t L = G.tableLog;
    int nv = 0;
    if (lane <= G.tabMaxSym) nv = G.norm[lane];
    const uint64_t resetMask = __ballot(nv == 1 || nv == -1);    // symbols that send every state to the same successor
    gc_wave_sync();
    uint32_t carry = 0;                               // state after the last sequence of the previous tile
    for (uint32_t tb = 0; tb < nSeq; tb += SEQ_CHAIN_TILE) {
        const uint32_t tileLen = min(SEQ_CHAIN_TILE, nSeq - tb);
        if (prof && t == 0) tsub = gc_clock();
        // stage the codes (independent loads, eight in flight per lane), then per 64 sequences one mask of the positions
        // that hold a count-1 symbol ("meeting points")
        // (tile element u is sequence nSeq - 1 - (tb + u): sixteen elements per lane and load, read upwards in memory and stored back to front)
        for (uint32_t u16 = lane * 16u; u16 < tileLen; u16 += 64u * 16u) {
            const uint32_t jTop = nSeq - 1u - (tb + u16);         // sequence of element u16; the lane's elements are jTop, jTop - 1, ...
            if (u16 + 16u <= tileLen) {
                GcU4 v; __builtin_memcpy(&v, C + (jTop - 15u), 16);
                const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                for (uint32_t i = 0; i < 16u; i++) tCode[SEQ_TC(u16 + i)] = (uint8_t)(w[(15u - i) >> 2] >> (((15u - i) & 3u) * 8u));
            } else
                for (uint32_t i = 0; u16 + i < tileLen; i++) tCode[SEQ_TC(u16 + i)] = C[jTop - i];
        }
        gc_wave_sync();
        SEQ_SUB(5);
        for (uint32_t k = 0; k < tileLen; k += 64u) {
            const uint32_t u = k + lane;
            const bool isR = u < tileLen && ((resetMask >> tCode[SEQ_TC(u)]) & 1ull) != 0ull;
            const uint64_t bal = __ballot(isR);
            if (lane == 0u) sMeet[k >> 6] = bal;
        }
        gc_wave_sync();
        const uint32_t nSegs = (tileLen + SEQ_CHAIN_SEG - 1u) / SEQ_CHAIN_SEG;      // <= 64: lane i walks segment i
        const uint32_t u0 = lane * SEQ_CHAIN_SEG, u1 = min(u0 + SEQ_CHAIN_SEG, tileLen);
        uint32_t first = u0;                          // first index of the segment that emits bits
        uint32_t st0 = carry;
        if (lane < nSegs) {
            if (tb + u0 == 0u) { st0 = gc_fse_init_state(sState, sTT[tCode[SEQ_TC(0)]]); S[0] = 0; first = 1u; }   // FSE_initCState2: no bits
            else if (lane != 0u) {
                // run ahead from the nearest meeting point in the SEQ_WARM_SEGS segments before this one (from there the
                // state is exact whatever it was before); if there is none, from SEQ_WARM_SEGS segments back (a guess)
                uint32_t w = lane > SEQ_WARM_SEGS ? (lane - SEQ_WARM_SEGS) * SEQ_CHAIN_SEG : 0u;
                for (uint32_t c = lane; c-- > 0u && c + SEQ_WARM_SEGS >= lane; ) {
                    const uint64_t m = sMeet[c];
                    if (m) { w = c * 64u + 63u - (uint32_t)__clzll((long long)m); break; }
                }
                if (tb + w == 0u) { st0 = gc_fse_init_state(sState, sTT[tCode[SEQ_TC(0)]]); w = 1u; }                   // exact, not a guess
                else st0 = 1u << L;
                if (w < u0) {                              // the symbol's table entry does not depend on the state: fetched one
                    GcFseSym sy = sTT[tCode[SEQ_TC(w)]];      // step ahead, so a step costs one dependent LDS read, not three
                    for (; w < u0; w++) {
                        const GcFseSym nx = sTT[tCode[SEQ_TC(w + 1u < u0 ? w + 1u : w)]];
                        const uint32_t nb = (st0 + sy.deltaNbBits) >> 16;
                        st0 = sState[(st0 >> nb) + (uint32_t)sy.deltaFindState];
                        sy = nx;
                    }
                }
            }
        }
        SEQ_SUB(6);
        uint32_t fin = st0;
        bool redo = lane < nSegs;
        bool firstPass = true;
        uint16_t* chk = sChk + lane * (SEQ_CHAIN_SEG / SEQ_CHK + 1u);
        uint16_t* Sk = S + tb + ((lane >> 2) << 8) + (lane & 3u);      // symbol k of this lane's segment -> Sk[4 * k]
#ifdef HIPEMU
        uint32_t dbgRounds = 0, dbgRedo = 0;
#endif
        for (;;) {
            if (redo) {
                uint32_t state = st0, u = first;
                GcFseSym sy = sTT[tCode[SEQ_TC(u < u1 ? u : u1 - 1u)]];
                for (; u < u1; u++) {
                    const uint32_t k = u - u0;
                    if ((k & (SEQ_CHK - 1u)) == 0u) {
                        if (!firstPass && chk[k / SEQ_CHK] == (uint16_t)state) break;     // rejoined: the rest is what it was
                        chk[k / SEQ_CHK] = (uint16_t)state;
                    }
                    const GcFseSym nx = sTT[tCode[SEQ_TC(u + 1u < u1 ? u + 1u : u)]];
                    const uint32_t nb = (state + sy.deltaNbBits) >> 16;
                    Sk[4u * k] = (uint16_t)((nb << 10) | (state & 0x3FFu));
                    state = sState[(state >> nb) + (uint32_t)sy.deltaFindState];
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
        if (getenv("GC_TRACE_CHAIN") && lane == 0) fprintf(stderr, "chain table=%u tile=%u len=%u L=%u rounds=%u redone=%u\n", which, tb, tileLen, L, dbgRounds, dbgRedo);
#endif
        carry = __shfl(fin, (int)(nSegs - 1u));
        gc_wave_sync();
        SEQ_SUB(7);
    }
    if (lane == 0u) G.finalState = carry;
    SEQ_PHASE(3);         // state chains
}

// ------------------------------------------------------------------------------------------------ K3d pack
extern "C" __global__ void __launch_bounds__(SEQ_T)
gc_zstd_seq_pack_kernel(const uint64_t* __restrict__ seqPacked, const uint8_t* __restrict__ codes, const uint16_t* __restrict__ states,
                        const GcSeqTabG* __restrict__ tabs, uint8_t* __restrict__ seqSec, GcSectionInfo* __restrict__ info, uint64_t srcSize,
                        unsigned long long* __restrict__ prof)
{
    __shared__ uint32_t sWave[SEQ_T / 64u];
    __shared__ uint32_t sTile[SEQ_TILE_WORDS];
    __shared__ uint32_t sMode[3], sLog[3], sFinal[3], sDesc[3];
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const uint32_t nSeq = info[b].nSeq;
    if (nSeq == 0u) return;                                   // (the codes kernel has written the one-byte section)
    unsigned long long tprev = prof ? gc_clock() : 0ull;
    const uint64_t blockBase = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const uint64_t* P = seqPacked + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    const uint8_t* cLL = codes + (uint64_t)b * 3u * GC_MAX_SEQ_PER_BLOCK;
    const uint8_t* cOF = cLL + GC_MAX_SEQ_PER_BLOCK;
    const uint8_t* cML = cOF + GC_MAX_SEQ_PER_BLOCK;
    const uint16_t* stLL = states + (uint64_t)b * 3u * GC_SEQ_ST_STRIDE;
    const uint16_t* stOF = stLL + GC_SEQ_ST_STRIDE;
    const uint16_t* stML = stOF + GC_SEQ_ST_STRIDE;
    const GcSeqTabG* G = tabs + (uint64_t)b * 3u;
    uint8_t* out = seqSec + (uint64_t)b * GC_SEQSEC_STRIDE;
    if (t < 3u) { sMode[t] = G[t].mode; sLog[t] = G[t].tableLog; sFinal[t] = G[t].finalState; sDesc[t] = G[t].descSize; }
    __syncthreads();

    // ---- section header (zstd_compress.c:2939-2953 nbSeq; modes byte; table descriptions LL, OF, ML)
    uint32_t hdrLen;
    {
        uint32_t h = nSeq < 128u ? 1u : (nSeq < 0x7F00u ? 2u : 3u);
        hdrLen = h + 1u + sDesc[0] + sDesc[1] + sDesc[2];
        if (t == 0) {
            if (h == 1u) out[0] = (uint8_t)nSeq;
            else if (h == 2u) { out[0] = (uint8_t)((nSeq >> 8) + 0x80u); out[1] = (uint8_t)nSeq; }
            else { out[0] = 0xFF; out[1] = (uint8_t)(nSeq - 0x7F00u); out[2] = (uint8_t)((nSeq - 0x7F00u) >> 8); }
            out[h] = (uint8_t)((sMode[0] << 6) | (sMode[1] << 4) | (sMode[2] << 2));
            uint32_t p = h + 1u;
            for (int k = 0; k < 3; k++) for (uint32_t i = 0; i < sDesc[k]; i++) out[p++] = G[k].desc[i];
        }
    }
    const bool rleLL = sMode[0] == 1u, rleOF = sMode[1] == 1u, rleML = sMode[2] == 1u;

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
                const uint32_t si = SEQ_ST_IDX(u);
                const uint32_t so_ = rleOF ? 0u : stOF[si], sm = rleML ? 0u : stML[si], sl = rleLL ? 0u : stLL[si];
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
                seq_or_bits(sTile, e, sFinal[2] & ((1u << sLog[2]) - 1u), sLog[2]); e += sLog[2];
                seq_or_bits(sTile, e, sFinal[1] & ((1u << sLog[1]) - 1u), sLog[1]); e += sLog[1];
                seq_or_bits(sTile, e, sFinal[0] & ((1u << sLog[0]) - 1u), sLog[0]); e += sLog[0];
                seq_or_bits(sTile, e, 1u, 1u);
            }
            endBits += sLog[0] + sLog[1] + sLog[2] + 1u;
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
    if (t == 0) info[b].seqSecSize = overflow ? 0xFFFFFFFFu : hdrLen + outBytes;
}
