// gc_lz_dpl.hip -- W7L: the price-based ("optimal") parse with one LANE per window (round 4).
//
// Same job as W7 of gc_lz_price.hip (LZMA_optimalParse, C/fast-lzma2/lzma2_enc.c:949-1440; ZSTD_compressBlock_opt_generic,
// C/zstd/zstd_opt.c:1077): the cheapest way through a window of positions under static per-block prices, a step being a literal,
// any prefix of a match candidate, or -- LZMA -- a repeat of one of the FOUR last distances of the path (rep0..rep3,
// LZMA_getRepPrice lzma2_enc.c:289, the rep loop of the optimal parser :1090, four distances per node :990-998).
//
// W7 gave a window to a WAVE: the 64 open nodes in one VGPR, one node per ~45 instructions, i.e. 64 lanes busy with the edges of ONE
// position, most of which do not exist.  The windows are independent and there are tens of thousands of them (211.9 MB = 103 000
// windows of 2 KiB), so here a window is a LANE: every lane runs the textbook forward programme on its own window, the 64 lanes of a
// wave step through their windows in lockstep (node i of all 64 windows at step i), and one vector instruction works on 64 positions.
// What a lane needs per node lives in LDS columns of its own ([slot][lane]: no bank conflicts, no sharing, no barriers):
//   sCost  ring of the DPL_M open nodes: 64-bit words cost | bytes left of a capped match | distance | class | length, relaxed with
//          ds_min_u64 (fire and forget: the minimum carries its back pointer AND its distance)
//   sReps  ring of the last DPL_M final nodes: the four repeat distances of the cheapest way to each (the programme compares distances, it
//          never dereferences one: what a repeat matches comes from the hints below)
// Edges are at most DPL_M = 16 bytes long: a longer match is a chain of pieces -- the head is priced with the length price of the
// whole match, the rest of it travels with the node it reaches ("bytes left") and is offered there at DP_CONT_PRICE, any length --
// which keeps both rings at 16 slots (24 KiB of LDS per wave with the price table: 5-6 waves per CU) and the relax loops short.
//
// Repeats at EVERY position without a dependent memory access in the node loop: which bytes repeat at position p at distance d is a
// property of (p, d), not of the path.  The path only decides WHICH d are its repeats.  So the HINTS of a position are fixed a group of
// four positions ahead of the programme: the four repeat distances of its newest final node (a path that only adds literals or repeats
// keeps them; a new match pushes them down one place) plus the last two distances of a cheap "shadow" greedy parse of the finder's
// records that runs along (the distance of a match the path is about to take).  For each hint 16 bytes at p - d are requested one group
// early and compared with the window's own bytes when they have arrived.  At node i a hint becomes an edge (lengths 2..16, price of
// rep k) iff its distance is rep k of THAT node; with rep0 and one byte equal it is LZMA's short repeat.
// Windows start DPL_WARM positions early: the programme runs over the end of the window in front (clipping its edges at the window's
// first node exactly as that window's own lane does) for nothing but the state it arrives with -- the repeat distances and the rest of
// a match cut by the boundary -- which is what makes 2 KiB windows affordable on data that is coded with repeats (8 MiB of ROCm shared
// objects: 4 096 windows that each re-establish four distances with full-price matches cost 2.5 % of the stream).
//
// The back pointers of a window go to the window's own slice of the record array (one word per node); the walk back from the last
// node rewrites that slice in place, slot by slot in lockstep, into what W6 follows: (distance << 8 | length) where the path starts a
// match, 0 elsewhere; neighbouring pieces with one distance leave as records of up to 64 bytes.
#include "gc_mf.h"
#include "gc_lz_parse.h"
#ifdef HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif

#ifndef DPL_LENBITS
#define DPL_LENBITS  5u
#endif
#define DPL_M        (1u << DPL_LENBITS)
#define DPL_MMASK    (DPL_M - 1u)
#ifndef DPL_MR
#define DPL_MR       16u              // final nodes whose repeat distances are kept: the node an edge of more than DPL_MR bytes comes from is
#endif                                 // taken to have the distances of the oldest one (they differ only where the cheapest ways to the two differ)
// low word of a node: distance | class | length - 1
#define DPL_LO_LEN(lo)   (((lo) & DPL_MMASK) + 1u)
#define DPL_LO_CLS(lo)   (((lo) >> DPL_LENBITS) & 7u)
#define DPL_LO_DIST(lo)  ((lo) >> (DPL_LENBITS + 3u))          // 24 bits (round 6: 16 MiB frames; the "capped" flag moved to the high word)
#define DPL_LO(dist, cls, len) (((dist) << (DPL_LENBITS + 3u)) | ((cls) << DPL_LENBITS) | ((len) - 1u))
// high word of a node: cost << 7 | capped << 6 | bytes left of a capped match (0..63)
#define DPL_CSH          7u
#define DPL_HI_CAP       64u
#define DPL_INF      0xFFFFFFFFFFFFFFFFull
// Two waves per group of 64 windows (GC_DPL_THREADS = 128, gc_mf.h): both step through the nodes together, wave 0 expands a node's literal, finder candidates and
// the rest of a capped match, stores back pointers and walks back; wave 1 keeps the tracked distances and expands the node's repeats.  One LDS barrier per node.
// The rings are two slots longer than an edge / one longer than the reach of a repeat look-up, so that what one wave resets or rewrites during a step is nothing
// the other wave can still read or already target in that step: slot of node i - 1 is cleared in step i (node i + DPL_RC - 1 is out of reach of an edge of <= DPL_M).
#define DPL_RC       (DPL_M + 2u)     // cost ring
#define DPL_RR       (DPL_MR + 1u)    // ring of repeat distances
#define DPL_PAIR     (GC_DPL_THREADS > 64u)      // more than one wave per group
#if defined(HIPEMU)
#define DPL_BARRIER() do { if (DPL_PAIR) __syncthreads(); } while (0)
#else
#define DPL_BARRIER() do { if (DPL_PAIR) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } } while (0)      // (LDS only: the global loads in flight stay in flight)
#endif
#define DPL_CONT     4u               // the rest of a capped match: a quarter of a bit (as DP_CONT_PRICE of W7)
#define DPL_WARM     128              // positions in front of a window that the programme runs over for its state
#define DPL_NC       2u               // requests in flight per group of nodes: the last one for the shadow parse's distances
#define DPL_TABW     7u               // length prices kept in registers: lengths 0 .. 13, two per word (the static part of a relax loop ends at 12)
#define DPL_NT       6u               // tracked distances: rep0..rep3 of the newest final node + the last two of the shadow parse
// class of an edge (bits 4..6 of the low word)
#define DPL_LIT      0u
#define DPL_NEW      1u               // a match at a distance that is no repeat of the node
#define DPL_REP0     2u               // .. 5: rep0..rep3
#define DPL_SREP     6u               // LZMA short repeat (one byte at rep0)
#define DPL_CONTC    7u               // continuation of the capped piece in front of it
#ifndef DPL_SREP_ANY
#define DPL_SREP_ANY 1               // 1: the short repeat is offered at the node's rep0 whether or not that is known for certain (L2 codes a wrong one as a literal); 0: round 4
#endif
#ifndef DPL_ALL_LENGTHS
#define DPL_ALL_LENGTHS (MINLEN == 3u && REPS)      // zstd's W7L: every length of a candidate is an edge (relax())
#endif
#define DPL_SURE     0x80000000u      // in rep0 of a node: the distance is the decoder's rep0 for certain (a match of this window lies on the way)

__device__ __forceinline__ uint32_t dpl_item(uint32_t bid, uint32_t per) { return (bid & (GC_XCDS - 1u)) * per + (bid >> 3); }

struct DplReps { uint32_t r0 /* | DPL_SURE */, r1, r2, r3; };      // 0 = none
__device__ __forceinline__ uint32_t dpl_which(const DplReps& s, uint32_t d)     // index of distance d among the node's four (4 = none)
{
    return (s.r0 & ~DPL_SURE) == d ? 0u : (s.r1 == d ? 1u : (s.r2 == d ? 2u : (s.r3 == d ? 3u : 4u)));
}
// distance d moves to the front (a new one pushes the last out): LzmaDec.c's rep0..rep3 update
__device__ __forceinline__ void dpl_mtf(DplReps& s, uint32_t d, uint32_t sure)
{
    const uint32_t a = s.r0 & ~DPL_SURE;
    if (a == d) { s.r0 = d | sure | (s.r0 & DPL_SURE); return; }
    if (s.r1 == d) { s.r1 = a; }
    else if (s.r2 == d) { s.r2 = s.r1; s.r1 = a; }
    else { s.r3 = s.r2; s.r2 = s.r1; s.r1 = a; }
    s.r0 = d | sure;
}

#if defined(DPL_PROF) && !defined(HIPEMU)
__device__ unsigned long long g_dplProf[32];      // [0..15] wave 0 of a pair, [16..31] wave 1; per wave: sections 0..7 of a node step, 8 = walk back, 9 = waiting at the barrier
extern "C" int gc_dpl_prof_read(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dplProf), sizeof(g_dplProf)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = { 0 }; if (hipMemcpyToSymbol(HIP_SYMBOL(g_dplProf), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define DPL_T(k) do { const unsigned long long now_ = clock64(); pacc[k] += now_ - ptick; ptick = now_; } while (0)
#else
#define DPL_T(k) do { } while (0)
#endif

template <bool REPS, uint32_t MINLEN, bool SAMPLE /* a wave = two blocks: false = all their windows (2 x 32 x 4 KiB), true = a sample (32 x 512 B of each, counted) */>
__device__ __forceinline__ void dpl_run(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t nBlocks, uint32_t per, uint32_t frameBlocks, uint32_t phaseArg, uint32_t* __restrict__ dpStat,
                                        uint32_t litCtxArg, const uint32_t* __restrict__ rec, const uint16_t* __restrict__ rec3, const uint16_t* __restrict__ priceTab,
                                        uint32_t* __restrict__ recOut, uint32_t* __restrict__ winCost, const uint8_t* __restrict__ litPrice)
{
    __shared__ unsigned long long sCost[DPL_RC][64];
    __shared__ GcU4 sReps[REPS ? DPL_RR : 1u][64];
    constexpr uint32_t BPW = 2u;
    __shared__ uint16_t sPrice[BPW][GC_PRICE_WORDS - GC_PRICE_LEN];      // (the literal rows are not needed here: gc_mf_litprice_kernel has priced every position)
    __shared__ uint32_t sCnt[BPW][GC_DPS_WORDS];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t role = DPL_PAIR ? gc_uniform(threadIdx.x >> 6) : 0u;
    const bool isA = !DPL_PAIR || role == 0u, isB = !DPL_PAIR || role == 1u;      // wave 0 / wave 1 of the group (one wave: both)
    const bool isC = GC_DPL_THREADS == 192u ? role == 2u : isA;   // a third wave takes the literal, the capped rest and the short candidate off wave 0
    const uint32_t litCtxMask = litCtxArg & 0xFFu, hasPrev = litCtxArg >> 31;
    const bool win2k = !SAMPLE && (phaseArg & 16u) != 0u;         // every window of ONE block per wave, 64 x 2 KiB (more, shorter waves: the launch ends with its slowest wave)
    const bool allLenArg = (phaseArg & GC_DP_ALLLEN) != 0u;       // every length of a candidate is an edge (zstd levels >= 18)
    const bool selective = !SAMPLE && (phaseArg & GC_DP_SELECT) != 0u;      // phase B of the blocks whose sampled paths repeat distances; the others are W7's (gc_mf.h GC_DPS_RICH)
    phaseArg &= 15u;
    const uint32_t BPWr = win2k ? 1u : BPW;
    const bool phaseA = SAMPLE && phaseArg == 0u, phaseB = phaseArg == 1u;      // (the sample kernels count: phase A; the others run phase B, or phase 2 = W6's prices as they are)
    const uint32_t item = dpl_item(blockIdx.x, per);
    if (item * BPWr >= nBlocks) return;                           // (uniform)
    // ---- this lane's window
    const uint32_t lb = win2k ? 0u : lane >> 5;                   // block of the wave
    const uint32_t b = item * BPWr + lb;
    bool mineSel = true;
    if (selective) {                                              // (both blocks' counts are read by every lane: the wave's decisions are uniform)
        const bool s0 = GC_DPS_RICH(dpStat + (uint64_t)(item * BPWr) * GC_DPS_WORDS);
        const bool s1 = BPWr > 1u && item * BPWr + 1u < nBlocks && GC_DPS_RICH(dpStat + (uint64_t)(item * BPWr + 1u) * GC_DPS_WORDS);
        if (!s0 && !s1) return;
        mineSel = lb == 0u ? s0 : s1;
    }
    const bool blockLive = b < nBlocks && mineSel;
    const uint64_t base = (uint64_t)(blockLive ? b : 0u) * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = blockLive ? (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX) : 0u;
#ifdef HIPEMU
    static const int xWin4k = getenv("GC_X_WIN4K") ? atoi(getenv("GC_X_WIN4K")) : 0, xNoHint = getenv("GC_X_NOHINT") ? atoi(getenv("GC_X_NOHINT")) : 0, xWarm = getenv("GC_X_WARM") ? atoi(getenv("GC_X_WARM")) : DPL_WARM;
    static const int xSparse = getenv("GC_X_SPARSE") ? atoi(getenv("GC_X_SPARSE")) : 0;
    static const int xHints = getenv("GC_X_HINTS") ? atoi(getenv("GC_X_HINTS")) : 63;
    static const bool xSrepAny = getenv("GC_X_SREP_ANY") ? atoi(getenv("GC_X_SREP_ANY")) != 0 : DPL_SREP_ANY != 0;
    (void)xWin4k;
    static const bool allLenEnv = getenv("GC_X_ALL_LEN") ? atoi(getenv("GC_X_ALL_LEN")) != 0 : true;
    const bool allLen = DPL_ALL_LENGTHS != 0 && allLenArg && allLenEnv;
    const uint32_t winLen = SAMPLE ? 512u : (win2k ? 2048u : 4096u);
    const uint32_t w0 = win2k ? lane << 11 : ((lane & 31u) << 12) + (SAMPLE ? 1536u : 0u);
#else
    const int xWarm = DPL_WARM; const int xSparse = 0; const bool xSrepAny = DPL_SREP_ANY != 0;
    const bool allLen = DPL_ALL_LENGTHS != 0 && allLenArg;
    const uint32_t winLen = SAMPLE ? 512u : (win2k ? 2048u : 4096u);
    const uint32_t w0 = win2k ? lane << 11 : ((lane & 31u) << 12) + (SAMPLE ? 1536u : 0u);
#endif
    const uint32_t n = w0 < blockLen ? ((blockLen - w0) < winLen ? (blockLen - w0) : winLen) : 0u;       // nodes 0 .. n
    const uint32_t nMax = gc_wave_max(n);
    // ---- price tables (as W7: W6's table, phase A with optimistic ceilings where the greedy parse found no matches, phase B from phase A's counts); by wave 0
    if (isA) {
    for (uint32_t q = 0; q < BPW; q++) {
        const uint32_t bb = q < BPWr ? item * BPWr + q : nBlocks;
        if (bb < nBlocks) { const GcU4* T4 = (const GcU4*)(priceTab + (uint64_t)bb * GC_PRICE_WORDS + GC_PRICE_LEN); GcU4* S4 = (GcU4*)sPrice[q]; for (uint32_t i = lane; i < (GC_PRICE_WORDS - GC_PRICE_LEN) / 8u; i += 64u) S4[i] = T4[i]; }
        if (phaseA) for (uint32_t i = lane; i < GC_DPS_WORDS; i += 64u) sCnt[q][i] = 0;
    }
    gc_wave_sync();
    for (uint32_t q = 0; q < BPW; q++) {
        const uint32_t bb = q < BPWr ? item * BPWr + q : nBlocks;
        if (bb >= nBlocks) continue;                              // (uniform)
        uint16_t* P = sPrice[q] - GC_PRICE_LEN;                   // (indexed with the table's own offsets, all >= GC_PRICE_LEN)
        if (lane == 0u) {                                         // before anything is known: repeats 3 / 3.5 / 4 / 5 / 5 bits on top of the match flag, "no repeat" free
            P[GC_PRICE_FLAGS + 2u] = 48u; P[GC_PRICE_FLAGS + 3u] = 56u; P[GC_PRICE_FLAGS + 4u] = 64u; P[GC_PRICE_FLAGS + 5u] = 80u; P[GC_PRICE_FLAGS + 6u] = 80u; P[GC_PRICE_FLAGS + 7u] = 0u;
        }
        // lengths of repeats before anything is known: as those of matches, but the short ones (which hardly occur among the finder's matches) at 2.5 bits
        for (uint32_t i = lane; i < GC_PRICE_NLEN; i += 64u) { const uint32_t v = P[GC_PRICE_LEN + i]; P[GC_PRICE_REPLEN + i] = (uint16_t)(i < 10u && v > 40u ? 40u : v); }
        gc_wave_sync();
        const bool ceilings = phaseA && P[GC_PRICE_FLAGS + 1u] >= 64u;      // (read by every lane BEFORE the loop below rewrites that entry: lanes that run one after
        gc_wave_sync();                                                     //  another -- the emulator -- would otherwise stop capping from lane 17 on)
        if (ceilings) {
            for (uint32_t i = lane; i < GC_PRICE_NLEN + 64u + 1u; i += 64u) {
                const uint32_t idx = i < GC_PRICE_NLEN ? GC_PRICE_LEN + i : (i < GC_PRICE_NLEN + 64u ? GC_PRICE_SLOT + (i - GC_PRICE_NLEN) : GC_PRICE_FLAGS + 1u);
                const uint32_t cap = i < 10u ? 8u : (i < GC_PRICE_NLEN ? 0xFFFFu : (i < GC_PRICE_NLEN + 64u ? 48u : 16u));
                if (P[idx] > cap) P[idx] = (uint16_t)cap;
            }
        }
        if (phaseB) {
            const uint32_t* C = dpStat + (uint64_t)bb * GC_DPS_WORDS;
            const uint32_t nLit = C[GC_DPS_NLIT], nMat = C[GC_DPS_NMAT];
            if (nLit + nMat != 0u) {
                const uint32_t nRepAll = REPS ? C[GC_DPS_NREP] + C[GC_DPS_NSREP] + C[GC_DPS_NREP1] + C[GC_DPS_NREP2] + C[GC_DPS_NREP3] : 0u;
                const uint32_t nNew = nMat > nRepAll ? nMat - nRepAll : 0u;      // matches at a new distance: what lengths (of the match coder) and slots are counted over
                for (uint32_t i = lane; i < GC_PRICE_NLEN + 64u + 2u; i += 64u) {
                    if (i < GC_PRICE_NLEN) P[GC_PRICE_LEN + i] = (uint16_t)pz_price(8u * C[GC_DPS_LEN + i] + 1u, 8u * nNew + 63u);
                    else if (i < GC_PRICE_NLEN + 64u) P[GC_PRICE_SLOT + (i - GC_PRICE_NLEN)] = (uint16_t)pz_price(8u * C[GC_DPS_SLOT + (i - GC_PRICE_NLEN)] + 1u, 8u * nNew + 44u);
                    else if (i == GC_PRICE_NLEN + 64u) P[GC_PRICE_FLAGS] = (uint16_t)pz_price(nLit + 1u, nLit + nMat + 2u);
                    else P[GC_PRICE_FLAGS + 1u] = (uint16_t)pz_price(nMat + 1u, nLit + nMat + 2u);
                }
                if (REPS) {
                    uint32_t nr = 0; for (uint32_t i = 0; i < GC_PRICE_NLEN; i++) nr += C[GC_DPS_REPLEN + i];
                    if (nr >= 16u) for (uint32_t i = lane; i < GC_PRICE_NLEN; i += 64u) P[GC_PRICE_REPLEN + i] = (uint16_t)pz_price(8u * C[GC_DPS_REPLEN + i] + 1u, 8u * nr + 63u);
                }
                if (REPS && lane == 0u) {
                    // IsRep, IsRepG0, IsRep0Long, IsRepG1, IsRepG2 (LzmaEnc.c / lzma2_enc.c:289 LZMA_getRepPrice) from how often phase A's paths used each
                    const uint32_t n0 = C[GC_DPS_NREP], nS = C[GC_DPS_NSREP], n1 = C[GC_DPS_NREP1], n2 = C[GC_DPS_NREP2], n3 = C[GC_DPS_NREP3];
                    const uint32_t all = n0 + nS + n1 + n2 + n3;
                    const uint32_t isRep = pz_price(all + 1u, nMat + 2u), g0 = pz_price(n0 + nS + 1u, all + 2u), g0n = pz_price(n1 + n2 + n3 + 1u, all + 2u);
                    P[GC_PRICE_FLAGS + 2u] = (uint16_t)(isRep + g0 + pz_price(n0 + 1u, n0 + nS + 2u));
                    P[GC_PRICE_FLAGS + 3u] = (uint16_t)(isRep + g0 + pz_price(nS + 1u, n0 + nS + 2u));
                    P[GC_PRICE_FLAGS + 4u] = (uint16_t)(isRep + g0n + pz_price(n1 + 1u, n1 + n2 + n3 + 2u));
                    const uint32_t g1n = pz_price(n2 + n3 + 1u, n1 + n2 + n3 + 2u);
                    P[GC_PRICE_FLAGS + 5u] = (uint16_t)(isRep + g0n + g1n + pz_price(n2 + 1u, n2 + n3 + 2u));
                    P[GC_PRICE_FLAGS + 6u] = (uint16_t)(isRep + g0n + g1n + pz_price(n3 + 1u, n2 + n3 + 2u));
                    P[GC_PRICE_FLAGS + 7u] = (uint16_t)pz_price(nNew + 1u, nMat + 2u);      // "no repeat"
                }
            }
        }
    }
    gc_wave_sync();
    }
    DPL_BARRIER();
    const uint16_t* P = sPrice[lb] - GC_PRICE_LEN;
    // Tracking the repeat distances is half of a node's work.  Where phase A's paths of BOTH blocks of the wave hardly ever repeated a distance (text: one
    // match symbol in 30) phase B runs without it: repeats are then only found where a candidate of the finder has a repeat distance.
    bool trackOn = REPS;
    if (REPS && phaseB && !selective) {
        uint32_t rich = 0u;
        for (uint32_t q = 0; q < BPW; q++) {
            const uint32_t bb = q < BPWr ? item * BPWr + q : nBlocks;
            if (bb >= nBlocks) continue;
            const uint32_t* C = dpStat + (uint64_t)bb * GC_DPS_WORDS;
            const uint32_t reps = C[GC_DPS_NREP] + C[GC_DPS_NSREP] + C[GC_DPS_NREP1] + C[GC_DPS_NREP2] + C[GC_DPS_NREP3];
            if (reps * 20u >= C[GC_DPS_NMAT] || C[GC_DPS_NMAT] == 0u) rich = 1u;
        }
        trackOn = gc_uniform(rich) != 0u;
    }
    const uint32_t flagLit = P[GC_PRICE_FLAGS], flagMat = P[GC_PRICE_FLAGS + 1u];
    const uint32_t newAdd = flagMat + (REPS ? (uint32_t)P[GC_PRICE_FLAGS + 7u] : 0u);
    const uint32_t fRep0 = flagMat + P[GC_PRICE_FLAGS + 2u], fSrep = flagMat + P[GC_PRICE_FLAGS + 3u], fRep1 = flagMat + P[GC_PRICE_FLAGS + 4u], fRep2 = flagMat + P[GC_PRICE_FLAGS + 5u], fRep3 = flagMat + P[GC_PRICE_FLAGS + 6u];      // (in registers: an LDS read in the node loop is a stall of a lone wave)
    const uint8_t* S = src + base + w0;                           // window-relative addressing (positions in front of the window are negative)
    const uint32_t* R = rec + base + w0;
    const uint16_t* R3 = rec3 + base + w0;
    uint32_t* BP = recOut + base + w0;
    const uint64_t absW = base + w0;
    const uint64_t tailRoom = srcSize - absW;                     // bytes of the input from the window start on
    // the programme starts `warm` positions in front of the window (a multiple of four, inside the window's frame)
    int32_t warm = 0;
    int64_t inFrameW = 0;                                         // bytes of the window's frame in front of the window: a distance d reaches them from position q iff q + inFrameW >= d
    if (n != 0u) {
        const uint64_t frameBytes = (uint64_t)frameBlocks * GC_ZSTD_BLOCK_MAX;
        const uint64_t inFrame = absW % frameBytes;
        inFrameW = (int64_t)inFrame;
        warm = inFrame < (uint64_t)xWarm ? (int32_t)inFrame : xWarm;
        warm &= ~3;
    }
    const int32_t warmMax = (int32_t)gc_wave_max((uint32_t)warm);
    const int32_t N = (int32_t)n;

    // ring slots of node i: i mod DPL_RC / i mod DPL_RR, kept as counters of the (uniform) node loop
    uint32_t si = (uint32_t)(((-warmMax) % (int32_t)DPL_RC + (int32_t)DPL_RC) % (int32_t)DPL_RC), sr = (uint32_t)(((-warmMax) % (int32_t)DPL_RR + (int32_t)DPL_RR) % (int32_t)DPL_RR);
    if (isA) {
        for (uint32_t s = 0; s < DPL_RC; s++) sCost[s][lane] = DPL_INF;
        sCost[(si + (uint32_t)(warmMax - warm)) % DPL_RC][lane] = 0ull;          // the first node: cost 0
        if (REPS) { GcU4 v; v.x = v.y = v.z = v.w = 0u; for (uint32_t s = 0; s < DPL_RR; s++) sReps[s][lane] = v; }
    }
    DPL_BARRIER();
    DplReps st; st.r0 = st.r1 = st.r2 = st.r3 = 0u;               // repeat distances of the node being expanded

    // ---- pipelines: records, short candidates and literal prices two groups of four positions ahead
    uint32_t recG[4], recN[4], recNN[4], r3G[4], r3N[4], r3NN[4], lpG, lpN, lpNN;
    const uint8_t* LP = litPrice + base + w0;
    // tracked distances (REPS): where the window's bytes repeat at distance tD[k] -- bit j of tM[k]: S[pb + j] == S[pb + j - tD[k]], known for
    // positions below tE[k].  One slot is (re)filled per group of four nodes, 32 positions at a time, from two 32-byte reads issued a group earlier.
    uint32_t tD[DPL_NT]; int32_t tE[DPL_NT]; unsigned long long tM[DPL_NT];
#pragma unroll
    for (uint32_t k = 0; k < DPL_NT; k++) { tD[k] = 0u; tE[k] = 0; tM[k] = 0ull; }
    uint32_t lru[2] = { 0u, 0u }; int32_t sNext = -warm;         // the shadow parse: the last two distances of a greedy walk over the finder's records
    int32_t pb = 0;                                               // position of bit 0 of the masks (a multiple of four)
    // in flight, one request for the node's distances (slots 0..3) and one for the shadow parse's (slots 4, 5): slot (DPL_NT = none), its distance, first position, the bytes
    uint32_t fK[DPL_NC], fD[DPL_NC]; int32_t fQ[DPL_NC]; LzW16 fOwn[DPL_NC][2], fOth[DPL_NC][2];
#pragma unroll
    for (uint32_t c = 0; c < DPL_NC; c++) { fK[c] = DPL_NT; fD[c] = 0u; fQ[c] = 0; fOwn[c][0].a = fOwn[c][0].b = fOwn[c][1].a = fOwn[c][1].b = fOth[c][0].a = fOth[c][0].b = fOth[c][1].a = fOth[c][1].b = 0ull; }
    // group loader: records, short candidates and bytes of positions g4 .. g4 + 3 (g4 a multiple of four, possibly negative)
    auto load_group = [&](int32_t g4, uint32_t (&rr)[4], uint32_t (&r3)[4], uint32_t& by) {
#pragma unroll
        for (uint32_t u = 0; u < 4u; u++) { rr[u] = 0u; r3[u] = GC_SHORT_NONE; }
        by = 0u;
        if (g4 >= -warm && g4 + 4 <= N) {
            GcU4 v; __builtin_memcpy(&v, R + g4, 16); rr[0] = v.x; rr[1] = v.y; rr[2] = v.z; rr[3] = v.w;
            uint64_t h; __builtin_memcpy(&h, R3 + g4, 8); r3[0] = (uint32_t)h & 0xFFFFu; r3[1] = (uint32_t)(h >> 16) & 0xFFFFu; r3[2] = (uint32_t)(h >> 32) & 0xFFFFu; r3[3] = (uint32_t)(h >> 48);
            by = gc_ld32(LP + g4);
        } else if (g4 >= -warm) {
#pragma unroll
            for (int32_t u = 0; u < 4; u++) if (g4 + u < N) { rr[u] = R[g4 + u]; r3[u] = R3[g4 + u]; by |= (uint32_t)LP[g4 + u] << (8 * u); }
        }
    };
    // bytes of x that are zero, as a mask of 4 bits / of two 16-byte strings that are equal, as a mask of 16 bits
    auto zero4 = [](uint32_t x) -> uint32_t { uint32_t y = x | (x >> 4); y |= y >> 2; y |= y >> 1; y = ~y & 0x01010101u; return (y * 0x10204080u) >> 28; };
    auto eq16 = [&](const LzW16& x, const LzW16& y) -> uint32_t {
        const unsigned long long d0 = x.a ^ y.a, d1 = x.b ^ y.b;
        return zero4((uint32_t)d0) | (zero4((uint32_t)(d0 >> 32)) << 4) | (zero4((uint32_t)d1) << 8) | (zero4((uint32_t)(d1 >> 32)) << 12);
    };
    // between two groups of nodes (q0 = first position of the next group): what was requested a group ago arrives, the masks move on, the distances of the
    // newest final node that are not tracked yet take the slots of those that are none of its distances, and the slot that runs out first is refilled
    // the shadow parse moves over the records rr of positions g .. g + 3
    auto shadow_step = [&](int32_t g, const uint32_t (&rr)[4]) {
#pragma unroll
        for (int32_t u = 0; u < 4; u++) {
            const int32_t q = g + u;
            if (q >= sNext && q >= -warm && q < N) { const uint32_t L = rr[u] & 0xFFu; if (L >= 3u) { const uint32_t d = rr[u] >> 8; if (lru[0] != d) { lru[1] = lru[0]; lru[0] = d; } sNext = q + (int32_t)L; } else sNext = q + 1; }
        }
    };
    // between two groups of nodes (q0 = first position of the next group): what was requested a group ago arrives, the masks move on, wanted distances
    // that are not tracked take the slots of those nobody wants, and per request channel the slot that runs out first is refilled
    auto track_step = [&](int32_t q0) {
#pragma unroll
        for (uint32_t c = 0; c < DPL_NC; c++) {
            if (fK[c] < DPL_NT) {
#pragma unroll
                for (uint32_t k = 0; k < DPL_NT; k++) if (k == fK[c] && tD[k] == fD[c]) {
                    const unsigned long long bits = (unsigned long long)(eq16(fOwn[c][0], fOth[c][0]) | (eq16(fOwn[c][1], fOth[c][1]) << 16));
                    const uint32_t sh = (uint32_t)(fQ[c] - pb);
                    tM[k] = (tM[k] & ((1ull << sh) - 1ull)) | (bits << sh);
                    tE[k] = fQ[c] + 32;
                }
                fK[c] = DPL_NT;
            }
        }
        if (q0 - pb >= 32) {
            pb += 32;
#pragma unroll
            for (uint32_t k = 0; k < DPL_NT; k++) tM[k] >>= 32;
        }
        const uint32_t want[DPL_NT] = { st.r0 & ~DPL_SURE, st.r1, st.r2, st.r3, lru[0], lru[1] };
        uint32_t fillD = 0u, fillK = DPL_NT;                      // a distance of the node that gets its first 32 positions on the spot (below)
#pragma unroll
        for (uint32_t j = 0; j < DPL_NT; j++) {
            const uint32_t d = want[j];
            bool have = d == 0u;
#pragma unroll
            for (uint32_t k = 0; k < DPL_NT; k++) have = have || tD[k] == d;
            if (have) continue;
            uint32_t v = DPL_NT;                                  // a slot of its kind whose distance nobody wants
#pragma unroll
            for (uint32_t k = 0; k < DPL_NT; k++) {
                const uint32_t t = tD[k];
                const bool mine = j < 4u ? k < 4u : k >= 4u;
                if (mine && v == DPL_NT && t != want[0] && t != want[1] && t != want[2] && t != want[3] && t != want[4] && t != want[5]) v = k;
            }
#pragma unroll
            for (uint32_t k = 0; k < DPL_NT; k++) if (k == v) { tD[k] = d; tM[k] = 0ull; tE[k] = q0; }
            if (j < 4u && v < DPL_NT && fillK == DPL_NT) { fillK = v; fillD = d; }
        }
        // A newly tracked distance of the NODE is wanted now (the nodes that made it a repeat are the ones about to use it): its first 32 positions are read on
        // the spot -- an exposed memory latency, a few times per hundred nodes and lane -- instead of leaving the next group of nodes without it (the evaluation
        // slices: ROCm shared objects +0.5 % without).  The shadow parse's distances come early; their request channel serves them.
        if (fillK < DPL_NT && q0 < N && q0 >= -warm && (uint64_t)((int64_t)q0 + 32) <= tailRoom && (int64_t)q0 + inFrameW >= (int64_t)fillD) {
            const LzW16 o0 = lz_ld16(S + q0, 0), o1 = lz_ld16(S + q0 + 16, 0);
            const LzW16 p0 = lz_ld16(S + (int64_t)q0 - (int64_t)fillD, 0), p1 = lz_ld16(S + (int64_t)q0 - (int64_t)fillD + 16, 0);
            const unsigned long long bits = (unsigned long long)(eq16(o0, p0) | (eq16(o1, p1) << 16)) << (uint32_t)(q0 - pb);
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) if (k == fillK) { tM[k] = bits; tE[k] = q0 + 32; }
        }
        uint32_t taken = DPL_NT;
#pragma unroll
        for (uint32_t c = 0; c < DPL_NC; c++) {
            uint32_t pick = DPL_NT; int32_t best = 0x7FFFFFFF;
#pragma unroll
            for (uint32_t k = 0; k < DPL_NT; k++) if ((c + 1u < DPL_NC ? k < 4u : k >= 4u) && k != taken && tD[k] != 0u && tE[k] < best) { best = tE[k]; pick = k; }
            if (c + 2u < DPL_NC) taken = pick;
            if (pick < DPL_NT && best - q0 < 40 && q0 < N) {
                const int32_t fq = best > q0 ? best : q0;
                uint32_t dPick = 0u;
#pragma unroll
                for (uint32_t k = 0; k < DPL_NT; k++) if (k == pick) dPick = tD[k];
                if (fq - pb <= 32 && (uint64_t)((int64_t)fq + 32) <= tailRoom && fq >= -warm && (int64_t)fq + inFrameW >= (int64_t)dPick) {
                    fK[c] = pick; fQ[c] = fq; fD[c] = dPick;
                    fOwn[c][0] = lz_ld16(S + fq, 0); fOwn[c][1] = lz_ld16(S + fq + 16, 0);
                    fOth[c][0] = lz_ld16(S + (int64_t)fq - (int64_t)fD[c], 0); fOth[c][1] = lz_ld16(S + (int64_t)fq - (int64_t)fD[c] + 16, 0);
#pragma unroll
                    for (uint32_t k = 0; k < DPL_NT; k++) if (k == pick && tE[k] < q0) tE[k] = q0;      // (nothing known in between)
                }
            }
        }
    };
    load_group(-warmMax, recG, r3G, lpG);
    load_group(-warmMax + 4, recN, r3N, lpN);
    pb = -warmMax;
    if (REPS) { shadow_step(-warmMax, recG); shadow_step(-warmMax + 4, recN); }
    uint32_t contDist = 0u, contRem = 0u; bool contCapped = false;   // the node being expanded was reached by a capped piece
    uint32_t c1 = 0u, c2 = 0u, c3 = 0u;                           // back pointers waiting to be stored (nodes 4g+1 .. 4g+3)
    uint32_t cost0 = 0u, costN = 0u;                              // cost of the window's first / last node
    // length prices of the lane's block in registers, two per word (the relax loops below run over static lengths)
    uint32_t lenR[DPL_TABW], repR[DPL_TABW];
#pragma unroll
    for (uint32_t k = 0; k < DPL_TABW; k++) {
        lenR[k] = (uint32_t)P[GC_PRICE_LEN + 2u * k] | ((uint32_t)P[GC_PRICE_LEN + 2u * k + 1u] << 16);
        repR[k] = REPS ? (uint32_t)P[GC_PRICE_REPLEN + 2u * k] | ((uint32_t)P[GC_PRICE_REPLEN + 2u * k + 1u] << 16) : 0u;
    }
    unsigned long long* const myCost = &sCost[0][lane];           // this lane's column: slot s at myCost[s * 64]

    // One candidate as edges: Lm bytes (0: none) at a distance, lengths x0 .. Lm.  hiBase = (cost of the node + flags and distance) << 7; the edge of Lm bytes
    // carries hiLast / loLast instead (length price of the whole match, bytes left, "capped").  Every lane runs the loop (static lengths, its own predicate);
    // the wave leaves it once no lane has a longer candidate.
#ifdef HIPEMU
#define DPL_NONE_LONGER(x, L) ((x) > (L))
#else
#define DPL_NONE_LONGER(x, L) (!__any((x) <= (L)))
#endif
    // The lengths tried are x0 .. LO and the last four (Lm - 3 .. Lm): what lies between are prefixes of a long candidate that end nowhere in particular
    // (measured on the evaluation slices: all lengths against 2 .. 12 + the last four of the finder's candidates, 2 .. 4 + the last four of a repeat:
    // +0.05 ... 0.2 % size for half the relax instructions).  tabBase: where the length prices lie in LDS (for the lengths that are not static).
    auto slotOf = [&](uint32_t x) -> uint32_t { const uint32_t t = si + x; return t >= DPL_RC ? t - DPL_RC : t; };      // ring slot of node i + x, x <= DPL_M + 1 (si: the slot of node i)
    auto relax = [&](int32_t, uint32_t Lm, uint32_t x0, uint32_t hiBase, uint32_t loBase, uint32_t lastLen /* whose price the last edge carries */, uint32_t left, bool capLast, uint32_t loLast,
                     const uint32_t (&tab)[DPL_TABW], uint32_t tabBase, bool flat, const uint32_t LO) {
        if (DPL_NONE_LONGER(x0 > 1u ? x0 : 1u, Lm)) return;
        // the four prices with a length of the lane's own: read together, used below
        const uint32_t p3 = P[tabBase + (Lm > 3u ? Lm - 3u : 0u)], p2 = P[tabBase + (Lm > 2u ? Lm - 2u : 0u)], p1 = P[tabBase + (Lm > 1u ? Lm - 1u : 0u)], p0 = P[tabBase + lastLen];
        const uint32_t pre = Lm > 4u ? (Lm - 4u < LO ? Lm - 4u : LO) : 0u;        // lengths up to here in the static loop
        // (every lane issues every ds_min: a lane without that edge sends the neutral word -- a select instead of an exec-mask branch per edge)
#pragma unroll
        for (uint32_t x = 1u; x <= LO; x++) {
            if (x > 1u && (x & 3u) == 1u && DPL_NONE_LONGER(x, pre)) break;
            const uint32_t lp = flat ? 0u : ((tab[x >> 1] >> (16u * (x & 1u))) & 0xFFFFu);
            const uint32_t hi = (x >= x0 && x <= pre) ? hiBase + (lp << DPL_CSH) : 0xFFFFFFFFu;
            atomicMin(&myCost[slotOf(x) * 64u], ((unsigned long long)hi << 32) | (loBase | (x - 1u)));
        }
        if (allLen) {                                              // every length between the static ones and the last four (zstd levels >= 16: ZSTD_compressBlock_opt_generic prices every length of every match, zstd_opt.c:1230-1250)
            for (uint32_t x = LO + 1u; !DPL_NONE_LONGER(x + 4u, Lm); x++) {
                const uint32_t lp = flat ? 0u : (uint32_t)P[tabBase + x];
                const uint32_t hi = (x >= x0 && x + 4u <= Lm) ? hiBase + (lp << DPL_CSH) : 0xFFFFFFFFu;
                atomicMin(&myCost[slotOf(x <= DPL_M ? x : 1u) * 64u], ((unsigned long long)hi << 32) | (loBase | ((x - 1u) & DPL_MMASK)));
            }
        }
#pragma unroll
        for (uint32_t t = 3u; t >= 1u; t--) {
            const uint32_t x = Lm - t;                             // (per lane)
            const uint32_t lp = flat ? 0u : (t == 3u ? p3 : (t == 2u ? p2 : p1));
            const uint32_t hi = (Lm > t && x >= x0 && x > pre) ? hiBase + (lp << DPL_CSH) : 0xFFFFFFFFu;
            atomicMin(&myCost[slotOf(Lm > t ? x : 1u) * 64u], ((unsigned long long)hi << 32) | (loBase | ((x - 1u) & DPL_MMASK)));
        }
        {
            const uint32_t hi = (Lm >= x0 && Lm != 0u) ? ((hiBase + ((flat ? 0u : p0) << DPL_CSH)) | left | (capLast ? DPL_HI_CAP : 0u)) : 0xFFFFFFFFu;
            atomicMin(&myCost[slotOf(Lm) * 64u], ((unsigned long long)hi << 32) | loLast);
        }
    };

    // ---- the programme: node i = finalize (i > -warm) + expand (i < n); nodes below 0 are the warm-up.  The records of the group of four positions
    //      that holds i sit in recG / r3G / lpG and move down one place per node, so that the node's own are always in place 0
#if defined(DPL_PROF) && !defined(HIPEMU)
    unsigned long long pacc[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }, ptick = clock64();
#endif
    for (int32_t i = -warmMax; i < (int32_t)nMax + 4; i++) {
        const uint32_t u = (uint32_t)i & 3u;
        DPL_T(7);
        if (u == 0u) load_group(i + 8, recNN, r3NN, lpNN);
        const unsigned long long w = myCost[si * 64u];
        if (isA && i != -warmMax) myCost[(si == 0u ? DPL_RC - 1u : si - 1u) * 64u] = DPL_INF;      // node i - 1's slot becomes node i + DPL_RC - 1's (both waves have read it: the barrier of step i - 1)
        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
        const uint32_t c0 = hi >> DPL_CSH;
        const bool live = n != 0u && i >= -warm && i <= N;
        if (live && i > -warm) {                                   // ---- finalize node i
            const uint32_t len = DPL_LO_LEN(lo), cls = DPL_LO_CLS(lo), dist = DPL_LO_DIST(lo);
            if (REPS) {
                uint32_t ps = sr + DPL_RR - (len < DPL_MR ? len : DPL_MR); ps = ps >= DPL_RR ? ps - DPL_RR : ps;
                const GcU4 pv = sReps[ps][lane];                  // (an edge longer than the ring: the oldest node it still holds)
                st.r0 = pv.x; st.r1 = pv.y; st.r2 = pv.z; st.r3 = pv.w;
                if (cls != DPL_LIT && cls != DPL_SREP) dpl_mtf(st, dist, i - (int32_t)len >= 0 ? DPL_SURE : 0u);
                if (isA) { GcU4 nv; nv.x = st.r0; nv.y = st.r1; nv.z = st.r2; nv.w = st.r3; sReps[sr][lane] = nv; }
            }
            contCapped = (hi & DPL_HI_CAP) != 0u; contDist = dist; contRem = hi & 63u;
            if (i == 0) cost0 = c0;
            if (i == N) costN = c0;
        }
        DPL_T(0);
        // back pointers: node j lives in slot j - 1; nodes 4g-3 .. 4g leave together
        if (!isA) { }
        else if (u == 0u) {
            if (i >= 4 && i - 4 < N) {
                if (i <= N) { GcU4 v; v.x = c1; v.y = c2; v.z = c3; v.w = lo; __builtin_memcpy(BP + (i - 4), &v, 16); }
                else { BP[i - 4] = c1; if (i - 3 < N) BP[i - 3] = c2; if (i - 2 < N) BP[i - 2] = c3; }
            }
        } else { c1 = u == 1u ? lo : c1; c2 = u == 2u ? lo : c2; c3 = u == 3u ? lo : c3; }
        {                                                          // ---- expand node i (room = 0: a lane without this node)
            // bytes up to the end of the window -- in the warm-up: up to the window's first node, where the lane of the window in front stops too
            const uint32_t room = !live ? 0u : (i < 0 ? (uint32_t)(-i) : (uint32_t)(N - i));
            // a continuation of one byte is only a match together with the piece in front of it: at the window's first node that piece is the warm-up's guess
            const uint32_t contX0 = i == 0 ? (MINLEN > 2u ? MINLEN : 2u) : 1u;
            const uint32_t cbase = c0 << DPL_CSH;
            if (isC) atomicMin(&myCost[slotOf(1u) * 64u], (unsigned long long)(room != 0u ? cbase + ((flagLit + (lpG & 0xFFu)) << DPL_CSH) : 0xFFFFFFFFu) << 32);      // literal
            // the rest of a capped match whose length is known (handing these two to wave 1 as well -- 23.5 : 19.4 of the pair's work are wave 0's -- changed nothing: 23.85 -> 23.72 ms)
            if (isC) {
                uint32_t Lc = (room != 0u && contCapped) ? contRem : 0u;
                const uint32_t beh = Lc > room ? Lc - room : 0u;
                if (Lc > room) Lc = room;
                const uint32_t Lm = Lc < DPL_M ? Lc : DPL_M;
                uint32_t left = Lc - Lm + beh; if (left > 63u) left = 63u;
                const uint32_t hb = cbase + (DPL_CONT << DPL_CSH);
                relax(i, Lm, contX0, hb, DPL_LO(contDist, DPL_CONTC, 1u), 0u, left, left != 0u, DPL_LO(contDist, DPL_CONTC, Lm ? Lm : 1u), lenR, GC_PRICE_LEN, true, 4u);
            }
            DPL_T(1);
            // finder candidate, short candidate
            const uint32_t r = recG[0], r3 = r3G[0];
            uint32_t L = r & 0xFFu; const uint32_t D = r >> 8;
            uint32_t behL = L > room ? L - room : 0u;
            if (L > room) L = room;
            uint32_t L3 = 0u, D3 = 0u;
            if (r3 != GC_SHORT_NONE) { L3 = (r3 & 15u) + 2u; D3 = (r3 >> 4) + 1u; if (L3 > room) L3 = room; if (L3 > DPL_M) L3 = DPL_M; if (L3 < MINLEN || (L >= L3 && D <= D3)) L3 = 0u; }
            if (L < MINLEN && !(contCapped && D == contDist && L != 0u)) { L = 0u; behL = 0u; }
            {
#pragma unroll
            for (uint32_t cnd = 0; cnd < 2u; cnd++) {
                if (!(cnd ? isC : isA)) continue;
                const uint32_t Lx = cnd ? L3 : L, Dx = cnd ? D3 : (D ? D : 1u), behind = cnd ? 0u : behL;
                uint32_t cls = DPL_NEW, x0 = MINLEN;
                const uint32_t sl = gc_dist_slot(Dx - 1u);
                uint32_t add = newAdd + P[GC_PRICE_SLOT + sl] + (sl >= 4u ? 16u * ((sl >> 1) - 1u) : 0u);
                if (REPS) { const uint32_t k = dpl_which(st, Dx); if (k < 4u) { cls = DPL_REP0 + k; add = k == 0u ? fRep0 : (k == 1u ? fRep1 : (k == 2u ? fRep2 : fRep3)); } }
                const bool isCont = contCapped && Dx == contDist;
                if (isCont) { cls = DPL_CONTC; add = DPL_CONT; x0 = contX0; }
                const uint32_t Lm = Lx < DPL_M ? Lx : DPL_M;
                uint32_t left = Lx - Lm + behind; if (left > 63u) left = 63u;
                const bool openEnd = !cnd && (r & 0xFFu) == GC_MATCH_CAP;      // a capped record goes on in the record behind it
                const uint32_t wholeLen = Lx + behind < GC_MATCH_CAP ? Lx + behind : GC_MATCH_CAP;
                const bool repTab = REPS && cls >= DPL_REP0 && cls <= DPL_REP0 + 3u;
                const uint32_t hb = cbase + (add << DPL_CSH);
                const uint32_t loLast = DPL_LO(Dx, cls, Lm ? Lm : 1u), lastLen = left != 0u ? wholeLen : Lm;      // the last piece: the length price of the whole match
                const bool capLast = left != 0u || openEnd;
                if (repTab) relax(i, Lm, x0, hb, DPL_LO(Dx, cls, 1u), lastLen, left, capLast, loLast, repR, GC_PRICE_REPLEN, isCont, cnd ? 6u : 8u);
                else relax(i, Lm, x0, hb, DPL_LO(Dx, cls, 1u), lastLen, left, capLast, loLast, lenR, GC_PRICE_LEN, isCont, cnd ? 6u : 8u);
            }
            }
            DPL_T(2);
            // repeats of the node's own distances, where those are tracked.  Of the (up to four) that repeat here two become edges: the one with the
            // lowest repeat index -- the cheapest flags -- and the longest one (evaluation slices: against all of them +0.0x % size, a third of the work)
            if (REPS && trackOn && isB) {
                uint32_t bestK = 8u, bestL = 0u, bestD = 0u, longK = 8u, longL = 0u, longD = 0u, srepD = 0u; bool bestOpen = false, longOpen = false;
                const uint32_t r0d = st.r0 & ~DPL_SURE;
#pragma unroll
                for (uint32_t k = 0; k < DPL_NT; k++) {
                    const uint32_t hd = tD[k];
                    const int32_t avail = tE[k] - i;                               // positions from here on that the mask knows
                    const uint32_t av = avail > 0 ? (uint32_t)avail : 0u;
                    const unsigned long long run = ~(tM[k] >> (uint32_t)(i - pb)) | (1ull << 63);      // (straight-line: selects, no branch per slot)
                    uint32_t hl = gc_ctz64(run);
                    hl = hl > av ? av : hl;
                    hl = (hd != 0u && room != 0u) ? hl : 0u;
#ifdef HIPEMU
                    if (xNoHint) hl = 0u;
                    for (uint32_t z = 0; z < hl; z++) if (S[i + (int32_t)z] != S[(int64_t)i + z - (int64_t)hd]) { fprintf(stderr, "W7L: mask of distance %u wrong at %d + %u\n", hd, i, z); abort(); }
#endif
                    const bool open = hl != 0u && ((avail > 0 && hl == (uint32_t)avail) || hl >= DPL_M) && hl <= room;      // the run may go on behind what is known / what an edge holds
                    if (hl > DPL_M) hl = DPL_M;
                    if (hl > room) hl = room;
                    // which repeat of the node it is: 0..3; the continuation of a capped piece counts as the cheapest (index 0 - 1)
                    uint32_t kk = hd == r0d ? 1u : (hd == st.r1 ? 2u : (hd == st.r2 ? 3u : (hd == st.r3 ? 4u : 8u)));
                    if (contCapped && hd == contDist) kk = 0u;
                    if (MINLEN == 2u && hl != 0u && kk == 1u && (xSrepAny || (st.r0 & DPL_SURE))) srepD = hd;      // LZMA's short repeat: one byte at rep0 (round 4: only where rep0 is known for certain -- a match of this window on
                                                                                                                  // the way; round 5: L2 codes a short repeat that names another distance than its rep0 as a literal, so the distance the window arrived with will do)
                    const uint32_t covered = hd == D ? L : (hd == D3 ? L3 : 0u);                                // the candidate itself covers it
                    const bool use = hl >= (kk == 0u ? contX0 : MINLEN) && kk != 8u && covered < hl;
                    const bool b1 = use && kk < bestK, b2 = use && hl > longL;
                    bestK = b1 ? kk : bestK; bestL = b1 ? hl : bestL; bestD = b1 ? hd : bestD; bestOpen = b1 ? open : bestOpen;
                    longK = b2 ? kk : longK; longL = b2 ? hl : longL; longD = b2 ? hd : longD; longOpen = b2 ? open : longOpen;
                }
                if (MINLEN == 2u) atomicMin(&myCost[slotOf(1u) * 64u], ((unsigned long long)(srepD ? cbase + (fSrep << DPL_CSH) : 0xFFFFFFFFu) << 32) | DPL_LO(srepD, DPL_SREP, 1u));
                if (longK == bestK) longL = 0u;
#pragma unroll
                for (uint32_t e = 0; e < 2u; e++) {
                    const uint32_t kk = e ? longK : bestK, hl = e ? longL : bestL, hd = e ? longD : bestD; const bool open = e ? longOpen : bestOpen;
                    const bool isCont = kk == 0u;
                    const uint32_t ri = (kk - 1u) & 3u;                            // repeat index
                    const uint32_t cls = isCont ? DPL_CONTC : DPL_REP0 + ri, add = isCont ? DPL_CONT : (ri == 0u ? fRep0 : (ri == 1u ? fRep1 : (ri == 2u ? fRep2 : fRep3)));
                    const uint32_t hb = cbase + (add << DPL_CSH);
                    relax(i, hl, isCont ? contX0 : MINLEN, hb, DPL_LO(hd, cls, 1u), hl, 0u, open, DPL_LO(hd, cls, hl ? hl : 1u), repR, GC_PRICE_REPLEN, isCont, 4u);
                }
            }
        }
        DPL_T(3);
        // the group's records move down one place
        recG[0] = recG[1]; recG[1] = recG[2]; recG[2] = recG[3]; r3G[0] = r3G[1]; r3G[1] = r3G[2]; r3G[2] = r3G[3]; lpG >>= 8;
        if (u == 3u) {                                             // ---- between two groups: the shadow parse and the tracked distances move on, the next group's records come in
            if (REPS && trackOn && isB) { shadow_step(i + 5, recNN); track_step(i + 1); }
#pragma unroll
            for (uint32_t q = 0; q < 4u; q++) { recG[q] = recN[q]; recN[q] = recNN[q]; r3G[q] = r3N[q]; r3N[q] = r3NN[q]; }
            lpG = lpN; lpN = lpNN;
            DPL_T(4);
        }
        si = si + 1u == DPL_RC ? 0u : si + 1u; sr = sr + 1u == DPL_RR ? 0u : sr + 1u;
        DPL_T(5);
        DPL_BARRIER();                                             // every edge out of node i is in LDS before either wave reads node i + 1
        DPL_T(9);
    }
#if defined(DPL_PROF) && !defined(HIPEMU)
    if (lane == 0u && !SAMPLE) for (int k = 0; k < 10; k++) atomicAdd(&g_dplProf[16u * role + k], pacc[k]);
    ptick = clock64();
#endif
    if (!isA) return;                                              // wave 1 is done: the rest (window costs, walk back, counts) is wave 0's
#undef DPL_NONE_LONGER
    // back pointers of the last nodes (n not a multiple of four is covered above; n a multiple of four: nodes n-3 .. n left at g4 = n)
    gc_wave_sync_global();
    {                                                             // estimate per 4 KiB range-coder chunk (2 KiB windows: the sum of two lanes).
        // Straight-line on purpose: as `if (!win2k) store A else store B` behind the shuffle, hipcc 7.2 emitted the two stores under each other's condition
        // (seen in the ISA and in the output: half of the entries written, with the sum of two windows) -- L2 then stored segments by whatever the rest held
        const uint32_t mine = costN - cost0, other = __shfl_xor(mine, 1);
        const uint32_t idx = win2k ? lane >> 1 : lane & 31u, val = win2k ? mine + other : mine;
        const bool wr = winCost != nullptr && !SAMPLE && w0 < blockLen && (!win2k || (lane & 1u) == 0u);
        if (wr) winCost[(uint64_t)b * (GC_ZSTD_BLOCK_MAX >> 12) + idx] = val;
    }
    // ---- walk back, slot by slot in lockstep: slot q holds the back pointer of node q + 1 and receives the record of position q
    uint32_t j = n;                                               // end node of the edge the walk is inside of (slots s .. j - 1)
    uint32_t s = n, eDist = 0u, eCls = 0u;                        // its start node, distance, class of its FIRST piece so far
    uint32_t runEnd = 0u;                                         // end of the run of pieces with one distance that the edge belongs to
    uint32_t nextLo = n ? BP[n - 1u] : 0u;                        // back pointer of node j (read one slot early)
    uint32_t nRecs = 0u, nLitC = 0u;
    for (uint32_t it = 0; it < nMax; it++) {
        const uint32_t q = n - 1u - it;                           // (wraps for finished lanes)
        if (it < n) {
            if (q + 1u == j) {                                    // a new edge (walking backwards): ends at node j, starts at node s
                const uint32_t len = DPL_LO_LEN(nextLo);
                const uint32_t dist = DPL_LO_DIST(nextLo);
                const bool joins = dist != 0u && dist == eDist && s == j;      // same distance as the piece behind it: one run
                if (!joins) runEnd = j;
                eDist = dist; eCls = DPL_LO_CLS(nextLo); s = j - len;
            }
            uint32_t out = 0u;
            if (q == s) {                                         // start of the edge: read the back pointer of node s now (slot s - 1)
                nextLo = s ? BP[s - 1u] : 0u;
                const bool runGoesOn = s != 0u && eDist != 0u && DPL_LO_DIST(nextLo) == eDist;
                const uint32_t span = runEnd - q;                 // bytes from here to the end of the run
                if (eDist != 0u) {
                    if (!runGoesOn) {                             // first record of the run: what is left after the 64-byte records behind it
                        out = (eDist << 8) | (span - ((span - 1u) & ~63u));
                        if (phaseA) {
                            atomicAdd(&sCnt[lb][((eCls >= DPL_REP0 && eCls <= DPL_REP0 + 3u) ? GC_DPS_REPLEN : GC_DPS_LEN) + (span < 64u ? span : 64u)], 1u);
                            if (eCls == DPL_NEW) atomicAdd(&sCnt[lb][GC_DPS_SLOT + gc_dist_slot(eDist - 1u)], 1u);
                            else if (eCls == DPL_SREP) atomicAdd(&sCnt[lb][GC_DPS_NSREP], 1u);
                            else if (eCls == DPL_REP0) atomicAdd(&sCnt[lb][GC_DPS_NREP], 1u);
                            else if (eCls >= DPL_REP0 + 1u && eCls <= DPL_REP0 + 3u) atomicAdd(&sCnt[lb][GC_DPS_NREP1 + (eCls - DPL_REP0 - 1u)], 1u);
                            atomicAdd(&sCnt[lb][GC_DPS_NMAT], 1u);
                        }
                    } else if ((span & 63u) == 0u) out = (eDist << 8) | 64u;
                } else nLitC++;
                j = s;
            } else if (eDist != 0u && ((runEnd - q) & 63u) == 0u) out = (eDist << 8) | 64u;
            if (out != 0u) nRecs++;
            if (!phaseA) BP[q] = out;
        }
    }
#if defined(DPL_PROF) && !defined(HIPEMU)
    if (lane == 0u && !SAMPLE) atomicAdd(&g_dplProf[8], clock64() - ptick);
#endif
    if (phaseA) {
        if (n != 0u) atomicAdd(&sCnt[lb][GC_DPS_NLIT], nLitC);
        gc_wave_sync();
        for (uint32_t q = 0; q < BPW; q++) {
            const uint32_t bb = q < BPWr ? item * BPWr + q : nBlocks;
            if (bb >= nBlocks) continue;
            uint32_t* C = dpStat + (uint64_t)bb * GC_DPS_WORDS;
            for (uint32_t i = lane; i < GC_DPS_WORDS; i += 64u) { const uint32_t v = sCnt[q][i]; if (v) atomicAdd(&C[i], v); }
        }
        return;
    }
    // The sequence arrays behind W6 hold GC_MAX_SEQ_PER_BLOCK = 128 KiB / 5 entries per block: a window whose path has more records than its
    // share falls back to the finder's own records (followed from the window start they are matches of >= GC_MIN_MATCH bytes but the last one)
    const bool fallback = nRecs > n / GC_MIN_MATCH;
    if (__any(fallback)) {
        gc_wave_sync_global();
        if (fallback) for (uint32_t q = 0; q < n; q++) {          // (W6 follows them from the window start)
            const uint32_t r = R[q];
            uint32_t L = r & 0xFFu; if (L > n - q) L = n - q;
            BP[q] = (L >= MINLEN && (r & 0xFFu) >= GC_MIN_MATCH) ? ((r & ~0xFFu) | L) : 0u;
        }
    }
}

// one kernel per codec family and grid shape (the shared arrays of dpl_run are per instantiation)
#define DPL_KERNEL(name, REPS, MINLEN, SAMPLE) \
extern "C" __global__ void __launch_bounds__(GC_DPL_THREADS) \
name(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t nBlocks, uint32_t per, uint32_t frameBlocks, uint32_t phase, uint32_t* __restrict__ dpStat, uint32_t litCtxMask, \
     const uint32_t* __restrict__ rec, const uint16_t* __restrict__ rec3, const uint16_t* __restrict__ priceTab, uint32_t* __restrict__ recOut, uint32_t* __restrict__ winCost, \
     const uint8_t* __restrict__ litPrice) \
{ dpl_run<REPS, MINLEN, SAMPLE>(src, srcSize, nBlocks, per, frameBlocks, phase, dpStat, litCtxMask, rec, rec3, priceTab, recOut, winCost, litPrice); }

DPL_KERNEL(gc_mf_dpl2_kernel,  true, 2u, false)   // LZMA: every window
DPL_KERNEL(gc_mf_dpl2s_kernel, true, 2u, true)    // LZMA: the sample of phase A
DPL_KERNEL(gc_mf_dplz_kernel,  true, 3u, false)   // zstd: its three repeat offsets are the first three of the four kept here (ZSTD_updateRep, zstd_compress_internal.h:818: the same move-to-front); no match below three bytes, no short repeat
DPL_KERNEL(gc_mf_dplzs_kernel, true, 3u, true)

// The literal price of every position under its block's table (W6's statistics: literal given the top bits of the byte in front of it), one byte each
// (prices stop at GC_PRICE_MAX = 240): position-parallel, so the lanes of W7L read four prices per group of nodes instead of holding 4 KiB of table per
// block in LDS.  One workgroup per block.
extern "C" __global__ void __launch_bounds__(256)
gc_mf_litprice_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t nBlocks, uint32_t per, const uint16_t* __restrict__ priceTab, uint32_t litCtxArg, uint8_t* __restrict__ out)
{
    __shared__ uint16_t sLit[GC_PRICE_LEN];
    const uint32_t t = threadIdx.x, b = dpl_item(blockIdx.x, per);
    if (b >= nBlocks) return;
    const uint32_t litCtxMask = litCtxArg & 0xFFu, hasPrev = litCtxArg >> 31;
    { const GcU4* T4 = (const GcU4*)(priceTab + (uint64_t)b * GC_PRICE_WORDS); GcU4* S4 = (GcU4*)sLit; for (uint32_t i = t; i < GC_PRICE_LEN / 8u; i += 256u) S4[i] = T4[i]; }
    __syncthreads();
    const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t n = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
    for (uint32_t p = t * 16u; p < n; p += 256u * 16u) {
        uint8_t in[17], o[16];
        in[0] = (base + p + hasPrev) ? src[base + p - 1u] : (uint8_t)0;
        if (p + 16u <= n) __builtin_memcpy(in + 1, src + base + p, 16); else for (uint32_t k = 0; k < 16u; k++) in[1 + k] = p + k < n ? src[base + p + k] : (uint8_t)0;
#pragma unroll
        for (uint32_t k = 0; k < 16u; k++) o[k] = (uint8_t)sLit[((((uint32_t)in[k] >> 5) & litCtxMask) << 8) + in[k + 1u]];
        if (p + 16u <= n) __builtin_memcpy(out + base + p, o, 16); else for (uint32_t k = 0; k < 16u && p + k < n; k++) out[base + p + k] = o[k];
    }
}
