// gc_lz_price.hip -- price-based ("optimal") parse on top of the windowed match finder: W5s short candidates + W7 shortest path.
//
// Replaces the optimal parsers of the reference for the levels that use them: LZMA_optimalParse / LZMA_encodeOptimumSequence of
// Fast-LZMA2 (C/fast-lzma2/lzma2_enc.c:949, :1443; level 5 and up are FL2_opt / FL2_ultra, fl2_compress.c:37-104) and, with
// other price tables, ZSTD_compressBlock_opt_generic (C/zstd/zstd_opt.c:1077).  What those do: a forward dynamic programme
// over positions -- the cheapest way to reach every position, where a step is a literal or any prefix (length >= 2) of a match
// candidate, priced with the entropy coder's current statistics -- in buffers of 2-4 K positions, strictly serial.
//
// Here (measured first on the CPU with tools/lzma_parse_lab.c: what each simplification costs in compressed size):
//   - prices are STATIC per 128 KiB block: W6 parses the block greedily first and turns the symbol statistics of that parse
//     into price tables (literal given the top bits of the previous byte, piece length, distance slot, literal/match flag);
//     costs 0.1-0.8 % against prices taken from the adapting model, and makes every edge weight a pure function of the
//     position -- no coder state, no repeat-distance history travels along the path
//   - the unit is a WINDOW of 4 KiB (= one LZMA2 chunk / range-coder run, which no match crosses anyway): 32 independent
//     shortest-path problems per block, one WAVE each
//   - candidates per position: the finder's best match (any prefix of its <= 64 bytes) and a SHORT candidate (most recent
//     position within ~2-4 KiB with the same 3 bytes, lengths 2..17: W5s) -- what the reference gets from its 2-byte radix heads;
//     worth 1.4-3.7 % on binary data, nothing on text
//   - a match that fills the 64-byte cap is taken whole (the reference's fast-length rule) and the piece behind it with the
//     same distance is priced as its continuation
//
// W7 on the hardware: the open nodes of the programme are the next 64 positions -- ONE VGPR: while node i is expanded, lane t
// holds node i + t.  Lane t owns the edge of length t + 1 (so its length price is a loop invariant) and relaxes it into its own
// register after a one-lane DPP shift: no LDS, no atomics; per position a handful of VALU operations (a wave64 operation
// occupies its SIMD for four cycles: what the kernel costs is the number of vector instructions per node), the rest is scalar.
// A node is cost << 8 | kind << 6 | (length - 1), so the minimum carries its back pointer.  Back pointers of a window live in
// 4 KiB of LDS.
#include "gc_mf.h"
#include "gc_lz_parse.h"

__device__ __forceinline__ uint32_t ps_item(uint32_t bid, uint32_t per) { return (bid & (GC_XCDS - 1u)) * per + (bid >> 3); }

// ------------------------------------------------------------------------------------------------ W5s short candidates
// One wave per 2 KiB chunk, private tables in LDS (2^11 slots on a 3-byte hash, 2^10 slots on a 2-byte hash); the 2 KiB in front of
// the chunk (same frame) are inserted first.
// Per step 64 positions and one returning ds_max: the LDS unit serves the lanes in lane order, so a lane gets the most recent
// earlier position with its hash (same mechanism as W4).  Output: uint16 per position, (distance - 1) << 4 | (length - 2),
// GC_SHORT_NONE = no candidate.  Positions whose 17 bytes do not lie inside their block have none (keeps frames independent of
// what follows them).
#define SH_T        256u
#define SH_WAVES    (SH_T / 64u)
#define SH_CHUNK    2048u
#define SH_SLOT_LOG 11u
#define SH_SLOT2_LOG 10u              // second table, keyed by TWO bytes: the nearest earlier occurrence of a byte pair.  LZMA codes a 2-byte
#define SH_DIST2_MAX 127u             // match at a distance below 128 in ~13 bits (LZMA_optimalParse keeps them only there, lzma2_enc.c:1005-1010:
                                      // "len == 2 && dist >= 0x80" is dropped); where a literal costs 8-9 bits (binary data) that pays:
                                      // the reference's stream of lz-7zip holds 3.8 % of its symbols as such matches
#define SH_MAXLEN   17u
#define SH_STAGE_WORDS ((2u * SH_CHUNK + 32u) / 4u)

__device__ __forceinline__ uint32_t sh_lds_ld32(const uint32_t* sW, uint32_t i)     // bytes i .. i+3 of the staged window
{
    const uint32_t w = i >> 2, sh = (i & 3u) * 8u;
    return (uint32_t)((((uint64_t)sW[w + 1u] << 32) | sW[w]) >> sh);
}

extern "C" __global__ void __launch_bounds__(SH_T)
gc_mf_short_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t frameBlocks, uint32_t nChunks, uint32_t per, uint16_t* __restrict__ rec3)
{
    __shared__ uint32_t sTab[SH_WAVES][1u << SH_SLOT_LOG];
    __shared__ uint32_t sTab2[SH_WAVES][1u << SH_SLOT2_LOG];
    __shared__ uint32_t sW[SH_WAVES][SH_STAGE_WORDS];
    const uint32_t lane = threadIdx.x & 63u, wave = gc_uniform(threadIdx.x >> 6);    // (uniform: everything derived from it stays scalar)
    const uint32_t chunk = ps_item(blockIdx.x, per) * SH_WAVES + wave;
    if (chunk >= nChunks) return;                                 // no workgroup barrier below: waves are independent
    const uint64_t cs = (uint64_t)chunk * SH_CHUNK;
    const uint64_t frameBytes = (uint64_t)frameBlocks * GC_ZSTD_BLOCK_MAX;
    const uint64_t frameStart = cs / frameBytes * frameBytes;
    const uint32_t warm = cs > frameStart ? SH_CHUNK : 0u;
    const uint64_t ws = cs - warm;                                // window start (absolute)
    const uint64_t blockEnd = ((cs / GC_ZSTD_BLOCK_MAX + 1u) * GC_ZSTD_BLOCK_MAX) < srcSize ? (cs / GC_ZSTD_BLOCK_MAX + 1u) * GC_ZSTD_BLOCK_MAX : srcSize;
    uint32_t* tab = sTab[wave];
    uint32_t* tab2 = sTab2[wave];
    uint32_t* W = sW[wave];
    for (uint32_t i = lane; i < (1u << SH_SLOT_LOG); i += 64u) tab[i] = 0;
    for (uint32_t i = lane; i < (1u << SH_SLOT2_LOG); i += 64u) tab2[i] = 0;
    for (uint32_t c = lane; c < SH_STAGE_WORDS / 4u; c += 64u) {  // 16 bytes per lane; zero past the end of the input
        GcU4 v; v.x = v.y = v.z = v.w = 0;
        const uint64_t pos = ws + 16ull * c;
        if (pos + 16u <= srcSize) __builtin_memcpy(&v, src + pos, 16);
        else if (pos < srcSize) { uint8_t tmp[16]; for (uint32_t k = 0; k < 16u; k++) tmp[k] = pos + k < srcSize ? src[pos + k] : (uint8_t)0; __builtin_memcpy(&v, tmp, 16); }
        W[4u * c] = v.x; W[4u * c + 1u] = v.y; W[4u * c + 2u] = v.z; W[4u * c + 3u] = v.w;
    }
    gc_wave_sync();
    const uint32_t nPos = warm + SH_CHUNK;
    for (uint32_t q0 = 0; q0 < nPos; q0 += 64u) {
        const uint32_t q = q0 + lane;                             // window-relative
        const uint64_t P = ws + q;
        const bool listed = P + SH_MAXLEN <= blockEnd;
        const uint32_t x = sh_lds_ld32(W, q) & 0xFFFFFFu;
        const uint32_t h = x * 0x9E3779B1u;
        const uint32_t mine = ((q + 1u) << 8) | ((h >> 13) & 0xFFu);
        const uint32_t h2 = (x & 0xFFFFu) * 0x9E3779B1u;
        const uint32_t mine2 = ((q + 1u) << 8) | ((h2 >> 14) & 0xFFu);
        uint32_t seen = 0, seen2 = 0;
        if (listed) { seen = atomicMax(&tab[h >> (32u - SH_SLOT_LOG)], mine); seen2 = atomicMax(&tab2[h2 >> (32u - SH_SLOT2_LOG)], mine2); }
        gc_wave_step();
        if (q0 < warm) continue;                                  // uniform: the warm-up only inserts
        uint32_t out = GC_SHORT_NONE, bestLen = 0;
        if (listed && seen != 0u && seen < mine && ((seen ^ mine) & 0xFFu) == 0u) {
            const uint32_t c = (seen >> 8) - 1u;                  // candidate, window-relative, c < q
            uint32_t len = 0;
#pragma unroll
            for (uint32_t k = 0; k < 20u; k += 4u) {
                if (len == k) { const uint32_t d = sh_lds_ld32(W, q + k) ^ sh_lds_ld32(W, c + k); len += d ? (uint32_t)(__ffs((int)d) - 1) >> 3 : 4u; }
            }
            if (len > SH_MAXLEN) len = SH_MAXLEN;
            if (len >= 2u && q - c <= 4095u) { out = ((q - c - 1u) << 4) | (len - 2u); bestLen = len; }
        }
        if (listed && bestLen < 3u && seen2 != 0u && seen2 < mine2 && ((seen2 ^ mine2) & 0xFFu) == 0u) {
            // the byte pair's nearest earlier occurrence: only wanted where the 3-byte table has nothing of >= 3 bytes (a position
            // that shares three bytes shares two: if the pair's nearest occurrence were >= 3 long it would be the one found above)
            const uint32_t c = (seen2 >> 8) - 1u;
            const uint32_t d = sh_lds_ld32(W, q) ^ sh_lds_ld32(W, c);
            const uint32_t len = d ? (uint32_t)(__ffs((int)d) - 1) >> 3 : 4u;
            if (len >= 2u && q - c <= SH_DIST2_MAX && (bestLen < 2u || q - c - 1u < (out >> 4))) out = ((q - c - 1u) << 4) | ((len > 3u ? 3u : len) - 2u);
        }
        if (P < srcSize) rec3[P] = (uint16_t)out;
    }
}

// ------------------------------------------------------------------------------------------------ W7 shortest path
// Repeat hints of the positions p = p0 + lane of one group of 64 (LZMA): the distances of the finder records that end one to four bytes in
// front of p -- "match, a few literals (or short repeats), the same distance again" is how a copy with changed bytes looks, and where the path
// took that match its distance is the node's repeat distance.  Two distinct distances are kept, the nearest record end first.
// recHere = records of p0 + lane, recPrev = records of p0 - 64 + lane.  Only records of 2..8 bytes are looked at: the tail of a longer
// match is recorded at its later positions with the same distance.
__device__ __forceinline__ void dp_rep_hints(uint32_t recHere, uint32_t recPrev, uint32_t lane, uint32_t& d1, uint32_t& d2)
{
    uint32_t e1 = 0, e2 = 0, e3 = 0, e4 = 0;                      // e<k>: distance of the longest such record that ends at p - k
#pragma unroll
    for (uint32_t j = 3u; j <= 12u; j++) {                        // the record j positions in front of p
        const uint32_t a = __shfl(recHere, (int)((lane - j) & 63u)), b = __shfl(recPrev, (int)((lane - j) & 63u));
        const uint32_t r = lane >= j ? a : b;
        const uint32_t len = r & 0xFFu, dist = r >> 8;            // (longer records are visited later and win)
        if (j <= 9u && len == j - 1u) e1 = dist;
        if (j >= 4u && j <= 10u && len == j - 2u) e2 = dist;
        if (j >= 5u && j <= 11u && len == j - 3u) e3 = dist;
        if (j >= 6u && len == j - 4u) e4 = dist;
    }
    d1 = e1 ? e1 : (e2 ? e2 : (e3 ? e3 : e4));
    d2 = (e4 != 0u && e4 != d1) ? e4 : 0u;                        // nearest record end with another distance
    d2 = (e3 != 0u && e3 != d1) ? e3 : d2;
    d2 = (e2 != 0u && e2 != d1) ? e2 : d2;
    d2 = (e1 != 0u && e1 != d1) ? e1 : d2;
}

#define DP_T       256u
#define DP_WAVES   (DP_T / 64u)
#define DP_WIN_LOG 12u
#define DP_WIN     (1u << DP_WIN_LOG)
#define DP_WINS_PER_BLOCK (GC_ZSTD_BLOCK_MAX >> DP_WIN_LOG)
#define DP_ROWS    (DP_WIN / 64u + 1u)
#define DP_INF     0xFFFFFFFFu
#define DP_CONT_PRICE 4u               // continuation of a capped match: a quarter of a bit
#define DP_MAX_MATCHES (DP_WIN / GC_MIN_MATCH)                    // matches per window that the sequence arrays are sure to hold

// node word: cost << 8 | kind << 6 | (length - 1);  kind 0 literal (length 1), 1 finder candidate, 2 short candidate, 3 repeat.
// The low byte is the back pointer; a price in word units is price << 8.
#define DP_KIND1   (1u << 6)
#define DP_KIND2   (2u << 6)
#define DP_KIND3   (3u << 6)           // LZMA only: a repeat of the path's current distance (length 1 = LZMA's "short rep")

// Two phases (one launch each).  The price tables W6 leaves behind are those of a GREEDY parse whose matches have >= 5 bytes: it says
// nothing about the 2-4 byte matches, and where it finds no matches at all (16-bit samples, tables of small records) it makes every
// match look expensive, so a shortest path under those prices never tries what the reference codes such data with (its adaptive
// model prices a symbol by how often the parse itself has used it: the reference's stream of the PCM-like part of the Silesia
// stand-in is 19 % 3-byte matches, the greedy-priced path found 0.1 %).  So:
//   phase A  the shortest path of a SAMPLE of windows (4 of the 32 windows of a block), under W6's prices capped at optimistic
//            ceilings for the match side, and the symbol counts of those paths (lengths, distance slots, literals / matches)
//   phase B  every window, lengths / slots / flags priced from phase A's counts (literals keep W6's prices: which byte values
//            occur does not depend on the parse)
// i.e. one round of the iteration "parse -> statistics -> prices -> parse" (tools/lzma_parse_lab.c LAB_ITER: -0.6 % on text).
template <uint32_t minLen /* shortest match: 2 LZMA, 3 zstd */>
__device__ __forceinline__ void dp_window(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t nBlocks, uint32_t per, uint32_t frameBlocks, uint32_t phaseArg /* 0: phase A; 1: phase B; 2: one
                phase only (W6's prices as they are, every window, no counting) */, uint32_t* __restrict__ dpStat,
                uint32_t litCtxArg /* bits of (previous byte >> 5) that select the literal price row: 7 LZMA (lc = 3), 0 zstd; bit 31: the
                                      byte in front of src exists (src is a later part of one buffer) */,
                const uint32_t* __restrict__ rec, const uint16_t* __restrict__ rec3, const uint16_t* __restrict__ priceTab, uint32_t* __restrict__ recOut,
                uint32_t* __restrict__ winCost /* per window: cost of the cheapest path in 1/16 bit (an estimate of its coded size), or nullptr */)
{
    const uint32_t litCtxMask = litCtxArg & 0xFFu, hasPrev = litCtxArg >> 31;
    __shared__ uint16_t sPrice[GC_PRICE_WORDS];
    __shared__ uint8_t sRow[DP_WAVES][DP_ROWS][64];               // back pointers by end node, later edges by start node
    __shared__ uint32_t sCnt[GC_DPS_WORDS];                       // phase A: symbol counts of this workgroup's paths
    constexpr bool REPS = minLen == 2u;                           // LZMA: the path carries its last match distance (rep0) and may repeat it
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = gc_uniform(t >> 6);        // (uniform: the node index i must live in an SGPR)
    const bool selective = (phaseArg & GC_DP_SELECT) != 0u;       // phase B: blocks whose sampled paths repeat distances belong to W7L (gc_lz_dpl.hip)
    phaseArg &= 15u;
    const bool phaseA = phaseArg == 0u, phaseB = phaseArg == 1u;
    bool useReps = REPS;                                          // (uniform per workgroup = per block)
    // a workgroup = DP_WAVES windows of one block: consecutive ones, or in phase A every eighth one (windows 3, 11, 19, 27)
    const uint32_t item = ps_item(blockIdx.x, per);
    const uint32_t wgPerBlock = phaseA ? DP_WINS_PER_BLOCK / DP_WAVES / 8u : DP_WINS_PER_BLOCK / DP_WAVES;
    const uint32_t b = item / wgPerBlock;
    if (b >= nBlocks) return;
    if (selective && phaseB && GC_DPS_RICH(dpStat + (uint64_t)b * GC_DPS_WORDS)) return;      // (uniform per workgroup)
    const uint32_t win = phaseA ? ((item % wgPerBlock) * DP_WAVES + wave) * 8u + 3u : (item % wgPerBlock) * DP_WAVES + wave;
    { const GcU4* T4 = (const GcU4*)(priceTab + (uint64_t)b * GC_PRICE_WORDS); GcU4* S4 = (GcU4*)sPrice;
      for (uint32_t i = t; i < GC_PRICE_WORDS / 8u; i += DP_T) S4[i] = T4[i]; }
    if (phaseA) for (uint32_t i = t; i < GC_DPS_WORDS; i += DP_T) sCnt[i] = 0;
    __syncthreads();
    if (REPS && t == 0u) { sPrice[GC_PRICE_FLAGS + 2u] = 48u; sPrice[GC_PRICE_FLAGS + 3u] = 56u; }     // repeats before anything is known: 3 / 3.5 bits on top of the match flag
    __syncthreads();
    if (phaseA && sPrice[GC_PRICE_FLAGS + 1u] >= 64u) {
        // A block whose greedy parse is (almost) all literals -- fewer than one symbol in 16 is a match -- says nothing about what matches would
        // cost if they were used: price the match side at optimistic ceilings (1/16 bit: flag 1, length <= 9: 0.5, slot 3 bits: what they cost once a third of the symbols are such matches), so that the
        // sampled paths take the short matches that exist, and let phase B price them by how often they were taken.  (Blocks with matches keep
        // W6's prices in phase A: ceilings there made text 0.3 % larger.)
        __syncthreads();
        for (uint32_t i = t; i < GC_PRICE_NLEN + 64u + 1u; i += DP_T) {
            const uint32_t idx = i < GC_PRICE_NLEN ? GC_PRICE_LEN + i : (i < GC_PRICE_NLEN + 64u ? GC_PRICE_SLOT + (i - GC_PRICE_NLEN) : GC_PRICE_FLAGS + 1u);
            const uint32_t cap = i < 10u ? 8u : (i < GC_PRICE_NLEN ? 0xFFFFu : (i < GC_PRICE_NLEN + 64u ? 48u : 16u));
            if (sPrice[idx] > cap) sPrice[idx] = (uint16_t)cap;
        }
        __syncthreads();
    }
    if (phaseB) {                                                 // lengths, slots and flags from the counts of phase A's paths
        const uint32_t* C = dpStat + (uint64_t)b * GC_DPS_WORDS;
        const uint32_t nLit = C[GC_DPS_NLIT], nMat = C[GC_DPS_NMAT];
        if (REPS && nLit + nMat != 0u) useReps = (C[GC_DPS_NREP] + C[GC_DPS_NSREP]) * 64u >= nMat;     // fewer than 1.6 % repeats among the matches: not worth the longer loop
        if (nLit + nMat != 0u) {                                  // (uniform; a block whose sampled windows do not exist keeps W6's prices)
            for (uint32_t i = t; i < GC_PRICE_NLEN + 64u + 2u; i += DP_T) {
                if (i < GC_PRICE_NLEN) sPrice[GC_PRICE_LEN + i] = (uint16_t)pz_price(8u * C[GC_DPS_LEN + i] + 1u, 8u * nMat + 63u);
                else if (i < GC_PRICE_NLEN + 64u) sPrice[GC_PRICE_SLOT + (i - GC_PRICE_NLEN)] = (uint16_t)pz_price(8u * C[GC_DPS_SLOT + (i - GC_PRICE_NLEN)] + 1u, 8u * nMat + 44u);
                else if (i == GC_PRICE_NLEN + 64u) sPrice[GC_PRICE_FLAGS] = (uint16_t)pz_price(nLit + 1u, nLit + nMat + 2u);
                else sPrice[GC_PRICE_FLAGS + 1u] = (uint16_t)pz_price(nMat + 1u, nLit + nMat + 2u);
            }
            if (REPS && t == 0u) {                                // "is a repeat" + "repeat 0" + long / short, from how often phase A's paths repeated
                const uint32_t nRep = C[GC_DPS_NREP], nSrep = C[GC_DPS_NSREP], isRep = pz_price(nRep + nSrep + 1u, nMat + 2u);
                sPrice[GC_PRICE_FLAGS + 2u] = (uint16_t)(isRep + 4u + pz_price(nRep + 1u, nRep + nSrep + 2u));
                sPrice[GC_PRICE_FLAGS + 3u] = (uint16_t)(isRep + 4u + pz_price(nSrep + 1u, nRep + nSrep + 2u));
            }
        }
        __syncthreads();
    }
    const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
    const uint32_t w0 = win << DP_WIN_LOG;
    const bool live = w0 < blockLen;                              // (a wave without a window still meets the others at the barrier below)
    if (live) {
    const uint32_t n = (blockLen - w0) < DP_WIN ? (blockLen - w0) : DP_WIN;          // nodes 0 .. n
    const uint32_t* R = rec + base + w0;
    const uint16_t* R3 = rec3 + base + w0;
    const uint8_t* S = src + base + w0;
    uint8_t (*row)[64] = sRow[wave];
    const uint32_t flagLit = sPrice[GC_PRICE_FLAGS], flagMat = sPrice[GC_PRICE_FLAGS + 1u];

    // Lane t holds node i + t while node i is expanded (lane 0 = node i, final), and owns the edge of length t + 1: its length
    // price is a loop invariant.  After the expansion the register is shifted by one lane (DPP), node i + 64 enters at lane 63.
    // What the loop costs is its instruction count -- a wave issues at most one instruction per four cycles, vector or scalar --
    // so everything that depends on the position only is prepared 64 positions at a time (lane = position) as ready-made word
    // addends, and a step is: read node i, two scalar adds, shift, two vector adds, two compares + selects, two minima.
    const uint32_t X1 = ((uint32_t)sPrice[GC_PRICE_LEN + lane + 1u] << 8) | DP_KIND1 | lane;     // length price + back pointer of this lane's edge
    const uint32_t X2 = (X1 & ~0xC0u) | DP_KIND2;
    const uint32_t X3 = X1 | DP_KIND3;
    const uint32_t capAdd = ((uint32_t)sPrice[GC_PRICE_LEN + GC_MATCH_CAP] << 8) | DP_KIND1 | (GC_MATCH_CAP - 1u);
    // LZMA: a repeat of the path's last match distance costs the match flag + a few bits instead of slot + footer.  Every node carries
    // that distance (ringR, shifted and selected along with ring); a candidate whose distance equals it is priced as a repeat, and where
    // a hint (dp_rep_hints: the distance of a record that ends just in front of the node) names it, the bytes that repeat at the node
    // are edges of their own (kind 3): lengths 2.., and length 1 = LZMA's short repeat -- what the reference codes copies with changed
    // bytes, 16-bit samples and small records with (LZMA_optimalParse, lzma2_enc.c:949-1440: rep0 / shortrep edges at every position).
    // (Looking at EVERY node for repeats at near distances, from the window's bytes, was tried as well: +0.05 % for a third more time.)
    // Blocks whose sampled paths (phase A) hardly ever repeated run phase B without any of this (useReps).
    const uint32_t repAdd = ((uint32_t)flagMat + sPrice[GC_PRICE_FLAGS + 2u]) << 8, srepAdd = (((uint32_t)flagMat + sPrice[GC_PRICE_FLAGS + 3u]) << 8) | DP_KIND3;
    uint32_t ring = lane == 0u ? 0u : DP_INF;                     // node 0: cost 0
    uint32_t ringR = 0;                                           // node 0: no distance known
    uint32_t cc = 0;                                              // back pointers of the current group's nodes (lane = node mod 64)
    uint32_t curG = 0xFFFFFFFFu;
    uint32_t contBit = 0, contOff = 0;                            // contBit = 64: the node being expanded is the end of a capped match with this distance
    GcPub pL, pA, pL3, pA3, pC, pD, pD3;  // per position of the group: candidate lengths, word addends (distance price), literal addend, distances
    // raw inputs of the NEXT group, requested one group (64 nodes) ahead of their use
    uint32_t nR = 0, nR3 = GC_SHORT_NONE, nPB = 0;                // nPB: the position's byte << 8 | the byte in front of it -- ONE load, taken apart when it is used
    if (lane < n) { nR = R[lane]; nR3 = R3[lane]; nPB = ((uint32_t)S[lane] << 8) | ((base + w0 + lane + hasPrev) ? (uint32_t)S[(int64_t)lane - 1] : 0u); }
    // LZMA: the records are requested TWO groups ahead (nR = next group, nR2 = the one behind it), because the repeat hints of a group need its
    // records one group early: their distances tell which 16 bytes to fetch for the comparison (hB1 / hB2, one group ahead of their use)
    GcPub pH1, pH2;                                               // per position: hint distance << 5 | bytes that repeat there (1..16), 0 = none
    uint32_t nR2 = 0, hD1 = 0, hD2 = 0; LzW16 hB0, hB1, hB2; hB0.a = hB0.b = hB1.a = hB1.b = hB2.a = hB2.b = 0;     // hB0: the position's own 16 bytes
    if (useReps) {
        if (64u + lane < n) nR2 = R[64u + lane];
        dp_rep_hints(nR, 0u, lane, hD1, hD2);                     // group 0: nothing in front of the window
        const uint64_t absP = base + w0 + lane;
        if (lane >= n || absP + 16u > srcSize) { hD1 = 0; hD2 = 0; }
        if (hD1 | hD2) hB0 = lz_ld16(S, (uint64_t)lane);
        if (hD1) hB1 = lz_ld16(S + (int64_t)lane - (int64_t)hD1, 0);
        if (hD2) hB2 = lz_ld16(S + (int64_t)lane - (int64_t)hD2, 0);
    }
    uint32_t i = 0;
    // outer loop: one round per group of 64 positions (its inputs were requested a group earlier and are consumed before the new requests go out:
    // the loads have a whole group of nodes to arrive); inner loop: the nodes of the group.  (A forced jump of 64 lands in the next group.)
    while (i < n) {
        const uint32_t g = i >> 6;
        {                                                         // new group of 64 positions
            if (curG != 0xFFFFFFFFu) row[curG][lane] = (uint8_t)cc;
            cc = 0; curG = g;
            const uint32_t p = (g << 6) + lane;                   // window-relative
            uint32_t L = 0, A = 0, L3 = 0, A3 = 0, C = 0, D = 0, D3 = 0;
            if (p < n) {
                const uint32_t r = nR, r3 = nR3;
                uint32_t l = r & 0xFFu, off = r >> 8;
                if (l > n - p) l = n - p;
                if (l >= minLen) { const uint32_t sl = gc_dist_slot(off - 1u); L = l; A = (flagMat + sPrice[GC_PRICE_SLOT + sl] + (sl >= 4u ? 16u * ((sl >> 1) - 1u) : 0u)) << 8; D = off; }
                if (r3 != GC_SHORT_NONE) {
                    uint32_t l3 = (r3 & 15u) + 2u; const uint32_t off3 = (r3 >> 4) + 1u;
                    if (l3 > n - p) l3 = n - p;
                    if (l3 >= minLen && !(l >= l3 && off <= off3)) { const uint32_t sl = gc_dist_slot(off3 - 1u); L3 = l3; A3 = (flagMat + sPrice[GC_PRICE_SLOT + sl] + (sl >= 4u ? 16u * ((sl >> 1) - 1u) : 0u)) << 8; D3 = off3; }
                }
                C = (flagLit + sPrice[GC_PRICE_LIT + ((((nPB & 0xFFu) >> 5) & litCtxMask) << 8) + (nPB >> 8)]) << 8;
            }
            gc_publish(pL, L); gc_publish(pA, A); gc_publish(pL3, L3); gc_publish(pA3, A3); gc_publish(pC, C); gc_publish(pD, D);
            const uint32_t pn = p + 64u;
            if (useReps) {
                gc_publish(pD3, D3);
                uint32_t H1 = 0, H2 = 0;                          // hints of this group: the fetched bytes against the window's own (LDS)
                if (hD1 | hD2) {
                    const LzW16 me = hB0;
                    if (hD1) { uint32_t l = lz_cmp16(me, hB1); if (l > n - p) l = n - p; if (l) H1 = (hD1 << 5) | l; }
                    if (hD2) { uint32_t l = lz_cmp16(me, hB2); if (l > n - p) l = n - p; if (l) H2 = (hD2 << 5) | l; }
                }
                gc_publish(pH1, H1); gc_publish(pH2, H2);
                const uint32_t recThis = nR;                      // (nR still holds this group's records here)
                dp_rep_hints(nR2, recThis, lane, hD1, hD2);       // next group's hints: distances now, bytes in flight until its turn
                if (pn >= n || base + w0 + pn + 16u > srcSize) { hD1 = 0; hD2 = 0; }
                if (hD1 | hD2) hB0 = lz_ld16(S, (uint64_t)pn);
                if (hD1) hB1 = lz_ld16(S + (int64_t)pn - (int64_t)hD1, 0);
                if (hD2) hB2 = lz_ld16(S + (int64_t)pn - (int64_t)hD2, 0);
                nR = nR2; nR2 = pn + 64u < n ? R[pn + 64u] : 0u;
                if (pn < n) { nR3 = R3[pn]; uint16_t two; __builtin_memcpy(&two, S + pn - 1u, 2); nPB = two; }
            } else if (pn < n) { nR = R[pn]; nR3 = R3[pn]; uint16_t two; __builtin_memcpy(&two, S + pn - 1u, 2); nPB = two; }
        }
        while (i < n && (i >> 6) == g) {
        const uint32_t k = i & 63u;
        const uint32_t w = gc_readlane(ring, 0u);                 // node i: final
        const uint32_t Rn = useReps ? gc_readlane(ringR, 0u) : 0u;   // ... and the distance of the last match on the way to it (0: none yet)
        cc = gc_writelane(cc, w, k);                              // (its low byte is the back pointer)
        const uint32_t costw = w & ~0xFFu;                        // cost in word units
        const uint32_t L = gc_peek(pL, k);
        if ((L | contBit) >= GC_MATCH_CAP) {                      // uniform, rare: end and / or start of a capped match
            const bool cont = contBit != 0u && L != 0u && gc_peek(pD, k) == contOff;
            contBit = 0;
            if (L == GC_MATCH_CAP) {                              // take the capped match whole (L is clipped to the window)
                const uint32_t c = costw + (cont ? (DP_CONT_PRICE << 8) | DP_KIND1 | (GC_MATCH_CAP - 1u) : gc_peek(pA, k) + capAdd);
                ring = lane == 0u ? c : DP_INF;                   // node i + 64 is the only open node
                contBit = GC_MATCH_CAP; contOff = gc_peek(pD, k);
                if (useReps) ringR = contOff;
                i += GC_MATCH_CAP;
                continue;
            }
            if (cont) {                                           // the piece behind a capped match: same price for every length
                ring = gc_wave_shl1(ring, DP_INF);
                uint32_t cand = lane < L ? costw + ((DP_CONT_PRICE << 8) | DP_KIND1 | lane) : DP_INF;
                cand = gc_writelane_c<0>(cand, costw + gc_peek(pC, k));
                if (minLen > 2u) cand = gc_writelane_c<1>(cand, DP_INF);
                if (useReps) { ringR = gc_wave_shl1(ringR, 0u); const uint32_t candR = gc_writelane_c<0>(contOff, Rn); ringR = cand < ring ? candR : ringR; }
                ring = cand < ring ? cand : ring;
                i++;
                continue;
            }
        }
        ring = gc_wave_shl1(ring, DP_INF);                        // lane t: node i + 1 + t, the end of this lane's edge
        if (!useReps) {
            const uint32_t w1 = X1 + (costw + gc_peek(pA, k));
            uint32_t cand = lane < L ? w1 : DP_INF;
            const uint32_t L3 = gc_peek(pL3, k);
            if (L3 != 0u) {                                       // uniform
                const uint32_t w2 = X2 + (costw + gc_peek(pA3, k));
                cand = (lane < L3 && w2 < cand) ? w2 : cand;
            }
            cand = gc_writelane_c<0>(cand, costw + gc_peek(pC, k));   // lane 0: the literal (kind 0, length 1)
            if (minLen > 2u) cand = gc_writelane_c<1>(cand, DP_INF);  // zstd: no matches of two bytes
            ring = cand < ring ? cand : ring;
        } else {
            ringR = gc_wave_shl1(ringR, 0u);
            const uint32_t D = gc_peek(pD, k);
            const uint32_t w1 = X1 + (costw + ((L != 0u && D == Rn) ? repAdd : gc_peek(pA, k)));      // the candidate repeats the path's distance: repeat price
            uint32_t cand = lane < L ? w1 : DP_INF;
            uint32_t candR = D;
            const uint32_t L3 = gc_peek(pL3, k);
            if (L3 != 0u) {                                       // uniform
                const uint32_t D3 = gc_peek(pD3, k);
                const uint32_t w2 = X2 + (costw + (D3 == Rn ? repAdd : gc_peek(pA3, k)));
                const bool t2 = lane < L3 && w2 < cand;
                cand = t2 ? w2 : cand; candR = t2 ? D3 : candR;
            }
            uint32_t lane0 = costw + gc_peek(pC, k);              // the literal (kind 0, length 1)
            uint32_t Lr = 0;                                      // bytes from node i on that repeat at the path's distance: where a hint names this very distance
            if (Rn != 0u) {
                const uint32_t h1 = gc_peek(pH1, k), h2 = gc_peek(pH2, k);
                Lr = (h1 >> 5) == Rn ? (h1 & 31u) : ((h2 >> 5) == Rn ? (h2 & 31u) : 0u);
            }
            if (Lr != 0u) {                                       // uniform
                const uint32_t w3 = X3 + (costw + repAdd);
                const bool t3 = lane < Lr && lane != 0u && w3 < cand;
                cand = t3 ? w3 : cand; candR = t3 ? Rn : candR;
                const uint32_t sr = costw + srepAdd;              // short repeat: one byte, kind 3, length 1
                lane0 = sr < lane0 ? sr : lane0;
            }
            cand = gc_writelane_c<0>(cand, lane0);
            candR = gc_writelane_c<0>(candR, Rn);                 // a literal or a short repeat keeps the distance
            const bool better = cand < ring;
            ring = better ? cand : ring; ringR = better ? candR : ringR;
        }
        i++;
        }   // nodes of the group
    }
    // node n (i == n; a forced jump never passes n): lane 0
    {
        const uint32_t g = n >> 6, k = n & 63u;
        if (g != curG) { if (curG != 0xFFFFFFFFu) row[curG][lane] = (uint8_t)cc; cc = 0; curG = g; }
        const uint32_t wn = gc_readlane(ring, 0u);
        cc = gc_writelane(cc, wn, k);
        row[curG][lane] = (uint8_t)cc;
        if (winCost != nullptr && lane == 0u) winCost[(uint64_t)b * DP_WINS_PER_BLOCK + win] = wn >> 8;
    }
    gc_wave_sync();
    // ---- walk back from node n; the rows are rewritten in place: lane of the START position of every match on the path <- its byte
    uint32_t nMatch = 0;
    {
        uint32_t j = n, q = n >> 6;
        uint32_t cg = row[q][lane], fg = 0;
        while (j > 0u) {
            const uint32_t c = gc_readlane(cg, j & 63u);
            const uint32_t s = j - ((c & 63u) + 1u), qs = s >> 6;
            if (qs != q) { row[q][lane] = (uint8_t)fg; gc_wave_sync(); q = qs; cg = row[q][lane]; fg = 0; }
            if ((c >> 6) != 0u) { nMatch++; fg = gc_writelane(fg, c, s & 63u); }
            j = s;
        }
        row[q][lane] = (uint8_t)fg;
        gc_wave_sync();
    }
    // ---- the window's parse as records: a match where the path takes one, 0 (literal) elsewhere
    //      The sequence arrays behind W6 hold GC_MAX_SEQ_PER_BLOCK = 128 KiB / 5 entries per block.  A path of many 2-4 byte
    //      matches could exceed that, so a window whose path has more than its share (DP_MAX_MATCHES = 4096 / 5) falls back to the
    //      finder's own records, clipped to the window: followed from the window start they are matches of >= GC_MIN_MATCH bytes
    //      (but the last one), i.e. within the share.
    const bool fallback = nMatch > DP_MAX_MATCHES;                // uniform
    uint32_t* RO = recOut + base + w0;
    uint32_t lenSum = 0, nRepC = 0, nSrepC = 0;
    uint32_t lastD = 0;                                           // distance of the last kind 1 / 2 match of the path so far (uniform)
    for (uint32_t q = 0; (q << 6) < n; q++) {
        const uint32_t p = (q << 6) + lane;
        const uint32_t c = p < n ? row[q][lane] : 0u;
        uint32_t out = 0;
        if (fallback) {
            const uint32_t r = p < n ? R[p] : 0u;
            uint32_t L = r & 0xFFu; if (p < n && L > n - p) L = n - p;
            out = (L >= minLen && (r & 0xFFu) >= GC_MIN_MATCH) ? ((r & ~0xFFu) | L) : 0u;       // (records of the short pass are left out)
        } else {
            const uint32_t kind = c >> 6;
            uint32_t off = kind == 1u ? R[p] >> 8 : (kind == 2u ? ((uint32_t)R3[p] >> 4) + 1u : 0u);
            if (REPS) {                                           // a repeat takes the distance of the nearest path match in front of it that has one
                const uint32_t from = gc_wave_incl_max(off != 0u ? lane + 1u : 0u);
                const uint32_t offPrev = __shfl(off, (int)(from ? from - 1u : 0u));
                const uint32_t dHere = from ? offPrev : lastD;
                if (kind == 3u) { off = dHere; nRepC += (c & 63u) != 0u ? 1u : 0u; nSrepC += (c & 63u) == 0u ? 1u : 0u; }
                lastD = gc_readlane(dHere, 63u);
            }
            if (c != 0u && off != 0u) out = (off << 8) | ((c & 63u) + 1u);
        }
        if (p >= n) continue;
        if (!phaseA) RO[p] = out;
        else if (out != 0u && !fallback) {                        // phase A: count the path's symbols instead (nothing is written)
            atomicAdd(&sCnt[GC_DPS_LEN + (out & 0xFFu)], 1u);
            if ((c >> 6) != 3u) atomicAdd(&sCnt[GC_DPS_SLOT + gc_dist_slot((out >> 8) - 1u)], 1u);
            lenSum += out & 0xFFu;
        }
    }
    if (phaseA && !fallback) {
        lenSum = gc_wave_sum(lenSum);
        if (REPS) { nRepC = gc_wave_sum(nRepC); nSrepC = gc_wave_sum(nSrepC); }
        if (lane == 0u) { atomicAdd(&sCnt[GC_DPS_NMAT], nMatch); atomicAdd(&sCnt[GC_DPS_NLIT], n - lenSum); atomicAdd(&sCnt[GC_DPS_NREP], nRepC); atomicAdd(&sCnt[GC_DPS_NSREP], nSrepC); }
    }
    }   // live
    if (phaseA) {
        __syncthreads();
        uint32_t* C = dpStat + (uint64_t)b * GC_DPS_WORDS;
        for (uint32_t i = t; i < GC_DPS_WORDS; i += DP_T) { const uint32_t v = sCnt[i]; if (v) atomicAdd(&C[i], v); }
    }
}

// one kernel per shortest match length (the shared arrays of dp_window are per instantiation: two in one kernel would double its LDS)
extern "C" __global__ void __launch_bounds__(DP_T)
gc_mf_dp2_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t nBlocks, uint32_t per, uint32_t frameBlocks, uint32_t phase, uint32_t* __restrict__ dpStat, uint32_t litCtxMask,
                 const uint32_t* __restrict__ rec, const uint16_t* __restrict__ rec3, const uint16_t* __restrict__ priceTab, uint32_t* __restrict__ recOut,
                 uint32_t* __restrict__ winCost)
{
    dp_window<2u>(src, srcSize, nBlocks, per, frameBlocks, phase, dpStat, litCtxMask, rec, rec3, priceTab, recOut, winCost);
}
extern "C" __global__ void __launch_bounds__(DP_T)
gc_mf_dp3_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint32_t nBlocks, uint32_t per, uint32_t frameBlocks, uint32_t phase, uint32_t* __restrict__ dpStat, uint32_t litCtxMask,
                 const uint32_t* __restrict__ rec, const uint16_t* __restrict__ rec3, const uint16_t* __restrict__ priceTab, uint32_t* __restrict__ recOut,
                 uint32_t* __restrict__ winCost)
{
    dp_window<3u>(src, srcSize, nBlocks, per, frameBlocks, phase, dpStat, litCtxMask, rec, rec3, priceTab, recOut, winCost);
}
