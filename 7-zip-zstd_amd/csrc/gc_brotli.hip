// gc_brotli.hip -- BROTLI path (7-Zip method id 0x4F71102): one brotli meta-block per 128 KiB match-finder block.
//
// Replaces, behind NCompress::NBROTLI::CEncoder (CPP/7zip/Compress/BrotliEncoder.cpp:118-164) and the brotli-mt chunk
// framing (C/zstdmt/brotli-mt_compress.c:209-333), the per-chunk BrotliEncoderCompress: command formation
// (C/brotli/enc/command.h: insert / copy length codes, CombineLengthCodes; enc/prefix.h: distance prefix codes), histogram
// building + BrotliCreateHuffmanTree / BrotliConvertBitDepthsToSymbols (C/brotli/br_entropy_encode.c:68,474), the prefix
// code serialisation BuildAndStoreHuffmanTree / BrotliStoreHuffmanTree (C/brotli/br_brotli_bit_stream.c:349,254-330) and
// the command / literal / distance bit stream of BrotliStoreMetaBlock (br_brotli_bit_stream.c:947).
//
// What is format-normative (RFC 7932, re-derived by C/brotli/br_decode.c) and reproduced exactly: the stream header
// (WBITS), meta-block header (ISLAST, MNIBBLES, MLEN-1, ISUNCOMPRESSED, NBLTYPES*, NPOSTFIX, NDIRECT, context modes,
// NTREES*), simple and complex prefix code descriptions (code-length code in the fixed storage order with its fixed
// variable-length code, zero-run symbol 17, "stop when the Kraft sum is full"), canonical code assignment with bits sent
// least-significant-first (so code words are stored bit-reversed), the 704-cell insert&copy alphabet, distance codes
// (16 ring-buffer codes + 48 direct codes at NPOSTFIX = NDIRECT = 0) and the order cmd code, insert extra, copy extra,
// literals, distance code, distance extra.  Free choices made here: one block type per category, one distance tree, NPOSTFIX = NDIRECT = 0.
// Literal context modelling (round 6, quality >= 5 as in the reference, MIN_QUALITY_FOR_CONTEXT_MODELING): per meta-block either ONE literal tree or THIRTEEN, the
// literal's context id taken from its two predecessors in CONTEXT_UTF8 mode (RFC 7932 section 7.1, br_utf8_lut0 / br_utf8_lut1 below) and mapped to a tree by the static map
// of the reference's ShouldUseComplexStaticContextMap (C/brotli/br_encode.c:328-405: no clustering); the choice is made from the meta-block's own literal histograms (the
// reference samples 64 bytes in every 4 KiB of input, :358-375), context map coded per RFC 7932 section 7.3 (C/brotli/br_brotli_bit_stream.c:592-735) without run lengths.
//
// Every meta-block is followed by an empty metadata meta-block (6 bits + padding, RFC 7932 section 9.2), which byte-aligns
// the next one: meta-blocks are therefore independent byte strings that L-emit concatenates, and a 128 KiB block is one
// workgroup's work item like in the zstd and LZMA2 paths.
//
// Parallel structure (256 threads): merge of capped match records (scan), symbol histograms (LDS atomics), three Huffman
// codes (rank sort in parallel, tree + length limiting in one lane), header (one lane, a few hundred symbols), then the bit
// stream: per-command bit counts -> workgroup prefix sums -> every command and every literal ORs its bits at its absolute
// bit offset (global atomic OR into the zeroed staging area); literal runs longer than BR_LONG are handled cooperatively.
#include "gc_common.h"
#include "gc_device.h"
#include "gc_fse.h"
#include "gc_brotli.h"
#ifdef HIPEMU
#include <stdio.h>
#include <stdlib.h>
#endif
#include "gc_lz_parse.h"       // pz_log2_q8: integer logarithm (the same on the device and under the emulator)

#define BR_T 256u
#define BR_WIN_WORDS 2048u     // 8 KiB: the bits of one tile of 256 commands are assembled in LDS and leave as full words
#define BR_LONG 48u            // literal runs up to this length are walked by the command's own lane

// ---- format tables (RFC 7932 section 5)
__constant__ uint16_t kInsBase[24] = { 0,1,2,3,4,5,6,8,10,14,18,26,34,50,66,98,130,194,322,578,1090,2114,6210,22594 };
__constant__ uint8_t  kInsExtra[24] = { 0,0,0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,7,8,9,10,12,14,24 };
__constant__ uint16_t kCopyBase[24] = { 2,3,4,5,6,7,8,9,10,12,14,18,22,30,38,54,70,102,134,198,326,582,1094,2118 };
__constant__ uint8_t  kCopyExtra[24] = { 0,0,0,0,0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,7,8,9,10,24 };
// first symbol of the insert&copy cell for (insert code >> 3, copy code >> 3), explicit distance
__constant__ uint16_t kCellBase[9] = { 128, 192, 384, 256, 320, 512, 448, 576, 640 };

__device__ __forceinline__ uint32_t br_ins_code(uint32_t v)
{
    if (v < 6u) return v;
    if (v < 130u) { const uint32_t nb = gc_hibit32(v - 2u) - 1u; return (nb << 1) + ((v - 2u) >> nb) + 2u; }
    if (v < 2114u) return gc_hibit32(v - 66u) + 10u;
    return v < 6210u ? 21u : (v < 22594u ? 22u : 23u);
}
__device__ __forceinline__ uint32_t br_copy_code(uint32_t v)
{
    if (v < 10u) return v - 2u;
    if (v < 134u) { const uint32_t nb = gc_hibit32(v - 6u) - 1u; return (nb << 1) + ((v - 6u) >> nb) + 4u; }
    if (v < 2118u) return gc_hibit32(v - 70u) + 12u;
    return 23u;
}
__device__ __forceinline__ uint32_t br_rev(uint32_t code, uint32_t nbits) { return __brev(code) >> (32u - nbits); }

// one command: everything needed to count and to write it
struct BrCmd { uint32_t sym; uint32_t insExtraBits, insExtraVal, copyExtraBits, copyExtraVal; uint32_t dsym, dExtraBits, dExtraVal; bool hasDist; };
// ll literals then a copy of ml bytes at distance off; ml == 0: trailing insert-only command.  useLast: off equals the
// previous copy's distance of this meta-block
// rcode 1..15: the distance is one of the ring-buffer short codes (second / third / fourth last distance, last or second last -+ 1..3: RFC 7932
// section 4, no extra bits), found by br_ring_code; 0: none
__device__ __forceinline__ BrCmd br_command(uint32_t ll, uint32_t ml, uint32_t off, bool useLast, uint32_t rcode = 0u)
{
    BrCmd c;
    const uint32_t ic = br_ins_code(ll), cc = ml ? br_copy_code(ml) : 0u;
    c.insExtraBits = kInsExtra[ic]; c.insExtraVal = ll - kInsBase[ic];
    c.copyExtraBits = ml ? kCopyExtra[cc] : 0u; c.copyExtraVal = ml ? ml - kCopyBase[cc] : 0u;
    const uint32_t low = ((ic & 7u) << 3) | (cc & 7u);
    const bool implicitCell = (useLast || ml == 0u) && ic < 8u && cc < 16u;       // cells 0..127: distance code 0 implied
    if (implicitCell) { c.sym = low | (cc < 8u ? 0u : 64u); c.hasDist = false; }
    else { c.sym = kCellBase[(cc >> 3) + 3u * (ic >> 3)] | low; c.hasDist = ml != 0u; }
    c.dsym = 0; c.dExtraBits = 0; c.dExtraVal = 0;
    if (c.hasDist && !useLast && rcode != 0u) c.dsym = rcode;
    else if (c.hasDist && !useLast) {
        const uint32_t v = off + 3u, n = gc_hibit32(v) - 1u, b = (v >> n) & 1u;
        c.dsym = 16u + 2u * (n - 1u) + b; c.dExtraBits = n; c.dExtraVal = v - ((2u + b) << n);
    }
    return c;
}

// The short distance code of command j (a copy whose distance differs from the previous command's), 0 if none applies.  The decoder keeps the last
// four distances that were NOT coded with symbol 0 (br_decode.c:2277, TakeDistanceFromRingBuffer :1690-1715); since every command here uses symbol
// 0 whenever its distance equals the previous one, those are the distances of the last four "heads" -- commands whose distance differs from their
// predecessor's -- in front of j.  They are found by walking back over the packed commands (at most BR_RING_WALK of them; heads that lie further
// back, or in an earlier meta-block, are simply not used: an entry is only ever named when it is known exactly).
#define BR_RING_WALK 24u
__device__ __forceinline__ uint32_t br_ring_code(const uint64_t* P, uint32_t j, uint32_t off)
{
    uint32_t ring[4] = { 0u, 0u, 0u, 0u }, n = 0;
    uint32_t cur = (uint32_t)(P[j - 1u] >> 36) & 0xFFFFFFu;          // (j >= 1) the previous command's distance = last distance
    ring[n++] = cur;
    for (uint32_t k = 1; k < BR_RING_WALK && n < 4u && k < j; k++) {
        const uint32_t d = (uint32_t)(P[j - 1u - k] >> 36) & 0xFFFFFFu;
        if (d != cur) { ring[n++] = d; cur = d; }                    // P[j - k] was a head: the distance in front of it is the next older entry
    }
    // the walk stopped inside a run of equal distances or at the meta-block start: the oldest entry found is exact only if its run was left,
    // i.e. if it was recorded at a change -- entries are recorded at changes, except ring[0], which is always exact
    if (n >= 2u && off == ring[1]) return 1u;
    if (n >= 3u && off == ring[2]) return 2u;
    if (n >= 4u && off == ring[3]) return 3u;
    const int d0 = (int)off - (int)ring[0];
    if (d0 >= -3 && d0 <= 3 && d0 != 0) return 4u + (uint32_t)(d0 < 0 ? 2 * (-d0 - 1) : 2 * (d0 - 1) + 1);      // 4: -1, 5: +1, 6: -2, 7: +2, 8: -3, 9: +3
    if (n >= 2u) {
        const int d1 = (int)off - (int)ring[1];
        if (d1 >= -3 && d1 <= 3 && d1 != 0) return 10u + (uint32_t)(d1 < 0 ? 2 * (-d1 - 1) : 2 * (d1 - 1) + 1);
    }
    return 0u;
}

// eight source bytes from any address; are two byte ranges of the source equal (8 bytes at a time, never a byte beyond n)
__device__ __forceinline__ uint64_t br_ld8(const uint8_t* p) { uint64_t w; __builtin_memcpy(&w, p, 8); return w; }
__device__ __forceinline__ bool br_same(const uint8_t* a, const uint8_t* c, uint32_t n)
{
    uint32_t i = 0;
    for (; i + 8u <= n; i += 8u) if (br_ld8(a + i) != br_ld8(c + i)) return false;
    for (; i < n; i++) if (a[i] != c[i]) return false;
    return true;
}

// OR up to 64 bits into the (zero-initialised) global bit buffer at absolute bit offset `pos`
__device__ __forceinline__ void br_or_bits(uint32_t* buf, uint64_t pos, uint64_t v, uint32_t nbits)
{
    if (nbits == 0u) return;
    const uint64_t word = pos >> 5; const uint32_t sh = (uint32_t)pos & 31u;
    const uint32_t w0 = (uint32_t)(v << sh);
    const uint64_t rest = sh ? (v >> (32u - sh)) : (v >> 32);
    if (w0) atomicOr(&buf[word], w0);
    if ((uint32_t)rest) atomicOr(&buf[word + 1], (uint32_t)rest);
    if ((uint32_t)(rest >> 32)) atomicOr(&buf[word + 2], (uint32_t)(rest >> 32));
}
// the same into an LDS window (bit offset relative to the window's first word)
__device__ __forceinline__ void br_or_bits_lds(uint32_t* win, uint32_t pos, uint64_t v, uint32_t nbits)
{
    if (nbits == 0u) return;
    const uint32_t word = pos >> 5, sh = pos & 31u;
    const uint32_t w0 = (uint32_t)(v << sh);
    const uint64_t rest = sh ? (v >> (32u - sh)) : (v >> 32);
    if (w0) atomicOr(&win[word], w0);
    if ((uint32_t)rest) atomicOr(&win[word + 1u], (uint32_t)rest);
    if ((uint32_t)(rest >> 32)) atomicOr(&win[word + 2u], (uint32_t)(rest >> 32));
}

// ---------------------------------------------------------------------------------------------------------------------
// Length-limited Huffman code for an alphabet of n <= 704 symbols, executed by the whole workgroup.
// count[] -> depth[] (0 = unused) and bit-reversed canonical codes.  Returns the number of used symbols; with exactly one
// used symbol its depth is 0 (brotli's NSYM = 1 simple code: the symbol costs no bits).
struct BrHufScratch { uint16_t sorted[704]; uint32_t w[1408]; uint16_t parent[1408]; uint8_t d[1408]; uint32_t misc[4]; };

__device__ uint32_t br_build_code(const uint32_t* count, uint32_t n, uint32_t maxBits, uint8_t* depth, uint16_t* code, BrHufScratch& S, bool* ok)
{
    const uint32_t t = threadIdx.x;
    if (t == 0) S.misc[0] = 0;
    __syncthreads();
    // rank sort of the used symbols by (count, symbol)
    for (uint32_t s = t; s < n; s += BR_T) {
        const uint32_t c = count[s];
        depth[s] = 0; code[s] = 0;
        if (c) {
            uint32_t rank = 0;
            for (uint32_t u = 0; u < n; u++) { const uint32_t cu = count[u]; rank += (cu && (cu < c || (cu == c && u < s))) ? 1u : 0u; }
            S.sorted[rank] = (uint16_t)s;
            atomicAdd(&S.misc[0], 1u);
        }
    }
    __syncthreads();
    const uint32_t nsym = S.misc[0];
    if (t == 0) {
        S.misc[1] = 1u;
        if (nsym >= 2u) {
            for (uint32_t i = 0; i < nsym; i++) S.w[i] = count[S.sorted[i]];
            uint32_t li = 0, ii = nsym, ni = nsym;                       // two-queue merge over leaves sorted ascending
            for (uint32_t k = 0; k + 1u < nsym; k++) {
                uint32_t a, c;
                if (li < nsym && (ii >= ni || S.w[li] <= S.w[ii])) a = li++; else a = ii++;
                if (li < nsym && (ii >= ni || S.w[li] <= S.w[ii])) c = li++; else c = ii++;
                S.w[ni] = S.w[a] + S.w[c]; S.parent[a] = (uint16_t)ni; S.parent[c] = (uint16_t)ni; ni++;
            }
            const uint32_t root = 2u * nsym - 2u;
            S.d[root] = 0;
            for (int i = (int)root - 1; i >= 0; i--) { const uint32_t dd = S.d[S.parent[i]] + 1u; S.d[i] = (uint8_t)(dd > 255u ? 255u : dd); }
            uint32_t maxLen = 0;
            for (uint32_t i = 0; i < nsym; i++) maxLen = max(maxLen, (uint32_t)S.d[i]);
            if (maxLen > maxBits) {
                // clamp, then repay the Kraft debt from the least frequent symbols upward (same scheme as the HUF literals kernel)
                long long debt = 0;
                for (uint32_t i = 0; i < nsym; i++) { if (S.d[i] > maxBits) S.d[i] = (uint8_t)maxBits; debt += 1ll << (maxBits - S.d[i]); }
                debt -= 1ll << maxBits;
                for (int bl = (int)maxBits - 1; bl >= 1 && debt > 0; bl--) {
                    const long long r = 1ll << (maxBits - 1 - bl);
                    for (uint32_t i = 0; i < nsym && debt > 0; i++) if (S.d[i] == bl) { S.d[i] = (uint8_t)(bl + 1); debt -= r; }
                }
                for (int i = (int)nsym - 1; i >= 0 && debt < 0; i--) if (S.d[i] == maxBits) { S.d[i] = (uint8_t)(maxBits - 1u); debt += 1; }
                if (debt != 0) S.misc[1] = 0u;
            }
            for (uint32_t i = 0; i < nsym; i++) depth[S.sorted[i]] = S.d[i];
            // canonical codes: shorter first, ascending symbol inside a length (br_entropy_encode.c:474); stored bit-reversed
            uint32_t blCount[17], next[17];
            for (uint32_t l = 0; l <= 16u; l++) blCount[l] = 0;
            for (uint32_t s = 0; s < n; s++) blCount[depth[s]]++;
            blCount[0] = 0;
            { uint32_t c = 0; for (uint32_t l = 1; l <= 16u; l++) { c = (c + blCount[l - 1u]) << 1; next[l] = c; } }
            for (uint32_t s = 0; s < n; s++) if (depth[s]) code[s] = (uint16_t)br_rev(next[depth[s]]++, depth[s]);
        }
    }
    __syncthreads();
    if (S.misc[1] == 0u) *ok = false;
    return nsym;
}

// ---- serial bit writer into LDS bytes (header only)
struct BrBW { uint8_t* out; uint64_t acc; uint32_t nbits; uint32_t bytes; };
__device__ __forceinline__ void bw_put(BrBW& w, uint32_t v, uint32_t nb)
{
    w.acc |= (uint64_t)v << w.nbits; w.nbits += nb;
    while (w.nbits >= 8u) { w.out[w.bytes++] = (uint8_t)w.acc; w.acc >>= 8; w.nbits -= 8u; }
}
__device__ __forceinline__ uint32_t bw_bits(const BrBW& w) { return w.bytes * 8u + w.nbits; }
// VarLenUint8 (NBLTYPES - 1, NTREES - 1: RFC 7932 section 9.2; StoreVarLenUint8, C/brotli/br_brotli_bit_stream.c:96-110)
__device__ __forceinline__ void bw_varlen8(BrBW& w, uint32_t v)
{
    if (v == 0u) { bw_put(w, 0u, 1); return; }
    const uint32_t nb = gc_hibit32(v);
    bw_put(w, 1u, 1); bw_put(w, nb, 3); bw_put(w, v - (1u << nb), nb);
}

// prefix code description (single lane).  alphaBits = ceil(log2(alphabet size)).  scratch: n + 32 uint16 entries.
__device__ void br_store_code(BrBW& w, const uint8_t* depth, uint32_t n, uint32_t alphaBits, uint32_t nsym, uint16_t* seq)
{
    if (nsym <= 1u) {                                                    // simple code, NSYM = 1 (also for an unused alphabet)
        uint32_t s = 0; for (uint32_t i = 0; i < n; i++) if (depth[i] || nsym == 0u) { s = i; break; }
        if (nsym == 1u) { /* depth is 0 for a lone symbol: find it through the caller-provided marker in seq[0] */ s = seq[0]; }
        bw_put(w, 1u, 2); bw_put(w, 0u, 2); bw_put(w, s, alphaBits);
        return;
    }
    // code-length symbol sequence up to the last used symbol; zero runs as symbol 17 (3..10 zeros, 3 extra bits); two
    // 17s are never adjacent (a plain 0 separates them), so the decoder's run-compounding rule never applies
    uint32_t last = 0; for (uint32_t i = 0; i < n; i++) if (depth[i]) last = i;
    uint32_t m = 0, hist[18]; for (int i = 0; i < 18; i++) hist[i] = 0;
    for (uint32_t i = 0; i <= last;) {
        if (depth[i]) { seq[m++] = depth[i]; hist[depth[i]]++; i++; continue; }
        uint32_t run = 0; while (depth[i + run] == 0) run++;            // stops at `last` at the latest
        if (run >= 3u) {
            const uint32_t r = run < 10u ? run : 10u;
            seq[m++] = (uint16_t)(17u | ((r - 3u) << 8)); hist[17]++; i += r;
            if (depth[i] == 0) { seq[m++] = 0; hist[0]++; i++; }
        } else { seq[m++] = 0; hist[0]++; i++; }
    }
    // Huffman code over the 18 code-length symbols, depth <= 5 (tiny: selection by repeated minimum)
    uint8_t cd[18]; uint16_t cc[18]; uint32_t used = 0, only = 0;
    for (int i = 0; i < 18; i++) { cd[i] = 0; cc[i] = 0; if (hist[i]) { used++; only = (uint32_t)i; } }
    if (used == 1u) cd[only] = 1;        // lone symbol: any non-zero length in the header; it then costs 0 bits per use
    else {
        // package-free construction: build the tree with a 36-node pool
        uint32_t wgt[36]; int par[36]; bool alive[36]; int nn = 18;
        for (int i = 0; i < 18; i++) { wgt[i] = hist[i]; par[i] = -1; alive[i] = hist[i] != 0; }
        for (uint32_t k = 0; k + 1u < used; k++) {
            int a = -1, b = -1;
            for (int i = 0; i < nn; i++) if (alive[i]) { if (a < 0 || wgt[i] < wgt[a]) { b = a; a = i; } else if (b < 0 || wgt[i] < wgt[b]) b = i; }
            wgt[nn] = wgt[a] + wgt[b]; par[nn] = -1; alive[nn] = true; alive[a] = alive[b] = false; par[a] = par[b] = nn; nn++;
        }
        uint32_t maxd = 0;
        for (int i = 0; i < 18; i++) if (hist[i]) { uint32_t d = 0; for (int p = i; par[p] >= 0; p = par[p]) d++; cd[i] = (uint8_t)d; maxd = max(maxd, d); }
        if (maxd > 5u) {                                                 // rare: flatten to a valid 5-bit-limited code
            // clamp + Kraft repair on 5 bits, least frequent symbols first
            int order[18], no = 0;
            for (int i = 0; i < 18; i++) if (hist[i]) order[no++] = i;
            for (int i = 1; i < no; i++) { int v = order[i], j = i - 1; while (j >= 0 && hist[order[j]] > hist[v]) { order[j + 1] = order[j]; j--; } order[j + 1] = v; }
            int debt = 0;
            for (int i = 0; i < no; i++) { if (cd[order[i]] > 5) cd[order[i]] = 5; debt += 1 << (5 - cd[order[i]]); }
            debt -= 32;
            for (int bl = 4; bl >= 1 && debt > 0; bl--) for (int i = 0; i < no && debt > 0; i++) if (cd[order[i]] == bl) { cd[order[i]] = (uint8_t)(bl + 1); debt -= 1 << (4 - bl); }
            for (int i = no - 1; i >= 0 && debt < 0; i--) if (cd[order[i]] == 5) { cd[order[i]] = 4; debt += 1; }
        }
        uint32_t blc[7], nx[7]; for (int l = 0; l < 7; l++) blc[l] = 0;
        for (int i = 0; i < 18; i++) blc[cd[i]]++;
        blc[0] = 0;
        { uint32_t c = 0; for (int l = 1; l <= 6; l++) { c = (c + blc[l - 1]) << 1; nx[l] = c; } }
        for (int i = 0; i < 18; i++) if (cd[i]) cc[i] = (uint16_t)br_rev(nx[cd[i]]++, cd[i]);
    }
    // code-length code lengths in the fixed order, with the fixed variable-length code (RFC 7932 section 3.5)
    const uint8_t order18[18] = { 1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15 };
    const uint8_t vlcBits[6] = { 2, 4, 3, 2, 2, 4 }, vlcVal[6] = { 0, 7, 3, 2, 1, 15 };
    uint32_t skip = 0, store = 18;
    if (used > 1u) while (store > 0u && cd[order18[store - 1u]] == 0) store--;
    if (cd[order18[0]] == 0 && cd[order18[1]] == 0) { skip = 2; if (cd[order18[2]] == 0) skip = 3; }
    bw_put(w, skip, 2);
    for (uint32_t i = skip; i < store; i++) { const uint32_t l = cd[order18[i]]; bw_put(w, vlcVal[l], vlcBits[l]); }
    if (used == 1u) cd[only] = 0;
    for (uint32_t i = 0; i < m; i++) {
        const uint32_t sy = seq[i] & 0xFFu;
        bw_put(w, cc[sy], cd[sy]);
        if (sy == 17u) bw_put(w, seq[i] >> 8, 3);
    }
}

// workgroup exclusive sum of a 64-bit-safe uint32 per thread with running total
__device__ __forceinline__ uint32_t br_excl_scan(uint32_t v, uint32_t* sWave, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t incl = gc_wave_incl_sum(v);
    if (lane == 63u) sWave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < BR_T / 64u; w++) { const uint32_t c = sWave[w]; if (w < wave) before += c; all += c; }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

// ---- literal context modelling: CONTEXT_UTF8 (RFC 7932 section 7.1; the reference holds the same function as a table, C/brotli/br_context.c:79-120).
// context id = lut0(last byte) | lut1(second last byte): for two ASCII bytes 4 * class(last) + class2(second last)
#define BR_NT 13u                      // literal trees of the static map (BROTLI_MAX_STATIC_CONTEXTS, C/brotli/enc/quality.h)
__host__ __device__ __forceinline__ uint32_t br_utf8_lut0(uint32_t b)
{
    if (b >= 192u) return 2u + (b & 1u);                                  // UTF-8 lead byte
    if (b >= 128u) return b & 1u;                                         // continuation byte
    const bool up = b >= 'A' && b <= 'Z', lo = b >= 'a' && b <= 'z';
    const uint32_t l = b | 0x20u;
    const bool vowel = l == 'a' || l == 'e' || l == 'i' || l == 'o' || l == 'u';
    uint32_t k = 3u;                                                      // other punctuation
    if (b == 9u || b == 10u || b == 13u) k = 1u;
    else if (b < 32u || b == 127u) k = 0u;
    else if (b == ' ') k = 2u;
    else if (b == '"' || b == '\'') k = 4u;
    else if (b == '%') k = 5u;
    else if (b == '(' || b == '<' || b == '[' || b == '{') k = 6u;
    else if (b == ')' || b == '>' || b == ']' || b == '}') k = 7u;
    else if (b == ',' || b == ';' || b == ':') k = 8u;
    else if (b == '.') k = 9u;
    else if (b == '=') k = 10u;
    else if (b >= '0' && b <= '9') k = 11u;
    else if (up) k = vowel ? 12u : 13u;
    else if (lo) k = vowel ? 14u : 15u;
    return 4u * k;
}
__host__ __device__ __forceinline__ uint32_t br_utf8_lut1(uint32_t b)
{
    if (b >= 224u) return 2u;                                             // (lead bytes of three- and four-byte sequences; the table of RFC 7932 section 7.1 has 0 for 128 .. 223)
    if (b >= 128u) return 0u;
    if (b <= 32u || b == 127u) return 0u;                                 // control, space
    if ((b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z')) return 2u;
    if (b >= 'a' && b <= 'z') return 3u;
    return 1u;                                                            // punctuation
}
// context id (0..63) -> literal tree (0..12): what follows a line feed, a space, an opening / closing bracket, a digit, an upper- / lower-case letter ... each gets a code of its
// own (the grouping of kStaticContextMapComplexUTF8, C/brotli/br_encode.c:330-347, written as the rule it encodes)
__host__ __device__ __forceinline__ uint32_t br_static_tree(uint32_t ctx)
{
    const uint32_t c1 = ctx >> 2, c2 = ctx & 3u;
    switch (c1) {
    case 0:  return c2 < 2u ? 11u : 12u;          // after a non-ASCII byte
    case 1:  return 0u;                            // line feed, tab
    case 2:  return c2 < 2u ? 1u : 9u;             // space: start of a word after punctuation / after a word
    case 3:  return 2u;
    case 4:  return 1u;                            // quotes
    case 5:  return c2 == 0u ? 8u : 3u;            // %
    case 6:  return 1u;                            // opening brackets
    case 7:  return 2u;                            // closing brackets
    case 8:  return c2 == 0u ? 8u : 4u;            // , ; :
    case 9:  return c2 == 0u ? 8u : (c2 == 1u ? 7u : 4u);      // .
    case 10: return c2 == 0u ? 8u : 0u;            // =
    case 11: return 3u;                            // digits
    case 12: case 13: return c2 == 2u ? 10u : 5u;  // upper case: inside an upper-case word / at the start of one
    default: return 6u;                            // lower case
    }
}
extern "C" void gc_brotli_context_tables(uint8_t lut[512], uint8_t map[64])      // (tests: the restated tables against the reference's, tests/test_brotli.py)
{
    for (uint32_t b = 0; b < 256u; b++) { lut[b] = (uint8_t)br_utf8_lut0(b); lut[256u + b] = (uint8_t)br_utf8_lut1(b); }
    for (uint32_t c = 0; c < 64u; c++) map[c] = (uint8_t)br_static_tree(c);
}
extern "C" __global__ void __launch_bounds__(BR_T)
gc_brotli_block_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, const GcSeqRaw* __restrict__ seqRaw,
                       const uint8_t* __restrict__ lit, const GcBlockMeta* __restrict__ meta,
                       uint64_t* __restrict__ seqPacked /* scratch: GC_MAX_SEQ_PER_BLOCK per block */,
                       uint32_t* __restrict__ seqLitStart /* scratch: GC_MAX_SEQ_PER_BLOCK per block */,
                       uint32_t blocksPerChunk /* 0xFFFFFFFF: ONE plain stream (no brotli-mt chunks) */, uint32_t plainFlags /* plain stream: bit 1 = this call is
                       not its first piece (no stream header), bit 2 = not its last (no closing meta-block) */, uint32_t repSub /* passes of the last-distance substitution */,
                       uint32_t ctxModel /* 1: literal context modelling may be chosen per meta-block (quality >= 5) */,
                       uint32_t* __restrict__ stage /* zeroed, GC_BR_STAGE_STRIDE bytes per block */, GcBrotliBlockInfo* __restrict__ info)
{
    __shared__ uint32_t hLit[BR_NT][256], hCmd[704], hDist[64], hMap[16];      // literal histograms per tree of the static context map (one tree: row 0)
    __shared__ uint8_t  dLit[BR_NT][256], dCmd[704], dDist[64], dMap[16];
    __shared__ uint16_t cLit[BR_NT][256], cCmd[704], cDist[64], cMap[16];
    __shared__ uint8_t  sLut[512], sMap[64];                               // CONTEXT_UTF8 lookup (last byte, second last byte); context id -> literal tree
    __shared__ uint32_t sNs[BR_NT], sOne[BR_NT];                           // per literal tree: symbols in use, the lone symbol
    __shared__ unsigned long long sAcc[BR_NT + 1u];                        // sum of h log2 h per tree / of the pooled histogram (units of 1/256 bit)
    __shared__ uint32_t sTot[BR_NT], sUsed[BR_NT + 1u];
    __shared__ BrHufScratch S;
    __shared__ uint16_t sSeq[704 + 32];
    __shared__ uint32_t sWave[8];
    __shared__ uint32_t sMisc[12];
    // two arrays live in space that is free by the time they are needed (55 KB would leave two workgroups per CU, 43 KB leaves three): the header bytes in the scratch of the code
    // construction (all codes are built before the header is written), the bit window of the command tiles in the literal histograms (dead once the codes exist)
    uint8_t* const sHdr = (uint8_t*)&S;
    uint32_t* const sBits = &hLit[0][0];
    static_assert(sizeof(BrHufScratch) >= 4096u && BR_NT * 256u >= BR_WIN_WORDS, "aliased arrays fit");

    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const uint64_t blockBase = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const uint32_t nRaw = meta[b].nSeqRaw, nLit = meta[b].nLit;
    const GcSeqRaw* R = seqRaw + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    const uint8_t* L = lit + (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    uint64_t* P = seqPacked + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint32_t* LS = seqLitStart + (uint64_t)b * GC_MAX_SEQ_PER_BLOCK;
    uint32_t* out = stage + (uint64_t)b * (GC_BR_STAGE_STRIDE / 4u);
    const bool plain = blocksPerChunk == 0xFFFFFFFFu;
    const bool firstInChunk = plain ? (b == 0u && !(plainFlags & 2u)) : (b % blocksPerChunk) == 0u;
    const bool lastInChunk = plain ? (blockBase + blockLen >= srcSize && !(plainFlags & 4u)) : (((b + 1u) % blocksPerChunk) == 0u || blockBase + blockLen >= srcSize);

    for (uint32_t i = t; i < BR_NT * 256u; i += BR_T) (&hLit[0][0])[i] = 0;
    for (uint32_t i = t; i < 704u; i += BR_T) hCmd[i] = 0;
    for (uint32_t i = t; i < 64u; i += BR_T) hDist[i] = 0;
    // The decoder's context of a literal is made of the two bytes in front of it IN ITS STREAM: none in front of a brotli-mt chunk's first byte.  A later piece of a plain
    // stream does not see the bytes in front of itself here: its first meta-block keeps one tree.
    const uint64_t chunkStart = plain ? 0ull : blockBase - (uint64_t)(b % blocksPerChunk) * GC_ZSTD_BLOCK_MAX;
    const bool ctxOk = ctxModel != 0u && !(plain && (plainFlags & 2u) != 0u && b == 0u);
    for (uint32_t i = t; i < 256u; i += BR_T) { sLut[i] = (uint8_t)br_utf8_lut0(i); sLut[256u + i] = (uint8_t)br_utf8_lut1(i); }
    if (t < 64u) sMap[t] = ctxOk ? (uint8_t)br_static_tree(t) : (uint8_t)0;
    if (t < 16u) hMap[t] = 0;
    if (t <= BR_NT) { sAcc[t] = 0ull; sUsed[t] = 0u; if (t < BR_NT) sTot[t] = 0u; }
    __syncthreads();

    // ---- merge chains of capped records (same offset, no literals in between) into commands: P[j] = ll | ml<<18 | off<<36,
    //      LS[j] = rank of the command's first literal in the literal stream
    uint32_t nSeq = 0;
    for (uint32_t tb = 0; tb < nRaw; tb += BR_T) {
        const uint32_t i = tb + t;
        uint32_t head = 0, ll = 0, off = 0, ml = 0, start = 0;
        if (i < nRaw) {
            const GcSeqRaw r = R[i];
            uint32_t prevRank = 0, prevOff = 0;
            if (i) { const GcSeqRaw q = R[i - 1u]; prevRank = q.litRank; prevOff = q.offml >> 8; }
            ll = r.litRank - prevRank; off = r.offml >> 8; ml = r.offml & 0xFFu; start = prevRank;
            head = (i == 0u || ll != 0u || off != prevOff) ? 1u : 0u;
            if (head) for (uint32_t k = i + 1u; k < nRaw; k++) {
                const GcSeqRaw c = R[k];
                if (c.litRank != r.litRank || (c.offml >> 8) != off) break;
                ml += c.offml & 0xFFu;
            }
        }
        uint32_t tot;
        const uint32_t j = nSeq + br_excl_scan(head, sWave, &tot);
        if (head) { P[j] = (uint64_t)ll | ((uint64_t)ml << 18) | ((uint64_t)off << 36); LS[j] = start; }
        nSeq += tot;
    }
    __syncthreads();
    // trailing literals form an insert-only command
    const uint32_t litBeforeTail = nRaw ? R[nRaw - 1u].litRank : 0u;
    const uint32_t tailLen = nLit - litBeforeTail;
    const uint32_t nCmd = nSeq + (tailLen ? 1u : 0u);
    if (t == 0 && tailLen) { P[nSeq] = (uint64_t)tailLen; LS[nSeq] = litBeforeTail; }
    __syncthreads();

    // ---- last-distance substitution (round 4; repSub passes): a copy of up to BR_SUB_MAX bytes whose bytes ALSO match at the previous command's distance
    //      takes that distance -- the same bytes come out, the parse keeps its shape, and the distance costs symbol 0 (or an implicit cell) instead of a code with
    //      extra bits.  The reference's hasher gets there by testing the last distances FIRST at every position (hash_longest_match64_inc.h:185); here the finder
    //      knows no parse, so the test runs on the commands.  A command's copy starts at blockBase + LS[j] + ll_j + (sum of the copy lengths in front of it):
    //      one scan per tile of commands.  Decisions of a pass use the distances as the pass found them (all reads of a tile, then its stores); any
    //      distance whose bytes match is valid whatever the neighbours did, a second pass picks up runs.
#define BR_SUB_MAX 512u
    for (uint32_t pass = 0; pass < repSub; pass++) {
        uint32_t carryM = 0;
        for (uint32_t tb = 0; tb < nSeq; tb += BR_T) {
            const uint32_t j = tb + t;
            uint32_t ml = 0, ll = 0, off = 0, prevOff = 0;
            if (j < nSeq) {
                const uint64_t pk = P[j];
                ll = (uint32_t)(pk & 0x3FFFFu); ml = (uint32_t)((pk >> 18) & 0x3FFFFu); off = (uint32_t)(pk >> 36) & 0xFFFFFFu;
                if (j) prevOff = (uint32_t)(P[j - 1u] >> 36) & 0xFFFFFFu;
            }
            uint32_t totM;
            const uint32_t before = carryM + br_excl_scan(ml, sWave, &totM);
            uint32_t subOff = 0;
            if (j >= 1u && j < nSeq && prevOff != off && ml <= BR_SUB_MAX) {
                const uint64_t pos = blockBase + LS[j] + ll + before;
                const uint8_t* a = src + pos;
                // the distances of the three commands in front, nearest first (the second / third of them are ring entries 1 / 2 where they differ: br_ring_code);
                // the walk ends at a command that is no copy or has this command's own distance (nothing to gain further back).  The first eight bytes of
                // the three candidates are fetched together, the rest is compared for the first one that passes
                uint32_t d[3] = { 0u, 0u, 0u };
                for (uint32_t k = 1; k <= 3u && k <= j; k++) {
                    const uint32_t dk = (uint32_t)(P[j - k] >> 36) & 0xFFFFFFu;
                    if (dk == 0u || dk == off) break;
                    d[k - 1u] = dk;
                }
                const bool wide = pos + 8u <= srcSize;                                 // (the copy may end with the input: no 8-byte read across that end)
                const uint64_t mask = ml >= 8u ? ~0ull : ((1ull << (8u * ml)) - 1ull);
                uint64_t wa = 0, wc[3] = { 0, 0, 0 };
                if (wide) { wa = br_ld8(a); for (int k = 0; k < 3; k++) if (d[k]) wc[k] = br_ld8(a - d[k]); }
                for (int k = 0; k < 3 && subOff == 0u; k++) {
                    if (d[k] == 0u) break;
                    const uint8_t* c = a - d[k];
                    bool same;
                    if (wide) same = ((wa ^ wc[k]) & mask) == 0ull && (ml <= 8u || br_same(a + 8, c + 8, ml - 8u));
                    else same = br_same(a, c, ml);
                    if (same) subOff = d[k];
                }
            }
            __syncthreads();
            if (subOff) P[j] = (P[j] & ~(0xFFFFFFull << 36)) | ((uint64_t)subOff << 36);
            __syncthreads();
            carryM += totM;
        }
    }

    // ---- short distance codes (round 3): bits 60..63 of a packed command (distances have 24 bits: WBITS = 24).  Computed a tile of commands at a
    //      time -- all reads of the tile's walks, then the stores (a walk reads the distance fields only, which never change)
    for (uint32_t tb = 0; tb < nSeq; tb += BR_T) {
        const uint32_t j = tb + t;
        uint32_t rcode = 0;
        if (j >= 1u && j < nSeq) {
            const uint32_t off = (uint32_t)(P[j] >> 36) & 0xFFFFFFu;
            if (off != ((uint32_t)(P[j - 1u] >> 36) & 0xFFFFFFu)) rcode = br_ring_code(P, j, off);
        }
        __syncthreads();
        if (rcode) P[j] |= (uint64_t)rcode << 60;
        __syncthreads();
    }

    // ---- histograms
    auto histFlat = [&]() {                                        // one tree: the literal stream as the parse wrote it
        for (uint32_t i = t * 4u; i < nLit; i += BR_T * 4u) {
            if (i + 4u <= nLit) {
                const uint32_t v = *(const uint32_t*)(L + i);
                atomicAdd(&hLit[0][v & 0xFFu], 1u); atomicAdd(&hLit[0][(v >> 8) & 0xFFu], 1u); atomicAdd(&hLit[0][(v >> 16) & 0xFFu], 1u); atomicAdd(&hLit[0][v >> 24], 1u);
            } else for (uint32_t k = i; k < nLit; k++) atomicAdd(&hLit[0][L[k]], 1u);
        }
    };
    // literal histograms per tree of the static map.  Where a literal stands in the INPUT: a command's first literal at block start + its literal rank + the copy lengths in
    // front of it (one scan per tile of commands, as the substitution above); its context: the two input bytes in front of it.  every: the literals of every n-th tile of
    // commands only (a sample for the decision below)
    auto histCtx = [&](uint32_t every) {
        uint32_t carryM = 0;
        for (uint32_t tb = 0; tb < nCmd; tb += BR_T) {
            const uint32_t j = tb + t;
            const bool sampled = ((tb / BR_T) % every) == 0u;      // uniform
            uint32_t ll = 0, ml = 0, ls = 0;
            if (j < nCmd) { const uint64_t pk = P[j]; ll = (uint32_t)(pk & 0x3FFFFu); ml = (uint32_t)((pk >> 18) & 0x3FFFFu); ls = LS[j]; }
            uint32_t totM;
            const uint64_t cs = blockBase + ls + carryM + br_excl_scan(ml, sWave, &totM);
            carryM += totM;
            if (!sampled) continue;
            const bool isLong = ll > BR_LONG;
            if (ll != 0u && !isLong) {
                uint32_t p1 = cs > chunkStart ? src[cs - 1u] : 0u, p2 = cs > chunkStart + 1u ? src[cs - 2u] : 0u;
                for (uint32_t i = 0; i < ll; i++) { const uint32_t by = src[cs + i]; atomicAdd(&hLit[sMap[sLut[p1] | sLut[256u + p2]]][by], 1u); p2 = p1; p1 = by; }
            }
            for (uint32_t w0 = 0; w0 < BR_T / 64u; w0++) {         // long runs: the whole workgroup, one after the other
                __syncthreads();
                if ((t >> 6) == w0) { const uint64_t m = __ballot(isLong); if ((t & 63u) == 0u) { sMisc[2] = (uint32_t)m; sMisc[3] = (uint32_t)(m >> 32); } }
                __syncthreads();
                uint64_t mask = (uint64_t)sMisc[2] | ((uint64_t)sMisc[3] << 32);
                while (mask) {
                    const uint32_t ln = gc_ctz64(mask); mask &= mask - 1ull;
                    const uint32_t jj = tb + w0 * 64u + ln;
                    if (j == jj) { sMisc[4] = (uint32_t)cs; sMisc[5] = (uint32_t)(cs >> 32); }
                    __syncthreads();
                    const uint64_t pos = (uint64_t)sMisc[4] | ((uint64_t)sMisc[5] << 32);
                    const uint32_t rl = (uint32_t)(P[jj] & 0x3FFFFu);
                    for (uint32_t i = t; i < rl; i += BR_T) {
                        const uint64_t q = pos + i;
                        const uint32_t p1 = q > chunkStart ? src[q - 1u] : 0u, p2 = q > chunkStart + 1u ? src[q - 2u] : 0u;
                        atomicAdd(&hLit[sMap[sLut[p1] | sLut[256u + p2]]][src[q]], 1u);
                    }
                    __syncthreads();
                }
            }
        }
    };
    // ---- one literal tree or thirteen?  Decided on a SAMPLE -- the literals of every eighth tile of 256 commands -- so that a meta-block that stays with one tree (web-text: all
    //      of them) pays an eighth of the walk; the reference samples 64 bytes in every 4 KiB (br_encode.c:358-375).  The cost of the literals under each choice from the sampled
    //      histograms: sum over trees of T log2 T - sum h log2 h (integer logarithm: the same choice on the device and under the emulator), plus what the tree descriptions and
    //      the context map cost (about 7 bits per symbol in use, 48 per tree, 360 for the map) and a margin of 1.5 % (thirteen small codes lose more to whole-bit lengths).
    uint32_t nTrees = 1u;
    if (ctxOk) {
        constexpr uint32_t EVERY = 8u;
        histCtx(EVERY);
        __syncthreads();
        uint32_t H = 0;
        for (uint32_t tr = 0; tr < BR_NT; tr++) {
            const uint32_t h = hLit[tr][t];                        // (BR_T = 256: thread t = literal value t)
            if (h) { atomicAdd(&sAcc[tr], (unsigned long long)h * pz_log2_q8(h)); atomicAdd(&sTot[tr], h); atomicAdd(&sUsed[tr], 1u); }
            H += h;
        }
        if (H) { atomicAdd(&sAcc[BR_NT], (unsigned long long)H * pz_log2_q8(H)); atomicAdd(&sUsed[BR_NT], 1u); }
        __syncthreads();
        if (t == 0) {
            unsigned long long c13 = 0, T = 0, d13 = 0;
            for (uint32_t tr = 0; tr < BR_NT; tr++) {
                const uint32_t tt = sTot[tr];
                if (tt) c13 += (unsigned long long)tt * pz_log2_q8(tt) - sAcc[tr];
                T += tt;
                d13 += sUsed[tr] ? 48u + 7u * sUsed[tr] : 8u;
            }
            const unsigned long long c1 = T ? T * pz_log2_q8((uint32_t)T) - sAcc[BR_NT] : 0ull, d1 = 48u + 7u * sUsed[BR_NT];
            const uint32_t scale = nCmd > BR_T ? EVERY : 1u;       // (what the sample stands for)
            sMisc[6] = (T * scale >= 512u && scale * (c13 + (c1 >> 6)) + ((d13 + 360u) << 8) < scale * c1 + (d1 << 8)) ? 1u : 0u;
        }
        __syncthreads();
        const bool use13 = sMisc[6] != 0u;
        for (uint32_t i = t; i < BR_NT * 256u; i += BR_T) (&hLit[0][0])[i] = 0;
        if (!use13 && t < 64u) sMap[t] = 0;                        // one tree: every context id maps to tree 0
        __syncthreads();
        if (use13) { nTrees = BR_NT; histCtx(1u); } else histFlat();
    } else histFlat();
    for (uint32_t j = t; j < nCmd; j += BR_T) {
        const uint64_t pk = P[j];
        const uint32_t ll = (uint32_t)(pk & 0x3FFFFu), ml = (uint32_t)((pk >> 18) & 0x3FFFFu), off = (uint32_t)(pk >> 36) & 0xFFFFFFu;
        const bool useLast = j > 0u && ml != 0u && ((uint32_t)(P[j - 1u] >> 36) & 0xFFFFFFu) == off;
        const BrCmd c = br_command(ll, ml, off, useLast, (uint32_t)(pk >> 60));
        atomicAdd(&hCmd[c.sym], 1u);
        if (c.hasDist) atomicAdd(&hDist[c.dsym], 1u);
    }
    __syncthreads();

    // ---- prefix codes
    bool ok = true;
    for (uint32_t tr = 0; tr < nTrees; tr++) {
        const uint32_t ns = br_build_code(hLit[tr], 256u, 15u, dLit[tr], cLit[tr], S, &ok);
        if (t == 0) { uint32_t one = 0; if (ns == 1u) { for (uint32_t s = 0; s < 256u; s++) if (hLit[tr][s]) one = s; } sNs[tr] = ns; sOne[tr] = one; }
    }
    uint32_t nsMap = 0, oneMap = 0;
    if (nTrees > 1u) {                                             // the context map's own prefix code: 64 entries over the alphabet of trees
        if (t < 64u) atomicAdd(&hMap[sMap[t]], 1u);
        __syncthreads();
        nsMap = br_build_code(hMap, BR_NT, 15u, dMap, cMap, S, &ok);
        if (nsMap == 1u) { for (uint32_t s = 0; s < BR_NT; s++) if (hMap[s]) oneMap = s; }
    }
#ifdef HIPEMU
    if (t == 0 && getenv("GC_BR_DEBUG")) { fprintf(stderr, "B1 block %u: nCmd %u nLit %u trees %u ok %d ns:", b, nCmd, nLit, nTrees, (int)ok); for (uint32_t tr = 0; tr < nTrees; tr++) fprintf(stderr, " %u", sNs[tr]); fprintf(stderr, "\n"); }
#endif
    const uint32_t nsCmd = br_build_code(hCmd, 704u, 15u, dCmd, cCmd, S, &ok);
    uint32_t oneCmd = 0;  if (nsCmd == 1u) { for (uint32_t s = 0; s < 704u; s++) if (hCmd[s]) oneCmd = s; }
    const uint32_t nsDist = br_build_code(hDist, 64u, 15u, dDist, cDist, S, &ok);
    uint32_t oneDist = 0; if (nsDist == 1u) { for (uint32_t s = 0; s < 64u; s++) if (hDist[s]) oneDist = s; }

    // ---- header: [stream header] meta-block header + the three code descriptions (one lane)
    if (t == 0) {
        BrBW w; w.out = sHdr; w.acc = 0; w.nbits = 0; w.bytes = 0;
        if (firstInChunk) bw_put(w, GC_BR_WBITS_CODE, 4);                  // WBITS = 24: '1' + n = 7 (RFC 7932 section 9.1); window 16 MiB - 16
        bw_put(w, 0u, 1);                                                  // ISLAST = 0 (the stream is closed by an empty last meta-block)
        const uint32_t mlen1 = blockLen - 1u;
        const uint32_t nib = mlen1 < (1u << 16) ? 4u : (mlen1 < (1u << 20) ? 5u : 6u);
        bw_put(w, nib - 4u, 2); bw_put(w, mlen1, nib * 4u);
        bw_put(w, 0u, 1);                                                  // ISUNCOMPRESSED = 0
        sMisc[1] = bw_bits(w);                                             // bits before ISUNCOMPRESSED + 1: reused by the stored fallback
        bw_put(w, 0u, 1); bw_put(w, 0u, 1); bw_put(w, 0u, 1);              // NBLTYPESL = NBLTYPESI = NBLTYPESD = 1
        bw_put(w, 0u, 2); bw_put(w, 0u, 4);                                // NPOSTFIX = 0, NDIRECT = 0
        bw_put(w, nTrees > 1u ? 2u : 0u, 2);                               // context mode of the single literal block type: 2 = UTF8 (0 = LSB6: unused with one tree)
        bw_varlen8(w, nTrees - 1u);                                        // NTREESL
        if (nTrees > 1u) {                                                 // literal context map (RFC 7932 section 7.3): no run-length codes, the 64 entries, no inverse move-to-front
            bw_put(w, 0u, 1);                                              // RLEMAX = 0
            sSeq[0] = (uint16_t)oneMap; br_store_code(w, dMap, BR_NT, 4u, nsMap, sSeq);
            for (uint32_t c = 0; c < 64u; c++) bw_put(w, cMap[sMap[c]], dMap[sMap[c]]);
            bw_put(w, 0u, 1);                                              // IMTF = 0
        }
        bw_put(w, 0u, 1);                                                  // NTREESD = 1
        for (uint32_t tr = 0; tr < nTrees; tr++) { sSeq[0] = (uint16_t)sOne[tr]; br_store_code(w, dLit[tr], 256u, 8u, sNs[tr], sSeq); }
        sSeq[0] = (uint16_t)oneCmd;  br_store_code(w, dCmd, 704u, 10u, nsCmd, sSeq);
        sSeq[0] = (uint16_t)oneDist; br_store_code(w, dDist, 64u, 6u, nsDist, sSeq);
        const uint32_t hb = bw_bits(w);
        if (w.nbits) { w.out[w.bytes++] = (uint8_t)w.acc; }
        sMisc[0] = hb;
    }
    __syncthreads();
    const uint32_t hdrBits = sMisc[0];
#ifdef HIPEMU
    if (t == 0 && getenv("GC_BR_DEBUG")) fprintf(stderr, "B1 block %u: header %u bits\n", b, hdrBits);
#endif
    for (uint32_t i = t; i < (hdrBits + 31u) / 32u; i += BR_T) {
        uint32_t v = 0; for (uint32_t k = 0; k < 4u; k++) { const uint32_t bi = i * 4u + k; if (bi * 8u < hdrBits) v |= (uint32_t)sHdr[bi] << (8u * k); }
        if (v) atomicOr(&out[i], v);
    }

    // ---- commands: bit counts -> offsets -> bits.  A command's literals are walked by its own lane unless the run is long.
    uint64_t bitBase = hdrBits;                                            // uniform running bit offset
    uint32_t carryC = 0;                                                   // copy lengths of the commands in front of the tile (several trees: where a command's literals stand in the input)
    const bool ctx = nTrees > 1u;                                          // uniform
    for (uint32_t tb = 0; tb < nCmd; tb += BR_T) {
        const uint32_t j = tb + t;
        uint32_t ll = 0, ml = 0, off = 0, ls = 0; bool valid = j < nCmd, useLast = false;
        BrCmd c; c.sym = 0; c.insExtraBits = c.insExtraVal = c.copyExtraBits = c.copyExtraVal = c.dsym = c.dExtraBits = c.dExtraVal = 0; c.hasDist = false;
        if (valid) {
            const uint64_t pk = P[j];
            ll = (uint32_t)(pk & 0x3FFFFu); ml = (uint32_t)((pk >> 18) & 0x3FFFFu); off = (uint32_t)(pk >> 36) & 0xFFFFFFu; ls = LS[j];
            useLast = j > 0u && ml != 0u && ((uint32_t)(P[j - 1u] >> 36) & 0xFFFFFFu) == off;
            c = br_command(ll, ml, off, useLast, (uint32_t)(pk >> 60));
        }
        const uint32_t headBits = valid ? dCmd[c.sym] + c.insExtraBits + c.copyExtraBits : 0u;
        const uint32_t tailBits = (valid && c.hasDist) ? dDist[c.dsym] + c.dExtraBits : 0u;
        uint32_t litBits = 0;
        const bool isLong = valid && ll > BR_LONG;
        uint64_t cs = 0;                                                   // absolute position of the command's first literal
        if (ctx) { uint32_t totM; cs = blockBase + ls + carryC + br_excl_scan(valid ? ml : 0u, sWave, &totM); carryC += totM; }
        uint32_t c1 = 0, c2 = 0;                                           // the two bytes in front of it
        if (ctx && valid && ll != 0u) { c1 = cs > chunkStart ? src[cs - 1u] : 0u; c2 = cs > chunkStart + 1u ? src[cs - 2u] : 0u; }
        if (valid && !isLong) {
            if (!ctx) for (uint32_t i = 0; i < ll; i++) litBits += dLit[0][L[ls + i]];
            else { uint32_t p1 = c1, p2 = c2; for (uint32_t i = 0; i < ll; i++) { const uint32_t by = L[ls + i]; litBits += dLit[sMap[sLut[p1] | sLut[256u + p2]]][by]; p2 = p1; p1 = by; } }
        }
        // long runs of this tile: summed by the whole workgroup, one after the other
        for (uint32_t w0 = 0; w0 < BR_T / 64u; w0++) {
            __syncthreads();
            if ((t >> 6) == w0) { const uint64_t m = __ballot(isLong); if ((t & 63u) == 0u) { sMisc[2] = (uint32_t)m; sMisc[3] = (uint32_t)(m >> 32); } }
            __syncthreads();
            uint64_t mask = (uint64_t)sMisc[2] | ((uint64_t)sMisc[3] << 32);
            while (mask) {
                const uint32_t ln = gc_ctz64(mask); mask &= mask - 1ull;
                const uint32_t jj = tb + w0 * 64u + ln;
                const uint32_t rl = (uint32_t)(P[jj] & 0x3FFFFu), rs = LS[jj];
                uint32_t part = 0;
                if (!ctx) for (uint32_t i = t; i < rl; i += BR_T) part += dLit[0][L[rs + i]];
                else {
                    if (j == jj) { sMisc[8] = (uint32_t)cs; sMisc[9] = (uint32_t)(cs >> 32); }
                    __syncthreads();
                    const uint64_t pos0 = (uint64_t)sMisc[8] | ((uint64_t)sMisc[9] << 32);
                    for (uint32_t i = t; i < rl; i += BR_T) {
                        const uint64_t q = pos0 + i;
                        const uint32_t p1 = q > chunkStart ? src[q - 1u] : 0u, p2 = q > chunkStart + 1u ? src[q - 2u] : 0u;
                        part += dLit[sMap[sLut[p1] | sLut[256u + p2]]][L[rs + i]];
                    }
                }
                uint32_t tot; br_excl_scan(part, sWave, &tot);
                if (j == jj) litBits = tot;
            }
        }
        uint32_t tileBits;
        const uint32_t myBits = headBits + litBits + tailBits;
        const uint64_t q = bitBase + br_excl_scan(myBits, sWave, &tileBits);
        if (bitBase + tileBits + 64u >= (uint64_t)blockLen * 8u) { ok = false; break; }   // no gain: stored meta-block; nothing past the raw size is ever written
        // The tile's bits form one contiguous range of the stream.  Without long literal runs (those are written cooperatively
        // below) and if the range fits the window, it is assembled in LDS and leaves as whole words: plain stores for the words
        // that lie entirely inside the range, an atomic OR only for its first and last word (shared with the neighbouring tiles /
        // the header).  Bit by bit into HBM, every output word took ~5 global atomics (PMC: 6 x the algorithmic bytes).
        uint32_t nLongTile; br_excl_scan(isLong ? 1u : 0u, sWave, &nLongTile);
        const uint64_t word0 = bitBase >> 5;
        const uint32_t nWinWords = (uint32_t)(((bitBase + tileBits + 31u) >> 5) - word0);
        const bool useWin = nLongTile == 0u && tileBits != 0u && nWinWords + 2u <= BR_WIN_WORDS;      // uniform
        if (useWin) { for (uint32_t i = t; i < nWinWords + 2u; i += BR_T) sBits[i] = 0; __syncthreads(); }
        if (valid) {
            uint64_t hv = cCmd[c.sym]; uint32_t hn = dCmd[c.sym];
            hv |= (uint64_t)c.insExtraVal << hn; hn += c.insExtraBits;
            hv |= (uint64_t)c.copyExtraVal << hn; hn += c.copyExtraBits;
            if (useWin) {
                const uint32_t rq = (uint32_t)(q - (word0 << 5));
                br_or_bits_lds(sBits, rq, hv, hn);
                { uint32_t pos = rq + headBits, p1 = c1, p2 = c2;
                  for (uint32_t i = 0; i < ll; i++) { const uint32_t sy = L[ls + i], tr = ctx ? sMap[sLut[p1] | sLut[256u + p2]] : 0u; br_or_bits_lds(sBits, pos, cLit[tr][sy], dLit[tr][sy]); pos += dLit[tr][sy]; p2 = p1; p1 = sy; } }
                if (c.hasDist) br_or_bits_lds(sBits, rq + headBits + litBits, (uint64_t)cDist[c.dsym] | ((uint64_t)c.dExtraVal << dDist[c.dsym]), dDist[c.dsym] + c.dExtraBits);
            } else {
                br_or_bits(out, q, hv, hn);
                if (!isLong) { uint64_t pos = q + headBits; uint32_t p1 = c1, p2 = c2;
                               for (uint32_t i = 0; i < ll; i++) { const uint32_t sy = L[ls + i], tr = ctx ? sMap[sLut[p1] | sLut[256u + p2]] : 0u; br_or_bits(out, pos, cLit[tr][sy], dLit[tr][sy]); pos += dLit[tr][sy]; p2 = p1; p1 = sy; } }
                if (c.hasDist) br_or_bits(out, q + headBits + litBits, (uint64_t)cDist[c.dsym] | ((uint64_t)c.dExtraVal << dDist[c.dsym]), dDist[c.dsym] + c.dExtraBits);
            }
        }
        if (useWin) {
            __syncthreads();
            for (uint32_t i = t; i < nWinWords; i += BR_T) {
                const uint32_t v = sBits[i];
                if (v) { if (i == 0u || i + 1u == nWinWords) atomicOr(&out[word0 + i], v); else out[word0 + i] = v; }
            }
        }
        // long runs: cooperative write, 256 literals per step with a running bit offset
        for (uint32_t w0 = 0; w0 < BR_T / 64u; w0++) {
            __syncthreads();
            if ((t >> 6) == w0) { const uint64_t m = __ballot(isLong); if ((t & 63u) == 0u) { sMisc[2] = (uint32_t)m; sMisc[3] = (uint32_t)(m >> 32); } }
            __syncthreads();
            uint64_t mask = (uint64_t)sMisc[2] | ((uint64_t)sMisc[3] << 32);
            while (mask) {
                const uint32_t ln = gc_ctz64(mask); mask &= mask - 1ull;
                const uint32_t jj = tb + w0 * 64u + ln;
                if (j == jj) { sMisc[4] = (uint32_t)(q + headBits); sMisc[5] = (uint32_t)((q + headBits) >> 32); sMisc[8] = (uint32_t)cs; sMisc[9] = (uint32_t)(cs >> 32); }
                __syncthreads();
                uint64_t pos = (uint64_t)sMisc[4] | ((uint64_t)sMisc[5] << 32);
                const uint64_t pos0 = (uint64_t)sMisc[8] | ((uint64_t)sMisc[9] << 32);
                const uint32_t rl = (uint32_t)(P[jj] & 0x3FFFFu), rs = LS[jj];
                for (uint32_t i0 = 0; i0 < rl; i0 += BR_T) {
                    const uint32_t i = i0 + t;
                    uint32_t sy = 0, nb = 0, tr = 0;
                    if (i < rl) {
                        sy = L[rs + i];
                        if (ctx) { const uint64_t qq = pos0 + i; const uint32_t p1 = qq > chunkStart ? src[qq - 1u] : 0u, p2 = qq > chunkStart + 1u ? src[qq - 2u] : 0u; tr = sMap[sLut[p1] | sLut[256u + p2]]; }
                        nb = dLit[tr][sy];
                    }
                    uint32_t tot;
                    const uint32_t at = br_excl_scan(nb, sWave, &tot);
                    if (i < rl) br_or_bits(out, pos + at, cLit[tr][sy], nb);
                    pos += tot;
                }
            }
        }
        bitBase += tileBits;
    }

    // ---- close: an empty metadata meta-block byte-aligns the next meta-block; the chunk's last block also appends the
    //      empty last meta-block (ISLAST = 1, ISLASTEMPTY = 1)
    if (t == 0) {
        uint64_t e = bitBase;
        const uint64_t rawLimit = (uint64_t)blockLen * 8u;
        const bool stored = !ok || e + 64u >= rawLimit;
        GcBrotliBlockInfo bi;
        if (!stored) {
            br_or_bits(out, e, 0x06u, 6); e += 6u;                         // ISLAST=0, MNIBBLES=11 (metadata), reserved 0, MSKIPBYTES=00
            e = (e + 7u) & ~7ull;
            if (lastInChunk) { br_or_bits(out, e, 0x3u, 2); e += 8u; }
            bi.size = (uint32_t)(e >> 3); bi.stored = 0; bi.hdrBits = 0;
        } else {
            // stored meta-block: header up to ISUNCOMPRESSED = 1, padding, raw bytes (emit kernel), then as above
            bi.stored = 1; bi.hdrBits = sMisc[1];
            const uint32_t hb = (sMisc[1] + 7u) >> 3;
            bi.size = hb + blockLen + (lastInChunk ? 1u : 0u);
        }
        bi.lastInChunk = lastInChunk ? 1u : 0u;
        info[b] = bi;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// plan: one workgroup; exclusive scan of block sizes with a 16-byte brotli-mt frame header in front of every chunk
// (C/zstdmt/brotli-mt_compress.c:299-321: 0x184D2A50, 8, compressed size, 0x5242, hint = 64 KiB units to allocate)
extern "C" __global__ void __launch_bounds__(1024)
gc_brotli_plan_kernel(const GcBrotliBlockInfo* __restrict__ info, uint32_t nBlocks, uint32_t blocksPerChunk /* 0xFFFFFFFF: plain stream, no brotli-mt frame headers */,
                      uint64_t dstCap, GcBrotliPlan* __restrict__ plan, uint64_t* __restrict__ result)
{
    const bool framed = blocksPerChunk != 0xFFFFFFFFu;
    __shared__ uint32_t sWave[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint64_t carry = 0;
    for (uint32_t tb = 0; tb < nBlocks; tb += 1024u) {
        const uint32_t b = tb + t;
        uint32_t size = 0;
        if (b < nBlocks) size = info[b].size + ((framed && (b % blocksPerChunk) == 0u) ? 16u : 0u);
        const uint32_t incl = gc_wave_incl_sum(size);
        if (lane == 63u) sWave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 16u; w++) { const uint32_t v = sWave[w]; if (w < wave) before += v; all += v; }
        __syncthreads();
        if (b < nBlocks) { GcBrotliPlan p; p.off = carry + before + incl - size; p.chunkSize = 0; p.pad = 0; plan[b] = p; }
        carry += all;
    }
    if (t == 0) { result[0] = carry; result[1] = carry > dstCap ? 1u : 0u; }
}

// emit: one workgroup per block
extern "C" __global__ void __launch_bounds__(256)
gc_brotli_emit_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, const uint8_t* __restrict__ stage,
                      const GcBrotliBlockInfo* __restrict__ info, const GcBrotliPlan* __restrict__ plan, uint32_t nBlocks,
                      uint32_t blocksPerChunk, uint32_t plainFlags, const uint64_t* __restrict__ result, uint8_t* __restrict__ dst)
{
    if (result[1]) return;
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const uint64_t blockBase = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - blockBase) < GC_ZSTD_BLOCK_MAX ? (srcSize - blockBase) : GC_ZSTD_BLOCK_MAX);
    const GcBrotliBlockInfo bi = info[b];
    uint8_t* o = dst + plan[b].off;
    const bool first = blocksPerChunk == 0xFFFFFFFFu ? (b == 0u && !(plainFlags & 2u)) : (b % blocksPerChunk) == 0u;    // first block of a stream: its stored form carries the stream header
    if (first && blocksPerChunk != 0xFFFFFFFFu) {                // ... and, in brotli-mt framing, the chunk's frame header goes in front
        if (t == 0) {
            // chunk extent: up to the next chunk's first block (or the end of the stream)
            const uint32_t nb = b + blocksPerChunk < nBlocks ? b + blocksPerChunk : nBlocks;
            const uint64_t end = nb < nBlocks ? plan[nb].off : result[0];
            const uint32_t csize = (uint32_t)(end - plan[b].off - 16u);
            const uint64_t usize = (nb < nBlocks ? (uint64_t)nb * GC_ZSTD_BLOCK_MAX : srcSize) - blockBase;
            const uint32_t hint = (uint32_t)(usize >> 16) + 1u;
            const uint32_t h[4] = { 0x184D2A50u, 8u, csize, 0x5242u | (hint << 16) };
            for (int i = 0; i < 16; i++) o[i] = (uint8_t)(h[i >> 2] >> (8 * (i & 3)));
        }
        o += 16;
    }
    if (!bi.stored) {
        const uint8_t* s = stage + (uint64_t)b * GC_BR_STAGE_STRIDE;
        for (uint32_t i = t; i < bi.size; i += 256u) o[i] = s[i];
    } else {
        // stored meta-block: [WBITS] ISLAST=0, MNIBBLES, MLEN-1, ISUNCOMPRESSED=1, zero padding, raw bytes
        uint32_t hb;
        {
            uint64_t acc = 0; uint32_t nb = 0;
            if (first) { acc |= (uint64_t)GC_BR_WBITS_CODE; nb = 4; }
            nb += 1;                                                       // ISLAST = 0
            const uint32_t mlen1 = blockLen - 1u, nib = mlen1 < (1u << 16) ? 4u : (mlen1 < (1u << 20) ? 5u : 6u);
            acc |= (uint64_t)(nib - 4u) << nb; nb += 2;
            acc |= (uint64_t)mlen1 << nb; nb += nib * 4u;
            acc |= 1ull << nb; nb += 1;                                    // ISUNCOMPRESSED
            hb = (nb + 7u) >> 3;
            if (t == 0) for (uint32_t i = 0; i < hb; i++) o[i] = (uint8_t)(acc >> (8u * i));
        }
        const uint8_t* s = src + blockBase;
        for (uint32_t i = t; i < blockLen; i += 256u) o[hb + i] = s[i];
        if (bi.lastInChunk && t == 0) o[hb + blockLen] = 0x03;            // ISLAST = 1, ISLASTEMPTY = 1
    }
}
