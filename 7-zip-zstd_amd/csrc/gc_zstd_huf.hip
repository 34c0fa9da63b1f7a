// gc_zstd_huf.hip -- K2: literals section of one zstd block per workgroup (256 threads).
//
// Replaces ZSTD_compressLiterals (C/zstd/zstd_compress_literals.c:129-235) and the huff0 encoder behind it:
// HIST_count (hist.c:76), HUF_buildCTable_wksp (huf_compress.c:756), HUF_writeCTable_wksp (:248),
// HUF_compressWeights (:138), HUF_compress4X_usingCTable_internal (:1168), HUF_compress1X body (:1056).
//
// Format-normative (must match the decoder): weight = maxLen+1-nbBits with the last symbol implicit; code
// values assigned longest-length-first in ascending symbol order (huf_compress.c:730-753); header byte <128 =
// FSE-compressed weights, >=128 = 127+n raw nibbles (:273-289); streams written last-symbol-first, LSB-first,
// closed by a 1 bit (bitstream.h:236); 4 streams of ceil(n/4) symbols behind a 6-byte jump table (:1168-1215);
// section header layouts (zstd_compress_literals.c:209-232).  Free: the code lengths themselves (any complete
// prefix code with lengths <= 11) and the raw/RLE/compressed decision.
//
// Parallel structure: histogram = LDS atomics on per-wave copies; symbol sort = rank-by-counting (one thread per
// symbol); tree construction and the tiny weight header are serial in lane 0 (<= 255 merges); the encode is
// fully parallel: per-literal bit lengths -> block prefix sums give every symbol its absolute bit offset, bits
// are OR-ed into an LDS tile (ds_or_b32) and whole bytes stream out; a <8-bit carry links consecutive tiles.
#include "gc_common.h"
#include "gc_device.h"
#include "gc_fse.h"

#define HUF_T        256u
#define HUF_MAXBITS  11u
#define HUF_V        8u                         // literals per thread per tile
#define HUF_TILE_SYMS (HUF_T * HUF_V)
#define HUF_TILE_WORDS ((HUF_TILE_SYMS * HUF_MAXBITS) / 32u + 8u)

// block-wide exclusive scan of one uint32 per thread; returns exclusive prefix, *total = block sum
__device__ __forceinline__ uint32_t huf_block_excl_scan(uint32_t v, uint32_t* sWave, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t incl = gc_wave_incl_sum(v);
    if (lane == 63u) sWave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (uint32_t w = 0; w < HUF_T / 64u; w++) { uint32_t c = sWave[w]; if (w < wave) before += c; all += c; }
    __syncthreads();
    *total = all;
    return before + incl - v;
}

// OR `nbits` (<= 44) bits of v into the LDS bit buffer at bit offset bitoff
__device__ __forceinline__ void huf_or_bits(uint32_t* buf, uint32_t bitoff, uint64_t v, uint32_t nbits)
{
    if (nbits == 0) return;
    uint32_t word = bitoff >> 5, sh = bitoff & 31u;
    uint32_t w0 = (uint32_t)(v << sh);
    uint64_t rest = sh ? (v >> (32u - sh)) : (v >> 32);
    if (w0) atomicOr(&buf[word], w0);
    if ((uint32_t)rest) atomicOr(&buf[word + 1], (uint32_t)rest);
    if ((uint32_t)(rest >> 32)) atomicOr(&buf[word + 2], (uint32_t)(rest >> 32));
}

extern "C" __global__ void __launch_bounds__(HUF_T)
gc_zstd_huf_kernel(const uint8_t* __restrict__ lit, const GcBlockMeta* __restrict__ meta,
                   uint8_t* __restrict__ litSec, GcSectionInfo* __restrict__ info)
{
    __shared__ uint32_t sHist[4][256];
    __shared__ uint32_t sCnt[256];
    __shared__ uint8_t  sNb[256];
    __shared__ uint16_t sCode[256];
    __shared__ uint16_t sSorted[256];
    __shared__ uint32_t sNodeW[512];
    __shared__ uint16_t sParent[512];
    __shared__ uint8_t  sDepth[512];
    __shared__ uint8_t  sHdr[192];
    __shared__ uint32_t sWave[8];
    __shared__ uint32_t sRed[4][4];
    __shared__ uint32_t sTile[HUF_TILE_WORDS];
    __shared__ uint32_t sMisc[8];     // 0 hdrSize, 1 maxLen, 2 ok flag
    // scratch of the FSE weight coder
    __shared__ int16_t  sWNorm[16];
    __shared__ uint32_t sWCount[16];
    __shared__ uint16_t sWState[64];
    __shared__ GcFseSym sWTT[16];
    __shared__ uint8_t  sWSpread[64];
    __shared__ uint16_t sWCumul[18];

    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, b = blockIdx.x;
    const uint32_t nlit = meta[b].nLit;
    const uint8_t* L = lit + (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    uint8_t* out = litSec + (uint64_t)b * GC_LITSEC_STRIDE;

    if (nlit == 0) { if (t == 0) { out[0] = 0; info[b].litSecSize = 1; info[b].flags = 1; } return; }

    // ---- histogram
    for (uint32_t i = t; i < 4u * 256u; i += HUF_T) (&sHist[0][0])[i] = 0;
    __syncthreads();
    for (uint32_t i = t * 4u; i < nlit; i += HUF_T * 4u) {
        if (i + 4u <= nlit) {
            uint32_t v = *(const uint32_t*)(L + i);
            atomicAdd(&sHist[wave][v & 0xFFu], 1u); atomicAdd(&sHist[wave][(v >> 8) & 0xFFu], 1u);
            atomicAdd(&sHist[wave][(v >> 16) & 0xFFu], 1u); atomicAdd(&sHist[wave][v >> 24], 1u);
        } else for (uint32_t j = i; j < nlit; j++) atomicAdd(&sHist[wave][L[j]], 1u);
    }
    __syncthreads();
    const uint32_t myCnt = sHist[0][t] + sHist[1][t] + sHist[2][t] + sHist[3][t];
    sCnt[t] = myCnt;
    uint32_t maxCount = gc_wave_max(myCnt);
    uint32_t nsym = gc_wave_sum(myCnt ? 1u : 0u);
    uint32_t maxSym = gc_wave_max(myCnt ? t : 0u);
    if (lane == 0) { sRed[wave][0] = maxCount; sRed[wave][1] = nsym; sRed[wave][2] = maxSym; }
    __syncthreads();
    maxCount = max(max(sRed[0][0], sRed[1][0]), max(sRed[2][0], sRed[3][0]));
    nsym = sRed[0][1] + sRed[1][1] + sRed[2][1] + sRed[3][1];
    maxSym = max(max(sRed[0][2], sRed[1][2]), max(sRed[2][2], sRed[3][2]));

    // ---- mode decision (zstd_compress_literals.c:142-172, huf_compress.c:1384)
    int mode = 2;                                     // 0 raw, 1 rle, 2 huffman
    if (nlit < 64u) mode = 0;
    else if (maxCount == nlit) mode = 1;
    else if (maxCount <= (nlit >> 7) + 4u) mode = 0;

    uint32_t bitsK[4] = { 0, 0, 0, 0 };
    const bool single = nlit < 256u;
    const uint32_t seg = single ? nlit : (nlit + 3u) / 4u;
    uint32_t hdrSize = 0, cSize = 0;

    if (mode == 2) {
        // ---- sort present symbols by (count, symbol): rank by counting
        if (myCnt) {
            uint32_t rank = 0;
            for (uint32_t s = 0; s < 256u; s++) { uint32_t c = sCnt[s]; rank += (c && (c < myCnt || (c == myCnt && s < t))) ? 1u : 0u; }
            sSorted[rank] = (uint16_t)t;
        }
        sNb[t] = 0;
        __syncthreads();
        if (t == 0) {
            // ---- Huffman tree by two-queue merge over leaves sorted ascending
            for (uint32_t i = 0; i < nsym; i++) sNodeW[i] = sCnt[sSorted[i]];
            uint32_t li = 0, ii = nsym, ni = nsym;
            for (uint32_t k = 0; k + 1u < nsym; k++) {
                uint32_t a, c;
                if (li < nsym && (ii >= ni || sNodeW[li] <= sNodeW[ii])) a = li++; else a = ii++;
                if (li < nsym && (ii >= ni || sNodeW[li] <= sNodeW[ii])) c = li++; else c = ii++;
                sNodeW[ni] = sNodeW[a] + sNodeW[c]; sParent[a] = (uint16_t)ni; sParent[c] = (uint16_t)ni; ni++;
            }
            const uint32_t root = 2u * nsym - 2u;
            sDepth[root] = 0;
            for (int i = (int)root - 1; i >= 0; i--) { uint32_t d = sDepth[sParent[i]] + 1u; sDepth[i] = (uint8_t)(d > 255u ? 255u : d); }
            uint32_t maxLen = 0;
            for (uint32_t i = 0; i < nsym; i++) maxLen = max(maxLen, (uint32_t)sDepth[i]);
            if (maxLen > HUF_MAXBITS) {
                // length limiting: clamp, then repay the Kraft debt from the least frequent symbols upward
                int debt = 0;
                for (uint32_t i = 0; i < nsym; i++) { if (sDepth[i] > HUF_MAXBITS) sDepth[i] = HUF_MAXBITS; debt += 1 << (HUF_MAXBITS - sDepth[i]); }
                debt -= 1 << HUF_MAXBITS;
                for (int bl = (int)HUF_MAXBITS - 1; bl >= 1 && debt > 0; bl--) {
                    const int r = 1 << (HUF_MAXBITS - 1 - bl);
                    for (uint32_t i = 0; i < nsym && debt > 0; i++)
                        if (sDepth[i] == bl) { sDepth[i] = (uint8_t)(bl + 1); debt -= r; }
                }
                for (int i = (int)nsym - 1; i >= 0 && debt < 0; i--)          // overshoot: shorten frequent 11-bit codes
                    if (sDepth[i] == HUF_MAXBITS) { sDepth[i] = HUF_MAXBITS - 1; debt += 1; }
                maxLen = HUF_MAXBITS;
                sMisc[2] = (debt == 0) ? 1u : 0u;
            } else sMisc[2] = 1u;
            for (uint32_t i = 0; i < nsym; i++) sNb[sSorted[i]] = sDepth[i];
            // ---- canonical code values (huf_compress.c:730-753): longest codes first, ascending symbol order
            uint32_t nbPerW[HUF_MAXBITS + 2], nextCode[HUF_MAXBITS + 2];
            for (uint32_t w = 0; w <= HUF_MAXBITS + 1u; w++) nbPerW[w] = 0;
            for (uint32_t s = 0; s <= maxSym; s++) if (sNb[s]) nbPerW[maxLen + 1u - sNb[s]]++;
            { uint32_t start = 0; for (uint32_t w = 1; w <= maxLen; w++) { nextCode[w] = start >> (w - 1u); start += nbPerW[w] << (w - 1u); } }
            for (uint32_t s = 0; s <= maxSym; s++) if (sNb[s]) sCode[s] = (uint16_t)(nextCode[maxLen + 1u - sNb[s]]++);
            sMisc[1] = maxLen;
            // ---- tree description (huf_compress.c:248-295)
            const uint32_t nW = maxSym;          // weights of symbols 0..maxSym-1; the last one is implied
            uint32_t best = 0xFFFFFFFFu;
            // (a) FSE-compressed weights (HUF_compressWeights, huf_compress.c:138-190)
            if (nW > 2u) {
                uint32_t wMax = 0, cMax = 0;
                for (uint32_t w = 0; w < 16u; w++) sWCount[w] = 0;
                for (uint32_t s = 0; s < nW; s++) { uint32_t w = sNb[s] ? maxLen + 1u - sNb[s] : 0u; sWCount[w]++; wMax = max(wMax, w); }
                for (uint32_t w = 0; w <= wMax; w++) cMax = max(cMax, sWCount[w]);
                if (cMax != nW && cMax != 1u) {
                    uint32_t tl = 6u, maxBitsSrc = gc_hibit32(nW - 1u) - 2u;
                    uint32_t minBits = min(gc_hibit32(nW) + 1u, gc_hibit32(wMax) + 2u);
                    if (maxBitsSrc < tl) tl = maxBitsSrc;
                    if (minBits > tl) tl = minBits;
                    if (tl < 5u) tl = 5u;
                    if (tl > 6u) tl = 6u;
                    gc_fse_normalize(sWCount, wMax, nW, tl, sWNorm);
                    uint32_t hb = gc_fse_write_ncount(sHdr + 1, sWNorm, wMax, tl);
                    gc_fse_build_ctable(sWNorm, wMax, tl, sWState, sWTT, sWSpread, sWCumul);
                    // two interleaved states, last weight first (fse_compress.c:551-611): even index -> state1
                    GcBitW bw; gc_bw_init(bw, sHdr + 1 + hb);
                    uint32_t st1 = 0, st2 = 0; bool i1 = false, i2 = false;
                    for (int i = (int)nW - 1; i >= 0; i--) {
                        uint32_t w = sNb[i] ? maxLen + 1u - sNb[i] : 0u;
                        GcFseSym sy = sWTT[w];
                        uint32_t& st = (i & 1) ? st2 : st1; bool& inited = (i & 1) ? i2 : i1;
                        if (!inited) { st = gc_fse_init_state(sWState, sy); inited = true; }
                        else { uint32_t nb = (st + sy.deltaNbBits) >> 16; gc_bw_add(bw, st, nb); st = sWState[(st >> nb) + (uint32_t)sy.deltaFindState]; }
                    }
                    gc_bw_add(bw, st2, tl); gc_bw_add(bw, st1, tl); gc_bw_add(bw, 1u, 1);
                    uint32_t sb = gc_bw_finish(bw);
                    if (hb + sb < 128u) { sHdr[0] = (uint8_t)(hb + sb); best = 1u + hb + sb; }
                }
            }
            // (b) raw 4-bit weights, only describable for <= 128 weights
            if (nW <= 128u && 1u + (nW + 1u) / 2u < best) {
                sHdr[0] = (uint8_t)(127u + nW);
                for (uint32_t s = 0; s < nW; s += 2u) {
                    uint32_t w0 = sNb[s] ? maxLen + 1u - sNb[s] : 0u;
                    uint32_t w1 = (s + 1u < nW && sNb[s + 1u]) ? maxLen + 1u - sNb[s + 1u] : 0u;
                    sHdr[1u + s / 2u] = (uint8_t)((w0 << 4) | w1);
                }
                best = 1u + (nW + 1u) / 2u;
            }
            sMisc[0] = best;
            if (best == 0xFFFFFFFFu) sMisc[2] = 0u;
        }
        __syncthreads();
        hdrSize = sMisc[0];
        if (sMisc[2] == 0u) mode = 0;
        else {
            // ---- per-stream bit totals
            uint32_t acc[4] = { 0, 0, 0, 0 };
            for (uint32_t i = t; i < nlit; i += HUF_T) { uint32_t k = single ? 0u : i / seg; acc[k] += sNb[L[i]]; }
            for (int k = 0; k < 4; k++) { uint32_t v = gc_wave_sum(acc[k]); if (lane == 0) sRed[wave][k] = v; }
            __syncthreads();
            for (int k = 0; k < 4; k++) bitsK[k] = sRed[0][k] + sRed[1][k] + sRed[2][k] + sRed[3][k];
            cSize = hdrSize + (single ? (bitsK[0] >> 3) + 1u
                                      : 6u + (bitsK[0] >> 3) + (bitsK[1] >> 3) + (bitsK[2] >> 3) + (bitsK[3] >> 3) + 4u);
            const uint32_t minGain = (nlit >> 6) + 2u;
            if (cSize + minGain >= nlit) mode = 0;
        }
    }

    if (mode != 2) {
        // ---- raw / RLE literals (zstd_compress_literals.c:40-127)
        const uint32_t hs = nlit < 32u ? 1u : (nlit < 4096u ? 2u : 3u);
        if (t == 0) {
            uint32_t ty = (uint32_t)mode;
            if (hs == 1u) out[0] = (uint8_t)(ty | (nlit << 3));
            else if (hs == 2u) { uint32_t v = ty | (1u << 2) | (nlit << 4); out[0] = (uint8_t)v; out[1] = (uint8_t)(v >> 8); }
            else { uint32_t v = ty | (3u << 2) | (nlit << 4); out[0] = (uint8_t)v; out[1] = (uint8_t)(v >> 8); out[2] = (uint8_t)(v >> 16); }
            if (mode == 1) out[hs] = L[0];
            info[b].litSecSize = hs + (mode == 1 ? 1u : nlit);
            info[b].flags = mode == 1 ? 2u : 1u;
        }
        if (mode == 0) for (uint32_t i = t; i < nlit; i += HUF_T) out[hs + i] = L[i];
        return;
    }

    // ---- compressed literals: section header, tree description, jump table
    const uint32_t lh = 3u + (nlit >= 1024u ? 1u : 0u) + (nlit >= 16384u ? 1u : 0u);
    if (t == 0) {
        if (lh == 3u) { uint32_t v = 2u | ((single ? 0u : 1u) << 2) | (nlit << 4) | (cSize << 14); out[0] = (uint8_t)v; out[1] = (uint8_t)(v >> 8); out[2] = (uint8_t)(v >> 16); }
        else if (lh == 4u) { uint32_t v = 2u | (2u << 2) | (nlit << 4) | (cSize << 18); out[0] = (uint8_t)v; out[1] = (uint8_t)(v >> 8); out[2] = (uint8_t)(v >> 16); out[3] = (uint8_t)(v >> 24); }
        else { uint64_t v = 2u | (3u << 2) | ((uint64_t)nlit << 4) | ((uint64_t)cSize << 22); for (int i = 0; i < 5; i++) out[i] = (uint8_t)(v >> (8 * i)); }
        info[b].litSecSize = lh + cSize;
        info[b].flags = 0;
        if (!single) {
            uint8_t* jt = out + lh + hdrSize;
            for (int k = 0; k < 3; k++) { uint32_t sz = (bitsK[k] >> 3) + 1u; jt[2 * k] = (uint8_t)sz; jt[2 * k + 1] = (uint8_t)(sz >> 8); }
        }
    }
    for (uint32_t i = t; i < hdrSize; i += HUF_T) out[lh + i] = sHdr[i];

    // ---- encode the streams
    uint32_t streamPos = lh + hdrSize + (single ? 0u : 6u);
    const uint32_t nStreams = single ? 1u : 4u;
    for (uint32_t k = 0; k < nStreams; k++) {
        const uint32_t s0 = k * seg, s1 = (k + 1u == nStreams) ? nlit : (k + 1u) * seg, len = s1 - s0;
        uint8_t* so = out + streamPos;
        uint32_t carryBits = 0, carryVal = 0, outBytes = 0;
        for (uint32_t tb = 0; tb < len; tb += HUF_TILE_SYMS) {
            for (uint32_t i = t; i < HUF_TILE_WORDS; i += HUF_T) sTile[i] = 0;
            __syncthreads();
            // reversed order: r = tb + t*V + j  <->  literal index s1-1-r
            uint64_t a0 = 0, a1 = 0; uint32_t n0 = 0, n1 = 0;
            const uint32_t r0 = tb + t * HUF_V;
            for (uint32_t j = 0; j < HUF_V; j++) {
                uint32_t r = r0 + j;
                if (r < len) {
                    uint32_t sym = L[s1 - 1u - r];
                    uint32_t nb = sNb[sym]; uint64_t c = sCode[sym];
                    if (j < 4u) { a0 |= c << n0; n0 += nb; } else { a1 |= c << n1; n1 += nb; }
                }
            }
            uint32_t tileBits;
            uint32_t off = huf_block_excl_scan(n0 + n1, sWave, &tileBits) + carryBits;
            if (t == 0 && carryBits) atomicOr(&sTile[0], carryVal);
            huf_or_bits(sTile, off, a0, n0);
            huf_or_bits(sTile, off + n0, a1, n1);
            uint32_t endBits = carryBits + tileBits;
            const bool lastTile = tb + HUF_TILE_SYMS >= len;
            if (lastTile) { if (t == 0) atomicOr(&sTile[endBits >> 5], 1u << (endBits & 31u)); endBits += 1u; }
            __syncthreads();
            const uint32_t flush = lastTile ? (endBits + 7u) >> 3 : endBits >> 3;
            for (uint32_t i = t; i < flush; i += HUF_T) so[outBytes + i] = (uint8_t)(sTile[i >> 2] >> ((i & 3u) * 8u));
            carryBits = endBits & 7u;
            carryVal = lastTile ? 0u : ((sTile[flush >> 2] >> ((flush & 3u) * 8u)) & 0xFFu);
            outBytes += flush;
            __syncthreads();
        }
        streamPos += (bitsK[k] >> 3) + 1u;
    }
}
