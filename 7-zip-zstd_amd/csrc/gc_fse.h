// gc_fse.h -- device-side FSE table toolkit (single-lane helpers, small alphabets).
//
// Format-normative parts restated from the reference (must match the decoder bit for bit):
//   fse_write_ncount   <- FSE_writeNCount_generic   C/zstd/fse_compress.c:234-328 (header bit layout)
//   fse_build_ctable   <- FSE_buildCTable_wksp      C/zstd/fse_compress.c:68-200  (spread, state table, symbolTT)
//   fse state step     <- FSE_initCState2 / FSE_encodeSymbol   C/zstd/fse.h:443-461
// Free choice (any distribution summing to 2^tableLog is decodable; reference heuristic = FSE_normalizeCount,
// fse_compress.c:465): fse_normalize below is a plain largest-remainder scaling.
#pragma once
#include "gc_common.h"
#include "gc_device.h"

typedef GcSeqSymG GcFseSym;       // { int32_t deltaFindState; uint32_t deltaNbBits; } -- also the element type of the tables that travel between K3's kernels

// log2(x) in 1/256 bit units (piecewise linear; only used to compare table costs)
__device__ __forceinline__ uint32_t gc_log2_q8(uint32_t x)
{
    uint32_t e = gc_hibit32(x);
    uint32_t frac = ((x << (31u - e)) >> 23) & 0xFFu;
    return (e << 8) + frac;
}

// Scale count[0..maxSym] (sum = total > 0) to norm[] summing to 1<<L, every present symbol >= 1.
__device__ inline void gc_fse_normalize(const uint32_t* count, uint32_t maxSym, uint32_t total, uint32_t L, int16_t* norm)
{
    const uint32_t size = 1u << L;
    uint32_t sum = 0, best = 0, bestCount = 0;
    for (uint32_t s = 0; s <= maxSym; s++) {
        uint32_t c = count[s], v = 0;
        if (c) {
            v = (uint32_t)(((uint64_t)c * size + (total >> 1)) / total);
            if (v == 0) v = 1;
            if (c > bestCount) { bestCount = c; best = s; }
        }
        norm[s] = (int16_t)v; sum += v;
    }
    if (sum <= size) { norm[best] = (int16_t)(norm[best] + (int)(size - sum)); return; }
    uint32_t excess = sum - size;
    if ((uint32_t)norm[best] > excess + (uint32_t)(norm[best] >> 2)) { norm[best] = (int16_t)(norm[best] - (int)excess); return; }
    while (excess) {     // rare: take one cell at a time from the currently largest entry
        uint32_t m = 0; int mv = 0;
        for (uint32_t s = 0; s <= maxSym; s++) if (norm[s] > mv) { mv = norm[s]; m = s; }
        norm[m]--; excess--;
    }
}

// little serial LSB-first bit writer over a byte buffer
struct GcBitW { uint8_t* out; uint64_t acc; uint32_t nbits; uint32_t bytes; };
__device__ __forceinline__ void gc_bw_init(GcBitW& w, uint8_t* out) { w.out = out; w.acc = 0; w.nbits = 0; w.bytes = 0; }
__device__ __forceinline__ void gc_bw_add(GcBitW& w, uint32_t v, uint32_t nb)
{
    w.acc |= (uint64_t)(v & ((nb >= 32u) ? 0xFFFFFFFFu : ((1u << nb) - 1u))) << w.nbits;
    w.nbits += nb;
    while (w.nbits >= 8u) { w.out[w.bytes++] = (uint8_t)w.acc; w.acc >>= 8; w.nbits -= 8u; }
}
__device__ __forceinline__ uint32_t gc_bw_finish(GcBitW& w)   // pads the last byte with zeros
{
    if (w.nbits) { w.out[w.bytes++] = (uint8_t)w.acc; w.acc = 0; w.nbits = 0; }
    return w.bytes;
}

// NCount header (fse_compress.c:234-328).  Returns the number of bytes written.
__device__ inline uint32_t gc_fse_write_ncount(uint8_t* out, const int16_t* norm, uint32_t maxSym, uint32_t L)
{
    GcBitW w; gc_bw_init(w, out);
    gc_bw_add(w, L - 5u, 4);
    int remaining = 1 << L;
    uint32_t s = 0;
    while (remaining > 0 && s <= maxSym) {
        int count = norm[s];
        uint32_t val = (uint32_t)(count + 1);
        uint32_t bits = gc_hibit32((uint32_t)remaining + 1u) + 1u;
        uint32_t lower = (1u << (bits - 1u)) - 1u;
        uint32_t thr = (1u << bits) - 1u - ((uint32_t)remaining + 1u);
        if (val < thr) gc_bw_add(w, val, bits - 1u);
        else if (val <= lower) gc_bw_add(w, val, bits);
        else gc_bw_add(w, val + thr, bits);
        remaining -= count < 0 ? -count : count;
        s++;
        if (count == 0) {
            uint32_t z = 0;
            while (s + z <= maxSym && norm[s + z] == 0) z++;
            s += z;
            while (z >= 3u) { gc_bw_add(w, 3u, 2); z -= 3u; }
            gc_bw_add(w, z, 2);
        }
    }
    return gc_bw_finish(w);
}

// Compression table (fse_compress.c:68-200).  stateTable has 1<<L entries, tt maxSym+1 entries;
// tableSymbol (1<<L bytes) and cumul (maxSym+2 entries) are scratch.
__device__ inline void gc_fse_build_ctable(const int16_t* norm, uint32_t maxSym, uint32_t L, uint16_t* stateTable,
                                           GcFseSym* tt, uint8_t* tableSymbol, uint16_t* cumul)
{
    const uint32_t size = 1u << L, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    uint32_t high = size - 1u;
    cumul[0] = 0;
    for (uint32_t u = 1; u <= maxSym + 1u; u++) {
        if (norm[u - 1] == -1) { cumul[u] = (uint16_t)(cumul[u - 1] + 1u); tableSymbol[high--] = (uint8_t)(u - 1u); }
        else cumul[u] = (uint16_t)(cumul[u - 1] + (uint16_t)norm[u - 1]);
    }
    {
        uint32_t pos = 0;
        for (uint32_t s = 0; s <= maxSym; s++) {
            int freq = norm[s];
            for (int i = 0; i < freq; i++) {
                tableSymbol[pos] = (uint8_t)s;
                pos = (pos + step) & mask;
                while (pos > high) pos = (pos + step) & mask;
            }
        }
    }
    for (uint32_t u = 0; u < size; u++) { uint32_t s = tableSymbol[u]; stateTable[cumul[s]++] = (uint16_t)(size + u); }
    {
        uint32_t total = 0;
        for (uint32_t s = 0; s <= maxSym; s++) {
            int n = norm[s];
            if (n == 0) { tt[s].deltaNbBits = ((L + 1u) << 16) - size; tt[s].deltaFindState = 0; }
            else if (n == -1 || n == 1) { tt[s].deltaNbBits = (L << 16) - size; tt[s].deltaFindState = (int32_t)total - 1; total++; }
            else {
                uint32_t maxBitsOut = L - gc_hibit32((uint32_t)n - 1u);
                uint32_t minStatePlus = (uint32_t)n << maxBitsOut;
                tt[s].deltaNbBits = (maxBitsOut << 16) - minStatePlus;
                tt[s].deltaFindState = (int32_t)total - n;
                total += (uint32_t)n;
            }
        }
    }
}

// first symbol of a state (fse.h:443): no bits are produced
__device__ __forceinline__ uint32_t gc_fse_init_state(const uint16_t* stateTable, GcFseSym sym)
{
    uint32_t nbBitsOut = (sym.deltaNbBits + (1u << 15)) >> 16;
    uint32_t value = (nbBitsOut << 16) - sym.deltaNbBits;
    return stateTable[(value >> nbBitsOut) + (uint32_t)sym.deltaFindState];
}
