// gc_brotli_dec.hip -- BROTLI decoder (7-Zip method id 0x4F71102): the decoding half of what NCompress::NBROTLI::CDecoder
// (CPP/7zip/Compress/BrotliDecoder.cpp:124) runs through BROTLIMT_decompressDCtx (C/zstdmt/brotli-mt_decompress.c:191-288: 16-byte frame headers
// 0x184D2A50, 8, compressed size, 0x5242, hint in 64 KiB units -- each followed by a complete RFC 7932 stream) and BrotliDecoderDecompressStream
// (C/brotli/br_decode.c).
//
// Brotli's literal trees are chosen by the bytes just produced and its copies reach back over everything the stream has produced, so INSIDE a stream the
// entropy decoder cannot run ahead of the output the way the zstd decoder's stages do.  What is independent are the brotli-mt chunks:
// config C5 is 1 590 of them.  One WAVE per chunk: lane 0 runs the stream's state machine (bit reader, prefix codes, context maps, block switches, the
// insert-and-copy commands), all 64 lanes move the bytes of a copy that is longer than a few bytes and does not overlap itself.  Every chunk decodes into a
// slot of its hint size; a scan over the chunks' real sizes and a copy kernel pack the content (brotli-mt's hint is an upper bound, not the size).
//
// Prefix codes are kept CANONICAL (RFC 7932 section 3.2): per code the count, the first code and the first symbol index of every length (96 bytes) + the symbols in
// code order, in an LDS arena of the wave; a meta-block of the reference's quality 10-11 can hold up to 256 literal and 256 distance trees: what does not fit LDS
// goes to a page in HBM (same code, flat addresses).  A symbol is decoded by the WAVE: lane l tests whether the next l bits are a code of length l (canonical
// codes: exactly one length answers), a ballot names the length -- no tables to build, which is what a meta-block header with 13-256 trees would spend its time on.
// The stream's state machine (bit reader, block switches, commands, distance ring) runs in all lanes alike; headers (prefix codes, context maps) are read by lane
// 0; copies are done by the wave (lane i writes byte i: a copy that overlaps itself reads `i mod distance`).
//
// The static dictionary of RFC 7932 Appendix A (122 784 bytes, CRC-32 0x5136cb04 as the RFC states it) is NORMATIVE DATA that this repository does not hold: the
// host hands it over once (gc_brotli_dec_set_dictionary: the plugin built inside the reference tree passes BrotliGetDictionary()->data of the host's own
// C/brotli/br_dictionary.c); without it a stream that refers to the dictionary is answered with GC_ERR_UNSUPPORTED -- this engine's own
// streams never do.  The 121 transforms of Appendix B are in gc_brotli_transforms.h.
#include "gpucodec.h"
#include "gc_common.h"
#include "gc_device.h"
#include "gc_brotli.h"
#include "gc_brotli_dec.h"
#ifdef HIPEMU
#include "hip_runtime_stub.h"
#include <stdio.h>
#else
#include <hip/hip_runtime.h>
#define GC_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif
#include <stdlib.h>
#include <string.h>
#include "gc_host_stream.h"
#include "gc_brotli_transforms.h"

// status of one chunk
#define BRD_OK          0u
#define BRD_CORRUPT     1u
#define BRD_DST_SMALL   2u
#define BRD_DICTIONARY  3u            // the stream refers to the static dictionary and none is loaded
#define BRD_LIMIT       4u            // more trees / block types than the arenas hold

#define BRD_PAGE        (800u * 1024u)      // what a chunk takes in HBM when LDS does not hold its meta-block (256 literal + 256 command + 256 distance trees at their largest: 0.77 MB)
#define BRD_MAX_WAVES   1792u         // 7 waves per CU (LDS) x 256 CUs: the launch's width; chunks beyond it are taken in turns

struct GcBrDecChunk { uint64_t srcOff; uint64_t stageOff; uint32_t srcSize; uint32_t hintBytes; };      // payload of one brotli-mt frame (or a whole plain stream)
struct GcBrDecResult { uint32_t size; uint32_t status; };
struct GcBrDict { const uint8_t* words; };                                                               // RFC 7932 Appendix A, or null

// ---- format tables (RFC 7932): insert / copy length codes (section 5), block counts (section 6), code length code order (section 3.5), dictionary buckets (section 8)
__constant__ uint16_t kdInsBase[24] = { 0,1,2,3,4,5,6,8,10,14,18,26,34,50,66,98,130,194,322,578,1090,2114,6210,22594 };
__constant__ uint8_t  kdInsExtra[24] = { 0,0,0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,7,8,9,10,12,14,24 };
__constant__ uint16_t kdCopyBase[24] = { 2,3,4,5,6,7,8,9,10,12,14,18,22,30,38,54,70,102,134,198,326,582,1094,2118 };
__constant__ uint8_t  kdCopyExtra[24] = { 0,0,0,0,0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,7,8,9,10,24 };
__constant__ uint8_t  kdCellIns[11] = { 0, 0, 0, 0, 8, 8, 0, 16, 8, 16, 16 };
__constant__ uint8_t  kdCellCopy[11] = { 0, 8, 0, 8, 0, 8, 16, 0, 16, 8, 16 };
__constant__ uint16_t kdBlockBase[26] = { 1,5,9,13,17,25,33,41,49,65,81,97,113,145,177,209,241,305,369,497,753,1265,2289,4337,8433,16625 };
__constant__ uint8_t  kdBlockExtra[26] = { 2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,6,6,7,8,9,10,11,12,13,24 };
__constant__ uint8_t  kdClcOrder[18] = { 1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15 };
__constant__ uint8_t  kdDictBits[25] = { 0,0,0,0,10,10,11,11,10,10,10,10,10,9,9,8,7,7,8,7,7,6,6,5,5 };
__constant__ uint32_t kdDictOff[25] = { 0,0,0,0,0,4096,9216,21504,35840,44032,53248,63488,74752,87040,93696,100864,104704,106752,108928,113536,115968,118528,119872,121280,122016 };

struct BrdQuad { uint32_t x, y, z, w; };
// LSB-first bit reader over the chunk's bytes; identical in every lane.  The stream arrives 16 bytes at a time: `q` (four words in scalar registers) feeds the accumulator, `pend`
// (the 16 bytes behind them, a load in flight in vector registers) becomes q when q is empty and the next load is issued then -- four refills ahead, which covers a trip to HBM;
// a word that is looked at the moment it is loaded (one refill ahead, `readfirstlane` on the spot) makes every refill wait for its load.
struct BrdBits {
    const uint8_t* p; uint64_t acc; uint32_t n; uint32_t pos, end; uint32_t over;      // pos: the next byte that enters acc
    uint64_t qlo, qhi; uint32_t qn;                                                     // qn words at pos, pos + 4, ...
    BrdQuad pend; uint32_t pendOk;                                                      // the 16 bytes at pos + 4 * qn
};
__device__ __forceinline__ BrdQuad brd_load128(const uint8_t* p) { BrdQuad v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void brd_next_quad(BrdBits& b)          // q is empty: pend -> q, the load behind it goes out
{
    if (!b.pendOk) return;
    b.qlo = (uint64_t)gc_uniform(b.pend.x) | ((uint64_t)gc_uniform(b.pend.y) << 32); b.qhi = (uint64_t)gc_uniform(b.pend.z) | ((uint64_t)gc_uniform(b.pend.w) << 32); b.qn = 4u;
    b.pendOk = b.pos + 32u <= b.end;
    if (b.pendOk) b.pend = brd_load128(b.p + b.pos + 16u);
}
__device__ __forceinline__ void brd_prime(BrdBits& b)              // after pos has been set
{
    b.qn = 0u; b.qlo = 0; b.qhi = 0; b.pend.x = b.pend.y = b.pend.z = b.pend.w = 0u;
    b.pendOk = b.pos + 16u <= b.end;
    if (b.pendOk) { b.pend = brd_load128(b.p + b.pos); brd_next_quad(b); }
}
__device__ __forceinline__ void brd_fill(BrdBits& b)               // afterwards n >= 33
{
    if (b.qn) {
        b.acc |= (uint64_t)(uint32_t)b.qlo << b.n; b.n += 32u; b.pos += 4u;
        b.qlo = (b.qlo >> 32) | (b.qhi << 32); b.qhi >>= 32; b.qn--;
        if (b.qn == 0u) brd_next_quad(b);
        return;
    }
    while (b.n <= 56u) { const uint64_t v = b.pos < b.end ? gc_uniform(b.p[b.pos]) : 0u; if (b.pos >= b.end + 16u) b.over = 1u; b.acc |= v << b.n; b.n += 8u; b.pos++; }
}
__device__ __forceinline__ uint32_t brd_take(BrdBits& b, uint32_t k)      // k <= 32
{
    if (b.n <= 32u) brd_fill(b);
    const uint32_t v = (uint32_t)(b.acc & ((1ull << k) - 1ull));
    b.acc >>= k; b.n -= k;
    return v;
}
__device__ __forceinline__ uint32_t brd_consumed(const BrdBits& b) { return b.pos - (b.n >> 3); }      // bytes whose bits have been taken (rounded up)

// Where a meta-block's tables live: the wave's LDS arena, and a page of HBM behind it for meta-blocks with hundreds of codes.  A HANDLE is a byte offset with
// bit 31 = "in the page" and bit 30 = "symbols are 16 bits wide" (alphabets above 256).
// A canonical prefix code at a handle: uint16 t[48] -- t[0..15] codes per length (t[0] = 1: ONE symbol, coded in zero bits), t[16..31] the first code of the length,
// t[32..47] the index of its first symbol -- then the symbols in code order.
#define BRD_H_HBM   0x80000000u
#define BRD_H_WIDE  0x40000000u
#define BRD_H_OFF   0x3FFFFFFFu
struct BrdMem { uint8_t* lds; uint8_t* hbm; };
__device__ __forceinline__ const uint8_t* brd_at(const BrdMem& m, uint32_t h) { return (h & BRD_H_HBM) ? m.hbm + (h & BRD_H_OFF) : m.lds + (h & BRD_H_OFF); }
__device__ __forceinline__ uint32_t brd_t(const BrdMem& m, uint32_t h, uint32_t i)
{
    return (h & BRD_H_HBM) ? ((const uint16_t*)(m.hbm + (h & BRD_H_OFF)))[i] : ((const uint16_t*)(m.lds + (h & BRD_H_OFF)))[i];
}
__device__ __forceinline__ uint32_t brd_symbol(const BrdMem& m, uint32_t h, uint32_t i)
{
    const uint8_t* s = brd_at(m, h) + 96u;
    return (h & BRD_H_WIDE) ? ((const uint16_t*)s)[i] : s[i];
}
// one lane on its own (headers)
__device__ __forceinline__ uint32_t brd_sym(BrdBits& b, const BrdMem& m, uint32_t h)
{
    if (brd_t(m, h, 0)) return brd_symbol(m, h, 0);
    if (b.n <= 32u) brd_fill(b);
    const uint32_t rev = __brev((uint32_t)b.acc);
#pragma unroll 1
    for (uint32_t len = 1; len <= 15u; len++) {
        const uint32_t c = rev >> (32u - len), first = brd_t(m, h, 16u + len);
        if (c - first < brd_t(m, h, len)) { b.acc >>= len; b.n -= len; return brd_symbol(m, h, brd_t(m, h, 32u + len) + c - first); }
    }
    b.over = 1u;                                                  // not a code word: the code is incomplete (damaged)
    return 0;
}
// the wave together: lane l tests whether the next l bits are a code of length l (canonical codes: one length answers); every lane gets the symbol
__device__ __forceinline__ uint32_t brd_sym_w(BrdBits& b, const BrdMem& m, uint32_t h, uint32_t lane)
{
    h = gc_uniform(h);
    if (gc_uniform(brd_t(m, h, 0))) return gc_uniform(brd_symbol(m, h, 0));
    if (b.n <= 32u) brd_fill(b);
    const uint32_t rev = __brev((uint32_t)b.acc), l = lane & 15u;
    const uint32_t c = l ? rev >> (32u - l) : 0u;
    const bool hit = l != 0u && c - brd_t(m, h, 16u + l) < brd_t(m, h, l);
    const uint64_t mk = __ballot(hit) & 0xFFFEull;
    if (mk == 0ull) { b.over = 1u; return 0; }
    const uint32_t len = (uint32_t)__ffsll((long long)mk) - 1u;
    const uint32_t i = gc_uniform(brd_t(m, h, 32u + len) + (rev >> (32u - len)) - brd_t(m, h, 16u + len));
    b.acc >>= len; b.n -= len;
    return gc_uniform(brd_symbol(m, h, i));
}
// ... through a table of 2^BITS entries in LDS (0x8000 | symbol << 4 | length; 0: the code is longer than BITS, the wave decodes it)
template <uint32_t BITS>
__device__ __forceinline__ uint32_t brd_sym_t(BrdBits& b, const uint16_t* tabs, uint32_t nTab, const uint32_t* dir, uint32_t tree, const BrdMem& m, uint32_t lane)
{
    if (tree < nTab) {
        if (b.n <= 32u) brd_fill(b);
        const uint32_t e = gc_uniform(tabs[(tree << BITS) + ((uint32_t)b.acc & ((1u << BITS) - 1u))]);
        if (e & 0x8000u) { const uint32_t len = e & 15u; b.acc >>= len; b.n -= len; return (e >> 4) & 0x7FFu; }
    }
    return brd_sym_w(b, m, dir[tree], lane);
}
// the wave fills the table of one code
__device__ __noinline__ void brd_table_w(const BrdMem m, uint32_t h, uint32_t bits, uint16_t* tab, uint32_t lane)
{
    const uint32_t entries = 1u << bits;
    if (brd_t(m, h, 0)) { const uint16_t e = (uint16_t)(0x8000u | (brd_symbol(m, h, 0) << 4)); for (uint32_t i = lane; i < entries; i += 64u) tab[i] = e; return; }
    for (uint32_t i = lane; i < entries; i += 64u) tab[i] = 0;
    gc_wave_sync();
    const uint32_t nShort = brd_t(m, h, 32u + bits) + brd_t(m, h, bits);      // the symbols with codes of at most `bits` bits come first in code order
    for (uint32_t i = lane; i < nShort; i += 64u) {
        uint32_t L = 1;
        while (L < bits && i >= brd_t(m, h, 32u + L) + brd_t(m, h, L)) L++;
        const uint32_t code = brd_t(m, h, 16u + L) + i - brd_t(m, h, 32u + L);
        const uint16_t e = (uint16_t)(0x8000u | (brd_symbol(m, h, i) << 4) | L);
        for (uint32_t k = __brev(code) >> (32u - L); k < entries; k += 1u << L) tab[k] = e;
    }
}

// lane 0's work space while it reads a meta-block's header, in LDS (private arrays indexed by data live in scratch memory, a trip to HBM per access)
struct BrdScratch { uint8_t len[704]; uint32_t cnt[16], off[16], cc[18]; uint8_t cl[18], csym[18], mtf[256]; };
struct BrdArena { BrdMem m; uint32_t ldsCap, ldsUsed; uint32_t hbmCap, hbmUsed; };
__device__ __forceinline__ uint32_t brd_alloc(BrdArena& a, uint32_t bytes)      // -> handle, or ~0u
{
    bytes = (bytes + 7u) & ~7u;
    if (a.ldsUsed + bytes <= a.ldsCap) { const uint32_t h = a.ldsUsed; a.ldsUsed += bytes; return h; }
    if (a.m.hbm && a.hbmUsed + bytes <= a.hbmCap) { const uint32_t h = a.hbmUsed | BRD_H_HBM; a.hbmUsed += bytes; return h; }
    return ~0u;
}
__device__ __forceinline__ uint32_t brd_alloc_lds(BrdArena& a, uint32_t bytes)  // maps, modes, directories: read per symbol, so LDS or nothing
{
    bytes = (bytes + 7u) & ~7u;
    if (a.ldsUsed + bytes <= a.ldsCap) { const uint32_t h = a.ldsUsed; a.ldsUsed += bytes; return h; }
    return ~0u;
}
// lengths (0..15) of `alpha` symbols -> the canonical code in the arena
__device__ __forceinline__ uint32_t brd_build(BrdScratch* S, uint32_t alpha, BrdArena& A)
{
    const bool wide = alpha > 256u;
    const uint8_t* len = S->len; uint32_t* cnt = S->cnt; uint32_t* off = S->off;
    for (uint32_t k = 0; k < 16u; k++) cnt[k] = 0;
    for (uint32_t sy = 0; sy < alpha; sy++) cnt[len[sy]]++;
    const uint32_t nUsed = alpha - cnt[0];
    const uint32_t h = brd_alloc(A, 96u + (nUsed ? nUsed : 1u) * (wide ? 2u : 1u));
    if (h == ~0u) return h;
    uint8_t* mem = (uint8_t*)brd_at(A.m, h);
    uint16_t* T = (uint16_t*)mem;
    uint32_t code = 0, index = 0;
    T[0] = 0; T[16] = 0; T[32] = 0;
    for (uint32_t L = 1; L <= 15u; L++) { T[L] = (uint16_t)cnt[L]; T[16u + L] = (uint16_t)code; T[32u + L] = (uint16_t)index; off[L] = index; index += cnt[L]; code = (code + cnt[L]) << 1; }
    for (uint32_t sy = 0; sy < alpha; sy++) { const uint32_t L = len[sy]; if (L) { const uint32_t k = off[L]++; if (wide) ((uint16_t*)(mem + 96u))[k] = (uint16_t)sy; else mem[96u + k] = (uint8_t)sy; } }
    return h | (wide ? BRD_H_WIDE : 0u);
}
__device__ __forceinline__ uint32_t brd_build_single(uint32_t sym, bool wide, BrdArena& A)
{
    const uint32_t h = brd_alloc(A, 96u + 2u);
    if (h == ~0u) return h;
    uint8_t* mem = (uint8_t*)brd_at(A.m, h);
    uint16_t* T = (uint16_t*)mem; for (uint32_t k = 0; k < 48u; k++) T[k] = 0;
    T[0] = 1;
    if (wide) ((uint16_t*)(mem + 96u))[0] = (uint16_t)sym; else mem[96] = (uint8_t)sym;
    return h | (wide ? BRD_H_WIDE : 0u);
}

// Reads one prefix code over an alphabet of `alpha` symbols (RFC 7932 sections 3.4 / 3.5) into the arena.  len[] = scratch of alpha bytes.  One lane.  -> handle
__device__ __forceinline__ uint32_t brd_read_code(BrdBits& b, uint32_t alpha, BrdArena& A, BrdScratch* S, uint32_t& status)
{
    uint8_t* const len = S->len;
    const uint32_t hskip = brd_take(b, 2);
    uint32_t h;
    if (hskip == 1u) {                                            // simple code: 1..4 symbols
        const uint32_t nsym = brd_take(b, 2) + 1u;
        uint32_t abits = 0; while ((1u << abits) < alpha) abits++;
        uint32_t s[4] = { 0, 0, 0, 0 };
        for (uint32_t i = 0; i < nsym; i++) { s[i] = brd_take(b, abits); if (s[i] >= alpha) { status = BRD_CORRUPT; return ~0u; } }
        for (uint32_t i = 0; i < nsym; i++) for (uint32_t j = i + 1u; j < nsym; j++) if (s[i] == s[j]) { status = BRD_CORRUPT; return ~0u; }
        if (nsym == 1u) h = brd_build_single(s[0], alpha > 256u, A);
        else {
            for (uint32_t i = 0; i < alpha; i++) len[i] = 0;
            if (nsym == 2u) { len[s[0]] = 1; len[s[1]] = 1; }
            else if (nsym == 3u) { len[s[0]] = 1; len[s[1]] = 2; len[s[2]] = 2; }
            else if (brd_take(b, 1)) { len[s[0]] = 1; len[s[1]] = 2; len[s[2]] = 3; len[s[3]] = 3; }
            else { len[s[0]] = 2; len[s[1]] = 2; len[s[2]] = 2; len[s[3]] = 2; }
            h = brd_build(S, alpha, A);
        }
        if (h == ~0u) status = BRD_LIMIT;
        return h;
    }
    // complex code: the lengths of the 18 code length symbols (a fixed code of 2-4 bits each), then the symbols' lengths under that code
    uint8_t* const cl = S->cl; for (uint32_t i = 0; i < 18u; i++) cl[i] = 0;
    int space = 32; uint32_t numCodes = 0;
    for (uint32_t i = hskip; i < 18u && space > 0; i++) {
        if (b.n <= 32u) brd_fill(b);
        const uint32_t p = (uint32_t)b.acc & 15u;
        uint32_t v, nb;
        if ((p & 3u) == 0u) { v = 0; nb = 2; } else if ((p & 3u) == 2u) { v = 3; nb = 2; } else if ((p & 3u) == 1u) { v = 4; nb = 2; }
        else if ((p & 4u) == 0u) { v = 2; nb = 3; } else if ((p & 8u) == 0u) { v = 1; nb = 4; } else { v = 5; nb = 4; }
        b.acc >>= nb; b.n -= nb;
        cl[kdClcOrder[i]] = (uint8_t)v;
        if (v) { space -= 32 >> v; numCodes++; }
    }
    if (!(numCodes == 1u || space == 0)) { status = BRD_CORRUPT; return ~0u; }
    // the code length code itself: a small canonical code in registers / private memory
    uint32_t* const cCnt = S->cc; uint32_t* const cFirst = S->cc + 6; uint32_t* const cBase = S->cc + 12; uint8_t* const csym = S->csym; uint32_t single = 0;
    for (uint32_t i = 0; i < 6u; i++) { cCnt[i] = 0; cFirst[i] = 0; cBase[i] = 0; }
    if (numCodes == 1u) { for (uint32_t sy = 0; sy < 18u; sy++) if (cl[sy]) single = sy; }
    else { uint32_t code = 0, k = 0; for (uint32_t L = 1; L <= 5u; L++) { cFirst[L] = code; cBase[L] = k; for (uint32_t sy = 0; sy < 18u; sy++) if (cl[sy] == L) { csym[k++] = (uint8_t)sy; cCnt[L]++; } code = (code + cCnt[L]) << 1; } }
    uint32_t i = 0, prev = 8, rep = 0, repLen = 0;
    int sp = 32768;
    while (i < alpha && sp > 0) {
        uint32_t v = single;
        if (numCodes != 1u) {
            if (b.n <= 32u) brd_fill(b);
            const uint32_t rev = __brev((uint32_t)b.acc);
            uint32_t L = 1;
            for (; L <= 5u; L++) { const uint32_t c = rev >> (32u - L); if (c - cFirst[L] < cCnt[L]) { v = csym[cBase[L] + c - cFirst[L]]; break; } }
            if (L > 5u) { status = BRD_CORRUPT; return ~0u; }
            b.acc >>= L; b.n -= L;
        }
        if (b.over) { status = BRD_CORRUPT; return ~0u; }
        if (v < 16u) {
            len[i++] = (uint8_t)v; rep = 0;
            if (v) { prev = v; sp -= 32768 >> v; }
        } else {
            const uint32_t extra = v == 16u ? 2u : 3u, newLen = v == 16u ? prev : 0u;
            if (repLen != newLen) { rep = 0; repLen = newLen; }
            const uint32_t old = rep;
            if (rep > 0u) rep = (rep - 2u) << extra;
            rep += brd_take(b, extra) + 3u;
            const uint32_t delta = rep - old;
            if (i + delta > alpha) { status = BRD_CORRUPT; return ~0u; }
            for (uint32_t k = 0; k < delta; k++) len[i + k] = (uint8_t)newLen;
            i += delta;
            if (newLen) sp -= (int)(delta << (15u - newLen));
        }
    }
    if (sp != 0) { status = BRD_CORRUPT; return ~0u; }
    for (; i < alpha; i++) len[i] = 0;
    h = brd_build(S, alpha, A);
    if (h == ~0u) status = BRD_LIMIT;
    return h;
}

__device__ __forceinline__ uint32_t brd_varlen8(BrdBits& b)     // VarLenUint8: 0..255
{
    if (!brd_take(b, 1)) return 0u;
    const uint32_t nb = brd_take(b, 3);
    return nb ? (1u << nb) + brd_take(b, nb) : 1u;
}

// context ids of the literal context modes (RFC 7932 section 7.1): for UTF8 lut[0..255] of the last byte | lut[256..511] of the one before
__device__ __forceinline__ uint32_t brd_utf8_0(uint32_t b)
{
    if (b >= 192u) return 2u + (b & 1u);
    if (b >= 128u) return b & 1u;
    const uint32_t l = b | 0x20u; const bool letter = l >= 'a' && l <= 'z';
    const bool vowel = l == 'a' || l == 'e' || l == 'i' || l == 'o' || l == 'u';
    uint32_t k = 3u;
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
    else if (letter) k = (b < 'a' ? 12u : 14u) + (vowel ? 0u : 1u);
    return 4u * k;
}
__device__ __forceinline__ uint32_t brd_utf8_1(uint32_t b)
{
    if (b >= 224u) return 2u;
    if (b >= 128u) return 0u;
    if (b <= 32u || b == 127u) return 0u;
    if ((b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z')) return 2u;
    if (b >= 'a' && b <= 'z') return 3u;
    return 1u;
}
__device__ __forceinline__ uint32_t brd_signed(uint32_t b) { return b == 0u ? 0u : (b < 16u ? 1u : (b < 64u ? 2u : (b < 128u ? 3u : (b < 192u ? 4u : (b < 240u ? 5u : (b < 255u ? 6u : 7u)))))); }

// the format's small tables, copied to LDS once per wave (a read of __constant__ memory with a computed index is a trip to the scalar cache per symbol)
struct alignas(8) BrdCmd { uint32_t x, y; };
struct BrdConst { uint16_t insBase[24], copyBase[24], blockBase[26]; uint8_t insExtra[24], copyExtra[24], blockExtra[26], cellIns[12], cellCopy[12]; };

// one category of block types (literals, insert-and-copy, distances): RFC 7932 section 6
struct BrdBlocks { uint32_t n, type, prev, left, typeCode, countCode; };
__device__ __forceinline__ void brd_switch_w(BrdBits& b, BrdBlocks& B, const BrdMem& m, const BrdConst& K, uint32_t lane)
{
    const uint32_t s = brd_sym_w(b, m, B.typeCode, lane);
    uint32_t t = s == 0u ? B.prev : (s == 1u ? B.type + 1u : s - 2u);
    if (t >= B.n) t -= B.n;
    B.prev = B.type; B.type = t;
    const uint32_t cs = brd_sym_w(b, m, B.countCode, lane);
    B.left = cs < 26u ? gc_uniform(K.blockBase[cs]) + brd_take(b, gc_uniform(K.blockExtra[cs])) : 0u;
}

// context map (RFC 7932 section 7.3): `size` entries in out[].  One lane.
__device__ __forceinline__ bool brd_context_map(BrdBits& b, uint32_t size, uint32_t& nTrees, uint8_t* out, BrdArena& A, BrdScratch* lenScratch, uint32_t& status)
{
    nTrees = brd_varlen8(b) + 1u;
    if (nTrees == 1u) { for (uint32_t i = 0; i < size; i++) out[i] = 0; return true; }
    uint32_t rleMax = 0;
    if (brd_take(b, 1)) rleMax = brd_take(b, 4) + 1u;
    BrdArena tmp = A;                                             // the map's own code is only needed here: its arena space is given back
    const uint32_t h = brd_read_code(b, nTrees + rleMax, tmp, lenScratch, status);
    if (h == ~0u) return false;
    for (uint32_t i = 0; i < size;) {
        const uint32_t s = brd_sym(b, tmp.m, h);
        if (b.over) { status = BRD_CORRUPT; return false; }
        if (s == 0u) out[i++] = 0;
        else if (s <= rleMax) { const uint32_t run = (1u << s) + brd_take(b, s); if (i + run > size) { status = BRD_CORRUPT; return false; } for (uint32_t k = 0; k < run; k++) out[i++] = 0; }
        else out[i++] = (uint8_t)(s - rleMax);
    }
    if (brd_take(b, 1)) {                                         // inverse move-to-front
        uint8_t* const mtf = lenScratch->mtf; for (uint32_t i = 0; i < 256u; i++) mtf[i] = (uint8_t)i;
        for (uint32_t i = 0; i < size; i++) { const uint32_t idx = out[i]; const uint8_t v = mtf[idx]; out[i] = v; for (uint32_t k = idx; k > 0u; k--) mtf[k] = mtf[k - 1u]; mtf[0] = v; }
    }
    for (uint32_t i = 0; i < size; i++) if (out[i] >= nTrees) { status = BRD_CORRUPT; return false; }
    return true;
}

// what lane 0 read from a meta-block's header, for the wave (LDS)
struct BrdMeta {
    uint64_t acc; uint32_t n, pos, over, status;
    uint32_t kind;                        // 0 compressed, 1 uncompressed (copy `mlen` bytes from srcAt), 2 nothing to produce, 3 the stream has ended
    uint32_t mlen, last, srcAt;
    uint32_t nTypes[3], left[3], typeCode[3], countCode[3];
    uint32_t npostfix, ndirect;
    uint32_t nTrees[3];                   // literal, insert-and-copy, distance codes
    uint32_t dir[3];                      // handles of the three directories (uint32 handles of the codes)
    uint32_t cmapL, cmapD, modes;         // handles
    uint32_t ctxTab, nCtxTab;             // LDS offset of the context tables (2 KiB per literal block type: tree of (last byte, class of the byte before)), and for how many types
    uint32_t tab[3], nTab[3];             // LDS offset of the decoding tables of the first nTab codes of each kind (512 / 2048 / 512 bytes each)
    uint32_t ldsUsed;
};

// The output goes to the ring first and to HBM in bursts: a byte store per literal keeps the wave's memory counter busy (every wait for the bit reader's next word then
// also waits for the stores in front of it).  flush: bytes [from, upto) of the output from the ring to HBM, 16 bytes per lane where the position is aligned.
template <uint32_t RING>
__device__ __forceinline__ void brd_flush(const uint8_t* ring, uint8_t* __restrict__ out, uint32_t from, uint32_t upto, uint32_t lane)
{
    constexpr uint32_t RMASK = RING - 1u;
    uint32_t a = from;
    const uint32_t head = (16u - (a & 15u)) & 15u, h = head < upto - a ? head : upto - a;
    if (lane < h) out[a + lane] = ring[(a + lane) & RMASK];
    a += h;
    const uint32_t n16 = (upto - a) >> 4;
    struct alignas(16) V16 { uint64_t x, y; };
    for (uint32_t k = lane; k < n16; k += 64u) *(V16*)(out + a + 16u * k) = *(const V16*)(ring + ((a + 16u * k) & RMASK));
    a += n16 << 4;
    if (lane < upto - a) out[a + lane] = ring[(a + lane) & RMASK];
}

// m literals in a row whose trees all have tables and whose context comes from one table (CLS 0: one tree; 2: UTF8 classes from the LUT; 3: the mode decides): the loop the
// decoder spends its time in on data that does not compress well -- no block-type, flush or table-presence checks inside (the caller has sized m by them)
template <uint32_t RING, uint32_t CLS>
__device__ __forceinline__ void brd_literals(BrdBits& hb, uint32_t m, const uint16_t* tabL, const uint8_t* ctRow, const uint8_t* lut, uint8_t* ring, uint32_t& pos, uint32_t& p1, uint32_t& p2,
                                             uint32_t& g1, uint32_t& g2, uint32_t mode, const uint32_t* dirL, const BrdMem& mem, uint32_t lane)
{
    for (; m != 0u; m--) {
        if (hb.n <= 32u) brd_fill(hb);
        const uint32_t tl = CLS ? gc_uniform(ctRow[(p1 << 3) | g2]) : 0u;
        const uint32_t e = gc_uniform(tabL[(tl << 8) + ((uint32_t)hb.acc & 255u)]);
        uint32_t lit;
        if (e & 0x8000u) { const uint32_t len = e & 15u; hb.acc >>= len; hb.n -= len; lit = (e >> 4) & 0xFFu; }
        else lit = brd_sym_w(hb, mem, dirL[tl], lane);
        if (lane == 0u) ring[pos & (RING - 1u)] = (uint8_t)lit;
        pos++; p2 = p1; p1 = lit;
        if (CLS == 2u) { g2 = g1; g1 = gc_uniform(lut[256u + lit]); }
        else if (CLS == 3u) { g2 = g1; g1 = mode == 3u ? brd_signed(lit) : 0u; }
    }
}

// (check build -DBRD_PROFILE: chunk 0 prints the clock ticks it spent per section)
// (check build -DBRD_STATS, emulator: chunk 0 prints where the stream's bits go -- header, command symbols + extra bits, literals, distances)
#ifdef BRD_STATS
#define BRD_BITS(hbv) ((uint64_t)(hbv).pos * 8u - (hbv).n)
#define BRD_S(k, hbv) { const uint64_t bNow = BRD_BITS(hbv); stat[k] += bNow - bLast; bLast = bNow; }
#else
#define BRD_S(k, hbv)
#endif
#ifdef BRD_PROFILE
#define BRD_T(k) { const uint64_t tNow = wall_clock64(); prof[k] += tNow - tLast; tLast = tNow; }
#else
#define BRD_T(k)
#endif
// ------------------------------------------------------------------------------------------------ one wave per chunk
// ARENA: bytes of LDS for a meta-block's codes, maps and decoding tables; RING: the last RING bytes of output, so that near copies read LDS (0: none).  The host picks the
// instance by the number of chunks: few chunks get the LDS of a whole CU each.
template <uint32_t ARENA, uint32_t RING>
__device__ __forceinline__ void brd_kernel_body(const uint8_t* __restrict__ src, const GcBrDecChunk* __restrict__ chunks, uint32_t nChunks, uint8_t* __restrict__ stage,
                                                uint8_t* __restrict__ pages, uint32_t nPages, uint32_t* __restrict__ pageCursor, GcBrDecResult* __restrict__ result, GcBrDict dict, uint32_t ldsCap)
{
    __shared__ __attribute__((aligned(8))) uint8_t sArena[ARENA];
    __shared__ __attribute__((aligned(16))) uint8_t sRing[RING];
    static_assert(RING >= 4096u && (RING & (RING - 1u)) == 0u, "the ring holds the output before it goes to HBM");
    __shared__ uint8_t sLut[512];                                 // UTF8 context ids (mode 2), the mode of every stream this engine writes and of nearly every one of the reference
    __shared__ BrdScratch sScratch;
    __shared__ BrdMeta sMeta;
    __shared__ BrdConst sK;
    __shared__ BrdCmd sCmd[704];                                   // per insert-and-copy symbol: {insert base | extra bits << 16, copy base | extra bits << 16} (RFC 7932 section 5): one read per command
    const uint32_t lane = threadIdx.x;
    constexpr uint32_t RMASK = RING - 1u;
    for (uint32_t i = lane; i < 256u; i += 64u) { sLut[i] = (uint8_t)brd_utf8_0(i); sLut[256u + i] = (uint8_t)brd_utf8_1(i); }
    if (lane < 24u) { sK.insBase[lane] = kdInsBase[lane]; sK.copyBase[lane] = kdCopyBase[lane]; sK.insExtra[lane] = kdInsExtra[lane]; sK.copyExtra[lane] = kdCopyExtra[lane]; }
    if (lane < 26u) { sK.blockBase[lane] = kdBlockBase[lane]; sK.blockExtra[lane] = kdBlockExtra[lane]; }
    if (lane < 11u) { sK.cellIns[lane] = kdCellIns[lane]; sK.cellCopy[lane] = kdCellCopy[lane]; }
    for (uint32_t cs = lane; cs < 704u; cs += 64u) {
        const uint32_t cell = cs >> 6, ic = kdCellIns[cell] + ((cs >> 3) & 7u), cc = kdCellCopy[cell] + (cs & 7u);
        BrdCmd e; e.x = kdInsBase[ic] | ((uint32_t)kdInsExtra[ic] << 16); e.y = kdCopyBase[cc] | ((uint32_t)kdCopyExtra[cc] << 16);
        sCmd[cs] = e;
    }
    gc_wave_sync();
    const BrdConst& K = sK;
    for (uint32_t c = blockIdx.x; c < nChunks; c += gridDim.x) {
        const GcBrDecChunk ck = chunks[c];
        uint8_t* const out = stage + ck.stageOff;
        const uint32_t cap = ck.hintBytes;
        BrdMem mem; mem.lds = sArena; mem.hbm = nullptr;          // (the page is taken from the pool when a meta-block needs it, kept for the chunk)
#ifdef BRD_PROFILE
        uint64_t prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tLast = wall_clock64(); uint32_t nCmd = 0, nLit = 0, nMeta = 0;
#endif
#ifdef BRD_STATS
        uint64_t stat[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, bLast = 0; uint32_t sCmdN = 0, sLitN = 0, sDistN = 0, sMetaN = 0, sCtxN = 0, sRingN = 0, sImplN = 0; uint64_t sCopyBytes = 0;
#endif
        uint32_t status = BRD_OK, pos = 0, flushed = 0;             // output [flushed, pos) is in the ring only
        bool unfenced = false;                                    // a flush has stored to HBM since the wave last waited for its stores
        BrdBits b; b.p = src + ck.srcOff; b.acc = 0; b.n = 0; b.pos = 0; b.end = ck.srcSize; b.over = 0u; brd_prime(b);
        // stream header: WBITS (RFC 7932 section 9.1)
        uint32_t wbits = 16;
        if (brd_take(b, 1)) { const uint32_t n = brd_take(b, 3); if (n) wbits = 17u + n; else { const uint32_t m = brd_take(b, 3); if (m == 1u) status = BRD_CORRUPT; else wbits = m ? 8u + m : 17u; } }
        const uint32_t maxBack = (1u << wbits) - 16u;
        int r1 = 4, r2 = 11, r3 = 15, r4 = 16;                    // the last four distances, r1 the most recent
        uint32_t p1 = 0, p2 = 0;                                  // the last two bytes of the output
        bool last = false;
        while (!last && status == BRD_OK) {
            // ---- meta-block header: lane 0 reads it (prefix codes and context maps are serial work), the wave takes over what it found
            if (lane == 0u) {
                BrdMeta& M = sMeta; M.kind = 2; M.mlen = 0; M.srcAt = 0;
                uint32_t st = BRD_OK;
                M.last = brd_take(b, 1);
                if (M.last && brd_take(b, 1)) M.kind = 3;         // ISLASTEMPTY
                else {
                    const uint32_t nibCode = brd_take(b, 2);
                    if (nibCode == 3u) {                          // metadata: skipped
                        if (brd_take(b, 1)) st = BRD_CORRUPT;
                        const uint32_t sb = brd_take(b, 2);
                        uint32_t skip = 0;
                        if (sb && st == BRD_OK) { skip = brd_take(b, 8u * sb); if (sb > 1u && (skip >> (8u * (sb - 1u))) == 0u) st = BRD_CORRUPT; skip += 1u; }
                        if (brd_take(b, b.n & 7u) != 0u) st = BRD_CORRUPT;
                        const uint32_t at = brd_consumed(b);
                        if (at + skip > b.end) st = BRD_CORRUPT;
                        b.acc = 0; b.n = 0; b.pos = at + skip; brd_prime(b);
                    } else {
                        const uint32_t nib = 4u + nibCode;
                        uint32_t mlen = 0;
                        for (uint32_t i = 0; i < nib; i++) { const uint32_t v = brd_take(b, 4); if (i + 1u == nib && nib > 4u && v == 0u) st = BRD_CORRUPT; mlen |= v << (4u * i); }
                        M.mlen = mlen + 1u;
                        if (st == BRD_OK && (uint64_t)pos + M.mlen > cap) st = BRD_DST_SMALL;
                        if (st == BRD_OK && !M.last && brd_take(b, 1)) {      // uncompressed: the bytes follow at the next byte boundary
                            if (brd_take(b, b.n & 7u) != 0u) st = BRD_CORRUPT;
                            M.srcAt = brd_consumed(b); M.kind = 1;
                            if (M.srcAt + M.mlen > b.end) st = BRD_CORRUPT;
                            b.acc = 0; b.n = 0; b.pos = M.srcAt + M.mlen; brd_prime(b);
                        } else if (st == BRD_OK) {
                            M.kind = 0;
                            // two passes at most: the second with a page of HBM behind the LDS arena
                            const BrdBits b0 = b;
                            for (uint32_t attempt = 0; attempt < 2u; attempt++) {
                                b = b0; st = BRD_OK;
                                BrdArena A; A.m = mem; A.ldsCap = ldsCap < ARENA ? ldsCap : ARENA; A.ldsUsed = 0; A.hbmCap = BRD_PAGE; A.hbmUsed = 0;
                                for (uint32_t k = 0; k < 3u && st == BRD_OK; k++) {
                                    M.nTypes[k] = brd_varlen8(b) + 1u; M.left[k] = ~0u; M.typeCode[k] = 0; M.countCode[k] = 0;
                                    if (M.nTypes[k] >= 2u) {
                                        M.typeCode[k] = brd_read_code(b, M.nTypes[k] + 2u, A, &sScratch, st); if (st != BRD_OK) break;
                                        M.countCode[k] = brd_read_code(b, 26u, A, &sScratch, st); if (st != BRD_OK) break;
                                        const uint32_t cs = brd_sym(b, A.m, M.countCode[k]);
                                        M.left[k] = cs < 26u ? kdBlockBase[cs] + brd_take(b, kdBlockExtra[cs]) : 0u;
                                    }
                                }
                                if (st == BRD_OK) {
                                    M.npostfix = brd_take(b, 2); M.ndirect = brd_take(b, 4) << M.npostfix;
                                    M.modes = brd_alloc_lds(A, M.nTypes[0]);
                                    if (M.modes == ~0u) st = BRD_LIMIT;
                                    else { uint8_t* md = sArena + M.modes; for (uint32_t i = 0; i < M.nTypes[0]; i++) md[i] = (uint8_t)brd_take(b, 2); }
                                }
                                M.nTrees[0] = 1; M.nTrees[1] = M.nTypes[1]; M.nTrees[2] = 1;
                                if (st == BRD_OK) {
                                    const uint32_t sl = 64u * M.nTypes[0], sd = 4u * M.nTypes[2];
                                    M.cmapL = brd_alloc_lds(A, sl + sd); M.cmapD = M.cmapL + sl;
                                    if (M.cmapL == ~0u) st = BRD_LIMIT;
                                    if (st == BRD_OK) brd_context_map(b, sl, M.nTrees[0], sArena + M.cmapL, A, &sScratch, st);
                                    if (st == BRD_OK) brd_context_map(b, sd, M.nTrees[2], sArena + M.cmapD, A, &sScratch, st);
                                }
                                if (st == BRD_OK) {
                                    const uint32_t d0 = brd_alloc_lds(A, (M.nTrees[0] + M.nTrees[1] + M.nTrees[2]) * 4u);
                                    if (d0 == ~0u) st = BRD_LIMIT;
                                    else {
                                        M.dir[0] = d0; M.dir[1] = d0 + 4u * M.nTrees[0]; M.dir[2] = M.dir[1] + 4u * M.nTrees[1];
                                        const uint32_t alpha[3] = { 256u, 704u, 16u + M.ndirect + (48u << M.npostfix) };
                                        for (uint32_t k = 0; k < 3u && st == BRD_OK; k++) {
                                            uint32_t* dir = (uint32_t*)(sArena + M.dir[k]);
                                            for (uint32_t i = 0; i < M.nTrees[k] && st == BRD_OK; i++) dir[i] = brd_read_code(b, alpha[k], A, &sScratch, st);
                                        }
                                    }
                                }
                                M.ldsUsed = A.ldsUsed;
                                if (st != BRD_LIMIT || mem.hbm) break;
                                // LDS does not hold this meta-block: take a page of the pool and read the header again
                                const uint32_t pg = atomicAdd(pageCursor, 1u);
                                if (pg >= nPages) break;
                                mem.hbm = pages + (uint64_t)pg * BRD_PAGE;
                            }
                            // decoding tables in what the arena has left: literal codes first (one lookup per byte), then the command codes, then the distance codes
                            if (st == BRD_OK) {
                                uint32_t at = (M.ldsUsed + 7u) & ~7u;
                                const uint32_t capT = ldsCap < ARENA ? ldsCap : ARENA;
                                const uint32_t bytesOf[3] = { 512u, 2048u, 512u };
                                for (uint32_t k = 0; k < 3u; k++) {
                                    const uint32_t room = capT > at ? (capT - at) / bytesOf[k] : 0u;
                                    M.tab[k] = at; M.nTab[k] = room < M.nTrees[k] ? room : M.nTrees[k];
                                    at += M.nTab[k] * bytesOf[k];
                                }
                                M.ctxTab = at; M.nCtxTab = 0;
                                if (M.nTrees[0] > 1u && M.nTab[0] == M.nTrees[0] && capT > at && (capT - at) / 2048u >= M.nTypes[0]) M.nCtxTab = M.nTypes[0];
                            }
                        }
                    }
                }
                if (b.over && st == BRD_OK) st = BRD_CORRUPT;
                M.acc = b.acc; M.n = b.n; M.pos = b.pos; M.over = b.over; M.status = st;
            }
            gc_wave_sync_global();
            BRD_T(0)
            const uint32_t kind = gc_uniform(sMeta.kind), mlen = gc_uniform(sMeta.mlen);
            status = gc_uniform(sMeta.status); last = gc_uniform(sMeta.last) != 0u;
            b.acc = sMeta.acc; b.n = sMeta.n; b.pos = sMeta.pos; b.over = sMeta.over; brd_prime(b);
            if (status != BRD_OK || kind == 3u) { gc_wave_sync(); break; }
            if (kind == 2u) { gc_wave_sync(); continue; }
            if (kind == 1u) {
                const uint8_t* s = src + ck.srcOff + gc_uniform(sMeta.srcAt);
                gc_wave_sync();
                brd_flush<RING>(sRing, out, flushed, pos, lane);
                for (uint32_t i = lane; i < mlen; i += 64u) { const uint8_t v = s[i]; out[pos + i] = v; sRing[(pos + i) & RMASK] = v; }
                p2 = mlen > 1u ? gc_uniform(s[mlen - 2u]) : p1; p1 = gc_uniform(s[mlen - 1u]);
                pos += mlen; flushed = pos;
                gc_wave_sync();
                continue;
            }
            // (the reader of the commands is a value of its own: the header code above hands its reader to functions, which pins that one to memory)
            BrdBits hb; hb.p = src + ck.srcOff; hb.end = ck.srcSize; { const uint64_t a = sMeta.acc; hb.acc = (uint64_t)gc_uniform((uint32_t)a) | ((uint64_t)gc_uniform((uint32_t)(a >> 32)) << 32); } hb.n = gc_uniform(sMeta.n); hb.pos = gc_uniform(sMeta.pos); hb.over = gc_uniform(sMeta.over); brd_prime(hb);
            // ---- the page, if lane 0 took one: its address travels as the pool index
            { uint64_t hp = (uint64_t)(uintptr_t)mem.hbm; hp = __shfl(hp, 0); mem.hbm = (uint8_t*)(uintptr_t)hp; }
            // ---- decoding tables
            const uint32_t nTabL = gc_uniform(sMeta.nTab[0]), nTabI = gc_uniform(sMeta.nTab[1]), nTabD = gc_uniform(sMeta.nTab[2]);
            const uint16_t* const tabL = (const uint16_t*)(sArena + sMeta.tab[0]); const uint16_t* const tabI = (const uint16_t*)(sArena + sMeta.tab[1]); const uint16_t* const tabD = (const uint16_t*)(sArena + sMeta.tab[2]);
            const uint32_t* const dirL = (const uint32_t*)(sArena + sMeta.dir[0]); const uint32_t* const dirI = (const uint32_t*)(sArena + sMeta.dir[1]); const uint32_t* const dirD = (const uint32_t*)(sArena + sMeta.dir[2]);
            for (uint32_t i = 0; i < nTabL; i++) brd_table_w(mem, dirL[i], 8u, (uint16_t*)tabL + 256u * i, lane);
            for (uint32_t i = 0; i < nTabI; i++) brd_table_w(mem, dirI[i], 10u, (uint16_t*)tabI + 1024u * i, lane);
            for (uint32_t i = 0; i < nTabD; i++) brd_table_w(mem, dirD[i], 8u, (uint16_t*)tabD + 256u * i, lane);
            // ---- context tables: the literal's tree from the last byte and the class of the byte before it in ONE read (context id, then context map, are two)
            const uint32_t nCtxTab = gc_uniform(sMeta.nCtxTab), nTreesL = gc_uniform(sMeta.nTrees[0]), nTreesD = gc_uniform(sMeta.nTrees[2]);
            uint8_t* const ctxTab = sArena + sMeta.ctxTab;
            for (uint32_t t = 0; t < nCtxTab; t++) {
                const uint32_t md = (sArena + sMeta.modes)[t]; const uint8_t* row = sArena + sMeta.cmapL + 64u * t;
                for (uint32_t e = lane; e < 2048u; e += 64u) {
                    const uint32_t a = e >> 3, j = e & 7u;
                    const uint32_t ctx = md == 2u ? (sLut[a] | (j & 3u)) : (md == 0u ? (a & 63u) : (md == 1u ? a >> 2 : ((brd_signed(a) << 3) | j)));
                    ctxTab[2048u * t + e] = row[ctx];
                }
            }
            // ---- commands: every lane runs the same state machine
            BrdBlocks BL[3];
            for (uint32_t k = 0; k < 3u; k++) { BL[k].n = gc_uniform(sMeta.nTypes[k]); BL[k].type = 0; BL[k].prev = 1; BL[k].left = gc_uniform(sMeta.left[k]); BL[k].typeCode = gc_uniform(sMeta.typeCode[k]); BL[k].countCode = gc_uniform(sMeta.countCode[k]); }
            const uint32_t npostfix = gc_uniform(sMeta.npostfix), ndirect = gc_uniform(sMeta.ndirect);
            const uint8_t* const cmapL = sArena + sMeta.cmapL; const uint8_t* const cmapD = sArena + sMeta.cmapD; const uint8_t* const modes = sArena + sMeta.modes;
            gc_wave_sync();                                       // (sMeta is lane 0's to write again from here; the tables are whole)
            BRD_T(1)
            BRD_S(0, hb)
#ifdef BRD_STATS
            sMetaN++; if (nTreesL > 1u) sCtxN++;
#endif
            const uint32_t mEnd = pos + mlen;
            uint32_t mode = gc_uniform(modes[0]);
            const uint8_t* cmRow = cmapL;
            const uint8_t* ctRow = ctxTab;
            // class of a byte as the byte BEFORE the last one (UTF8: 2 bits, SIGNED: 3 bits, the other modes do not look at it)
            #define BRD_CLASS(x) (mode == 2u ? (uint32_t)sLut[256u + (x)] : (mode == 3u ? brd_signed(x) : 0u))
            uint32_t g2 = nCtxTab ? gc_uniform(BRD_CLASS(p2)) : 0u, g1 = nCtxTab ? gc_uniform(BRD_CLASS(p1)) : 0u;
            const bool fastLit = nTabL == nTreesL && (nTreesL == 1u || nCtxTab != 0u);
            uint32_t stalled = 0;
            bool gStale = false;                                  // g1 / g2 are behind p1 / p2 (a copy has run: they are looked up when the next literal needs them)
            while (pos < mEnd && status == BRD_OK) {
                if (BL[1].left == 0u) brd_switch_w(hb, BL[1], mem, K, lane);
                BL[1].left--;
                const uint32_t ti = BL[1].type;
                const uint32_t cs = brd_sym_t<10>(hb, tabI, nTabI, dirI, ti, mem, lane);
                if (cs >= 704u) { status = BRD_CORRUPT; break; }
                const BrdCmd ci = sCmd[cs];
                const uint32_t ciI = gc_uniform(ci.x), ciC = gc_uniform(ci.y);
                uint32_t ins = ciI & 0xFFFFu; if (ciI >> 16) ins += brd_take(hb, ciI >> 16);
                uint32_t cplen = ciC & 0xFFFFu; if (ciC >> 16) cplen += brd_take(hb, ciC >> 16);
                if (pos + ins > mEnd) { status = BRD_CORRUPT; break; }
                BRD_T(2)
                BRD_S(1, hb)
#ifdef BRD_STATS
                sCmdN++; sLitN += ins; sCopyBytes += cplen; if (cs < 128u) sImplN++;
#endif
#ifdef BRD_PROFILE
                nCmd++; nLit += ins;
#endif
                if (gStale && ins != 0u) { g2 = gc_uniform(BRD_CLASS(p2)); g1 = gc_uniform(BRD_CLASS(p1)); gStale = false; }
                while (fastLit && ins != 0u) {
                    if (BL[0].left == 0u) {
                        brd_switch_w(hb, BL[0], mem, K, lane); mode = gc_uniform(modes[BL[0].type]); ctRow = ctxTab + 2048u * BL[0].type;
                        if (nCtxTab) { g2 = gc_uniform(BRD_CLASS(p2)); g1 = gc_uniform(BRD_CLASS(p1)); }
                    }
                    uint32_t room = RING - 64u - (pos - flushed);
                    if (room == 0u) { gc_wave_sync(); brd_flush<RING>(sRing, out, flushed, pos, lane); flushed = pos; unfenced = true; gc_wave_sync(); room = RING - 64u; }
                    uint32_t m = ins < BL[0].left ? ins : BL[0].left;
                    if (room < m) m = room;
                    ins -= m; BL[0].left -= m;
                    if (nTreesL == 1u) brd_literals<RING, 0u>(hb, m, tabL, ctRow, sLut, sRing, pos, p1, p2, g1, g2, mode, dirL, mem, lane);
                    else if (mode == 2u) brd_literals<RING, 2u>(hb, m, tabL, ctRow, sLut, sRing, pos, p1, p2, g1, g2, mode, dirL, mem, lane);
                    else brd_literals<RING, 3u>(hb, m, tabL, ctRow, sLut, sRing, pos, p1, p2, g1, g2, mode, dirL, mem, lane);
                }
                for (; ins != 0u; ins--) {
                    if (BL[0].left == 0u) {
                        brd_switch_w(hb, BL[0], mem, K, lane); mode = gc_uniform(modes[BL[0].type]); cmRow = cmapL + 64u * BL[0].type; ctRow = ctxTab + 2048u * BL[0].type;
                        if (nCtxTab) { g2 = gc_uniform(BRD_CLASS(p2)); g1 = gc_uniform(BRD_CLASS(p1)); }
                    }
                    BL[0].left--;
                    uint32_t tl = 0;
                    if (nCtxTab) tl = gc_uniform(ctRow[(p1 << 3) | g2]);
                    else if (nTreesL > 1u) {
                        uint32_t ctx;
                        if (mode == 2u) ctx = gc_uniform(sLut[p1] | sLut[256u + p2]);
                        else if (mode == 0u) ctx = p1 & 63u;
                        else if (mode == 1u) ctx = p1 >> 2;
                        else ctx = (brd_signed(p1) << 3) | brd_signed(p2);
                        tl = gc_uniform(cmRow[ctx]);
                    }
                    const uint32_t lit = brd_sym_t<8>(hb, tabL, nTabL, dirL, tl, mem, lane);
                    if (pos - flushed >= RING - 64u) { gc_wave_sync(); brd_flush<RING>(sRing, out, flushed, pos, lane); flushed = pos; unfenced = true; gc_wave_sync(); }
                    if (lane == 0u) sRing[pos & RMASK] = (uint8_t)lit;
                    pos++; p2 = p1; p1 = lit;
                    if (nCtxTab) { g2 = g1; g1 = gc_uniform(BRD_CLASS(lit)); }
                }
                BRD_T(3)
                BRD_S(2, hb)
                if (hb.over) { status = BRD_CORRUPT; break; }
                if (pos == mEnd) break;                           // the meta-block ends behind the literals: no copy
                int dist;
                uint32_t dcode = 0;
                if (cs >= 128u) {
                    if (BL[2].left == 0u) brd_switch_w(hb, BL[2], mem, K, lane);
                    BL[2].left--;
                    const uint32_t dctx = cplen > 4u ? 3u : cplen - 2u;
                    const uint32_t td = nTreesD > 1u ? gc_uniform(cmapD[4u * BL[2].type + dctx]) : 0u;
                    dcode = brd_sym_t<8>(hb, tabD, nTabD, dirD, td, mem, lane);
                }
                const uint32_t maxDist = pos < maxBack ? pos : maxBack;
                bool push = true;
                if (dcode >= 16u + ndirect) {
                    const uint32_t v = dcode - ndirect - 16u, hcode = v >> npostfix, lcode = v & ((1u << npostfix) - 1u), nb = 1u + (hcode >> 1);
                    const uint32_t off = ((2u + (hcode & 1u)) << nb) - 4u;
                    dist = (int)(((off + brd_take(hb, nb)) << npostfix) + lcode + ndirect + 1u);
                    if (dist <= 0) { status = BRD_CORRUPT; break; }
                } else if (dcode < 16u) {
                    if (dcode == 0u) { dist = r1; push = false; }
                    else if (dcode == 1u) dist = r2; else if (dcode == 2u) dist = r3; else if (dcode == 3u) dist = r4;
                    else if (dcode < 10u) { const int d = (int)((dcode - 4u) >> 1) + 1; dist = r1 + (((dcode - 4u) & 1u) ? d : -d); }
                    else { const int d = (int)((dcode - 10u) >> 1) + 1; dist = r2 + (((dcode - 10u) & 1u) ? d : -d); }
                    if (dist <= 0) { status = BRD_CORRUPT; break; }
                } else dist = (int)(dcode - 15u);
                if (hb.over) { status = BRD_CORRUPT; break; }
                if ((uint32_t)dist > maxDist) {
                    // ---- static dictionary reference (RFC 7932 section 8): the word of `cplen` bytes with one of the 121 transforms; it does not enter the ring
                    if (cplen < 4u || cplen > 24u) { status = BRD_CORRUPT; break; }
                    if (dict.words == nullptr) { status = BRD_DICTIONARY; break; }
                    const uint32_t id = (uint32_t)dist - maxDist - 1u, nbits = kdDictBits[cplen];
                    const uint32_t widx = id & ((1u << nbits) - 1u), tidx = id >> nbits;
                    if (tidx >= GC_BR_NUM_TRANSFORMS) { status = BRD_CORRUPT; break; }
                    const uint8_t* w = dict.words + kdDictOff[cplen] + widx * cplen;
                    const uint32_t type = kdTransforms[tidx][1];
                    const uint8_t* pre = kdAffixPool + kdTransforms[tidx][0]; const uint8_t* suf = kdAffixPool + kdTransforms[tidx][2];
                    uint32_t wl = cplen, skip = 0;
                    if (type >= 12u && type <= 20u) { skip = type - 11u; if (skip > wl) skip = wl; }            // omit the first n
                    else if (type >= 1u && type <= 9u) wl = wl > type ? wl - type : 0u;                        // omit the last n
                    const uint32_t body = wl - skip, total = pre[0] + body + suf[0];
                    if (pos + total > mEnd) { status = BRD_CORRUPT; break; }
                    // (a transform may leave nothing of a word; a damaged stream whose codes all have one symbol could ask for that forever without spending a bit)
                    if (total == 0u && ++stalled > 4096u) { status = BRD_CORRUPT; break; }
                    if (pos - flushed + total + 64u > RING) { gc_wave_sync(); brd_flush<RING>(sRing, out, flushed, pos, lane); flushed = pos; unfenced = true; gc_wave_sync(); }
                    if (lane == 0u) {
                        uint8_t word[40]; uint32_t o = 0;         // (prefix <= 8, word <= 24, suffix <= 8 bytes: RFC 7932 Appendix B)
                        for (uint32_t i = 0; i < pre[0]; i++) word[o++] = pre[1u + i];
                        const uint32_t w0 = o;
                        for (uint32_t i = skip; i < wl; i++) word[o++] = w[i];
                        if (type == 10u || type == 11u) {         // uppercase the first / every character (UTF-8 aware as the RFC defines it)
                            uint32_t q = w0;
                            while (q < o) {
                                uint32_t step;
                                if (word[q] < 192u) { if (word[q] >= 'a' && word[q] <= 'z') word[q] ^= 32u; step = 1; }
                                else if (word[q] < 224u) { if (q + 1u < o) word[q + 1u] ^= 32u; step = 2; }
                                else { if (q + 2u < o) word[q + 2u] ^= 5u; step = 3; }
                                if (type == 10u) break;
                                q += step;
                            }
                        }
                        for (uint32_t i = 0; i < suf[0]; i++) word[o++] = suf[1u + i];
                        for (uint32_t i = 0; i < o; i++) sRing[(pos + i) & RMASK] = word[i];
                    }
                    gc_wave_sync();
                    if (total) { p2 = total > 1u ? gc_uniform(sRing[(pos + total - 2u) & RMASK]) : p1; p1 = gc_uniform(sRing[(pos + total - 1u) & RMASK]); }
                    gStale = nCtxTab != 0u;
                    pos += total;
                    gc_wave_sync();
                    continue;
                }
                if (push) { r4 = r3; r3 = r2; r2 = r1; r1 = dist; }
                if (pos + cplen > mEnd) { status = BRD_CORRUPT; break; }
                BRD_T(4)
                BRD_S(3, hb)
#ifdef BRD_STATS
                if (cs >= 128u) { sDistN++; if (dcode < 16u) sRingN++; }
#ifdef BRD_DUMP
                if (c == 0u && lane == 0u) printf("C %u %u %d %u\n", pos, cplen, dist, cs < 128u ? 0u : (dcode < 16u ? 1u : 2u));
#endif
#endif
                {
                    const uint32_t d = (uint32_t)dist;
                    uint32_t v = 0;                               // the byte of this lane's last turn
                    gc_wave_sync();                               // lane 0's literals in the ring
                    // (gc_wave_step: the hardware runs a wave's LDS operations in program order; the emulator's lanes must not write the ring before the others have read it)
                    if (pos - flushed + cplen + 64u > RING) { brd_flush<RING>(sRing, out, flushed, pos, lane); flushed = pos; unfenced = true; gc_wave_step(); }
                    if (cplen + 64u > RING / 2u) {
                        // a copy the ring does not take beside what it holds: HBM to HBM (everything in front of it is there now), the ring follows
                        if (flushed != pos) { brd_flush<RING>(sRing, out, flushed, pos, lane); flushed = pos; }
                        gc_wave_sync_global(); unfenced = false;
                        uint8_t* const o = out + pos; const uint8_t* const s = o - d;
                        for (uint32_t i = lane; i < cplen; i += 64u) { v = s[d < cplen ? i % d : i]; o[i] = (uint8_t)v; sRing[(pos + i) & RMASK] = (uint8_t)v; }
                        flushed = pos + cplen;
                    } else if (d + cplen + 64u <= RING) {
                        // near: distance + length + 64 <= RING, so no slot this copy writes holds a byte it still reads, in whatever order the lanes run
                        if (cplen <= 64u && d >= cplen) { if (lane < cplen) { v = sRing[(pos - d + lane) & RMASK]; sRing[(pos + lane) & RMASK] = (uint8_t)v; } }
                        else for (uint32_t i = lane; i < cplen; i += 64u) { v = sRing[(pos - d + (d < cplen ? i % d : i)) & RMASK]; sRing[(pos + i) & RMASK] = (uint8_t)v; }
                    } else {
                        // far: the source is in HBM unless it reaches into what the ring has not handed over yet
                        if (pos - d + cplen > flushed) { brd_flush<RING>(sRing, out, flushed, pos, lane); flushed = pos; unfenced = true; }
                        if (unfenced) { gc_wave_sync_global(); unfenced = false; }
                        const uint8_t* const s = out + pos - d;
                        for (uint32_t i = lane; i < cplen; i += 64u) { v = s[d < cplen ? i % d : i]; sRing[(pos + i) & RMASK] = (uint8_t)v; }
                    }
                    p1 = gc_readlane(v, (cplen - 1u) & 63u); p2 = gc_readlane(v, (cplen - 2u) & 63u);
                    gStale = nCtxTab != 0u;
                }
                pos += cplen;
                gc_wave_sync();
                BRD_T(5)
            }
            #undef BRD_CLASS
            b = hb;
            if (b.over && status == BRD_OK) status = BRD_CORRUPT;
            gc_wave_sync_global();
        }
        gc_wave_sync();
        brd_flush<RING>(sRing, out, flushed, pos, lane);
#ifdef BRD_STATS
        if (c == 0u && lane == 0u) printf("brd stats chunk 0: bytes %u meta-blocks %u (with context maps %u) commands %u (implicit distance %u) literals %u distance symbols %u (ring codes %u) copy bytes %llu | bits: header %llu command %llu literal %llu distance %llu\n",
            pos, sMetaN, sCtxN, sCmdN, sImplN, sLitN, sDistN, sRingN, (unsigned long long)sCopyBytes, (unsigned long long)stat[0], (unsigned long long)stat[1], (unsigned long long)stat[2], (unsigned long long)stat[3]);
#endif
#ifdef BRD_PROFILE
        BRD_T(6)
        if (c == 0u && lane == 0u) printf("brd profile chunk 0: bytes %u commands %u literals %u | ticks (100 MHz): header %llu tables %llu command %llu literals %llu distance %llu copy %llu rest %llu\n", pos, nCmd, nLit,
            (unsigned long long)prof[0], (unsigned long long)prof[1], (unsigned long long)prof[2], (unsigned long long)prof[3], (unsigned long long)prof[4], (unsigned long long)prof[5], (unsigned long long)prof[6]);
#endif
        if (status == BRD_OK && brd_consumed(b) > ck.srcSize) status = BRD_CORRUPT;
        if (lane == 0u) { GcBrDecResult r; r.size = pos; r.status = status; result[c] = r; }
        gc_wave_sync_global();
    }
}
#define BRD_INSTANCE(NAME, ARENA, RING) \
extern "C" __global__ void __launch_bounds__(64) NAME(const uint8_t* __restrict__ src, const GcBrDecChunk* __restrict__ chunks, uint32_t nChunks, uint8_t* __restrict__ stage, \
    uint8_t* __restrict__ pages, uint32_t nPages, uint32_t* __restrict__ pageCursor, GcBrDecResult* __restrict__ result, GcBrDict dict, uint32_t ldsCap) \
{ brd_kernel_body<ARENA, RING>(src, chunks, nChunks, stage, pages, nPages, pageCursor, result, dict, ldsCap); }
BRD_INSTANCE(gc_brotli_dec_kernel_a, 81920u, 65536u)      // up to 256 chunks: a CU's LDS per wave
BRD_INSTANCE(gc_brotli_dec_kernel_b, 45056u, 32768u)      // up to 512: two waves per CU
BRD_INSTANCE(gc_brotli_dec_kernel_c, 20480u, 16384u)      // up to 1024: four
BRD_INSTANCE(gc_brotli_dec_kernel_d, 14336u, 4096u)       // more: seven

// sizes -> offsets (one workgroup), then the packed copy (a workgroup per 64 KiB of a chunk)
extern "C" __global__ void __launch_bounds__(1024)
gc_brotli_dec_plan_kernel(const GcBrDecResult* __restrict__ result, uint32_t nChunks, uint64_t dstCap, uint64_t* __restrict__ offs, uint64_t* __restrict__ total /* [0] bytes, [1] status */)
{
    __shared__ uint64_t sWave[16];
    __shared__ uint32_t sBad;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) sBad = 0;
    __syncthreads();
    uint64_t carry = 0; uint32_t bad = 0;
    for (uint32_t tb = 0; tb < nChunks; tb += 1024u) {
        const uint32_t i = tb + t;
        uint64_t v = 0;
        if (i < nChunks) { v = result[i].size; if (result[i].status > bad) bad = result[i].status; }
        uint64_t incl = v;
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint64_t o = __shfl_up(incl, d); if (lane >= d) incl += o; }
        if (lane == 63u) sWave[wave] = incl;
        __syncthreads();
        uint64_t before = 0, all = 0;
        for (uint32_t w = 0; w < 16u; w++) { if (w < wave) before += sWave[w]; all += sWave[w]; }
        __syncthreads();
        if (i < nChunks) offs[i] = carry + before + incl - v;
        carry += all;
    }
    if (bad) atomicMax(&sBad, bad);                               // any chunk's failure is the call's
    __syncthreads();
    if (t == 0) { total[0] = carry; total[1] = sBad ? sBad : (carry > dstCap ? BRD_DST_SMALL + 16u : 0u); }
}
extern "C" __global__ void __launch_bounds__(256)
gc_brotli_dec_pack_kernel(const uint8_t* __restrict__ stage, const GcBrDecChunk* __restrict__ chunks, const GcBrDecResult* __restrict__ result, const uint64_t* __restrict__ offs,
                          const uint64_t* __restrict__ total, uint32_t piecesPerChunk, uint8_t* __restrict__ dst)
{
    if (total[1]) return;
    const uint32_t c = blockIdx.x / piecesPerChunk, piece = blockIdx.x % piecesPerChunk;
    const uint64_t n = result[c].size, p0 = (uint64_t)piece << 16;
    if (p0 >= n) return;
    const uint32_t len = (uint32_t)(n - p0 < 65536u ? n - p0 : 65536u);
    const uint8_t* s = stage + chunks[c].stageOff + p0;
    uint8_t* d = dst + offs[c] + p0;
    if ((((uintptr_t)s | (uintptr_t)d) & 15u) == 0u) {
        struct alignas(16) V16 { uint64_t a, b; };
        const uint32_t n16 = len >> 4;
        for (uint32_t i = threadIdx.x; i < n16; i += 256u) ((V16*)d)[i] = ((const V16*)s)[i];
        for (uint32_t i = (n16 << 4) + threadIdx.x; i < len; i += 256u) d[i] = s[i];
    } else for (uint32_t i = threadIdx.x; i < len; i += 256u) d[i] = s[i];
}

// ------------------------------------------------------------------------------------------------ host side
// Walks the brotli-mt frames of a buffer (brotli-mt_decompress.c:191-288).  chunks may be null (count only).  *consumed = bytes of the whole frames.
extern "C" int gc_brotli_scan_prefix(const void* src, size_t n, gc_brotli_chunk* chunks, size_t maxChunks, size_t* nChunks, uint64_t* capacityTotal, size_t* consumed)
{
    if ((!src && n) || !nChunks) return GC_ERR_PARAM;
    const uint8_t* p = (const uint8_t*)src;
    size_t off = 0, k = 0; uint64_t cap = 0;
    while (off + 16u <= n) {
        uint32_t magic, eight, csize; uint16_t br, hint;
        memcpy(&magic, p + off, 4); memcpy(&eight, p + off + 4, 4); memcpy(&csize, p + off + 8, 4); memcpy(&br, p + off + 12, 2); memcpy(&hint, p + off + 14, 2);
        if (magic != 0x184D2A50u || eight != 8u || br != 0x5242u) return GC_ERR_CORRUPT;
        if (off + 16u + (size_t)csize > n) break;                 // the frame is not whole yet
        if (chunks) { if (k >= maxChunks) return GC_ERR_PARAM; chunks[k].src_off = off + 16u; chunks[k].src_size = csize; chunks[k].capacity = (uint32_t)hint << 16; }
        cap += (uint64_t)hint << 16; k++;
        off += 16u + (size_t)csize;
    }
    *nChunks = k;
    if (capacityTotal) *capacityTotal = cap;
    if (consumed) *consumed = off;
    return GC_OK;
}

// the process's copy of the static dictionary (RFC 7932 Appendix A); every device uploads it on its first use
static uint8_t* gBrDict = nullptr;
static uint64_t gBrDictStamp = 0;
static uint32_t brd_crc32(const uint8_t* p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) { c ^= p[i]; for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); }
    return ~c;
}
extern "C" int gc_brotli_dec_set_dictionary(const void* data, size_t n)
{
    if (!data && n == 0) { free(gBrDict); gBrDict = nullptr; gBrDictStamp++; return GC_OK; }       // (forget it: tests)
    if (!data || n != 122784u || brd_crc32((const uint8_t*)data, n) != 0x5136CB04u) return GC_ERR_PARAM;   // the CRC-32 RFC 7932 Appendix A states
    uint8_t* copy = (uint8_t*)malloc(n);
    if (!copy) return GC_ERR_NOMEM;
    memcpy(copy, data, n);
    free(gBrDict); gBrDict = copy; gBrDictStamp++;
    return GC_OK;
}
extern "C" int gc_brotli_dec_has_dictionary(void) { return gBrDict != nullptr; }

static bool brd_grow(uint8_t** p, size_t* cap, size_t need)
{
    if (need <= *cap) return true;
    if (*p) hipFree(*p);
    *p = nullptr; *cap = 0;
    if (hipMalloc((void**)p, need + 64) != hipSuccess) return false;
    *cap = need;
    return true;
}
void gc_brd_release(GcBrDecWork* w)
{
    if (w->stage) hipFree(w->stage);
    if (w->pages) hipFree(w->pages);
    if (w->meta) hipFree(w->meta);
    if (w->dict) hipFree(w->dict);
    if (w->ev0) hipEventDestroy((hipEvent_t)w->ev0);
    if (w->ev1) hipEventDestroy((hipEvent_t)w->ev1);
    memset(w, 0, sizeof(*w));
}
// d_src / d_dst: device memory; chunks: host memory (as the scan returned them).  Synchronous (the sizes are read back).
int gc_brd_decode(hipStream_t st, GcBrDecWork* w, const uint8_t* d_src, const gc_brotli_chunk* chunks, size_t nChunks, uint8_t* d_dst, size_t dstCap, size_t* produced, char* err, size_t errCap)
{
    *produced = 0;
    if (nChunks == 0) return GC_OK;
    if (nChunks > 0x7FFFFFFFu / 16u) return GC_ERR_PARAM;
    GcBrDecChunk* hc = (GcBrDecChunk*)malloc(nChunks * sizeof(GcBrDecChunk));
    if (!hc) return GC_ERR_NOMEM;
    uint64_t stageBytes = 0; uint32_t maxHint = 0;
    for (size_t i = 0; i < nChunks; i++) {
        hc[i].srcOff = chunks[i].src_off; hc[i].srcSize = chunks[i].src_size; hc[i].hintBytes = chunks[i].capacity; hc[i].stageOff = stageBytes;
        stageBytes += ((uint64_t)chunks[i].capacity + 63u) & ~63ull;
        if (chunks[i].capacity > maxHint) maxHint = chunks[i].capacity;
    }
    const size_t oChunks = 0, oRes = (nChunks * sizeof(GcBrDecChunk) + 63u) & ~(size_t)63u, oOffs = oRes + ((nChunks * sizeof(GcBrDecResult) + 63u) & ~(size_t)63u),
                 oTot = oOffs + ((nChunks * 8u + 63u) & ~(size_t)63u), oCur = oTot + 64u, metaBytes = oCur + 64u;
    int rc = GC_OK;
    if (!brd_grow(&w->stage, &w->stageCap, (size_t)stageBytes + 64u) || !brd_grow(&w->meta, &w->metaCap, metaBytes)) rc = GC_ERR_NOMEM;
    if (rc == GC_OK && gBrDict && (!w->dict || w->dictStamp != gBrDictStamp)) {
        if (!w->dict && hipMalloc((void**)&w->dict, 122784u + 64u) != hipSuccess) rc = GC_ERR_NOMEM;
        else if (hipMemcpyAsync(w->dict, gBrDict, 122784u, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = GC_ERR_HIP;
        w->dictStamp = gBrDictStamp;
    }
    if (rc == GC_OK && !w->ev0 && (hipEventCreate((hipEvent_t*)&w->ev0) != hipSuccess || hipEventCreate((hipEvent_t*)&w->ev1) != hipSuccess)) rc = GC_ERR_HIP;
    uint64_t tot[2] = { 0, 0 };
    for (int attempt = 0; attempt < 2 && rc == GC_OK; attempt++) {
        // pages: a chunk whose meta-block does not fit LDS takes one; few streams need any (the reference's qualities 10-11), so the pool starts small
        const uint32_t wantPages = attempt == 0 ? (uint32_t)(nChunks < 64u ? nChunks : 64u) : (uint32_t)nChunks;
        if (w->nPages < wantPages) {
            if (w->pages) hipFree(w->pages);
            w->pages = nullptr; w->nPages = 0;
            if (hipMalloc((void**)&w->pages, (size_t)wantPages * BRD_PAGE) != hipSuccess) { rc = GC_ERR_NOMEM; break; }
            w->nPages = wantPages;
        }
        if (hipMemcpyAsync(w->meta + oChunks, hc, nChunks * sizeof(GcBrDecChunk), hipMemcpyHostToDevice, st) != hipSuccess || hipMemsetAsync(w->meta + oCur, 0, 64, st) != hipSuccess) { rc = GC_ERR_HIP; break; }
        GcBrDict dict; dict.words = gBrDict ? w->dict : nullptr;
        const uint32_t grid = (uint32_t)(nChunks < BRD_MAX_WAVES ? nChunks : BRD_MAX_WAVES);
        hipEventRecord((hipEvent_t)w->ev0, st);
        const GcBrDecChunk* dc = (const GcBrDecChunk*)(w->meta + oChunks); uint32_t* cur = (uint32_t*)(w->meta + oCur); GcBrDecResult* res = (GcBrDecResult*)(w->meta + oRes);
        const uint32_t inst = w->instance ? w->instance : (nChunks <= 256u ? 1u : (nChunks <= 512u ? 2u : (nChunks <= 1024u ? 3u : 4u)));
        const uint32_t lc = w->ldsCap ? w->ldsCap : ~0u;
        if (inst == 1u) GC_LAUNCH(gc_brotli_dec_kernel_a, grid, 64, st, d_src, dc, (uint32_t)nChunks, w->stage, w->pages, w->nPages, cur, res, dict, lc);
        else if (inst == 2u) GC_LAUNCH(gc_brotli_dec_kernel_b, grid, 64, st, d_src, dc, (uint32_t)nChunks, w->stage, w->pages, w->nPages, cur, res, dict, lc);
        else if (inst == 3u) GC_LAUNCH(gc_brotli_dec_kernel_c, grid, 64, st, d_src, dc, (uint32_t)nChunks, w->stage, w->pages, w->nPages, cur, res, dict, lc);
        else GC_LAUNCH(gc_brotli_dec_kernel_d, grid, 64, st, d_src, dc, (uint32_t)nChunks, w->stage, w->pages, w->nPages, cur, res, dict, lc);
        GC_LAUNCH(gc_brotli_dec_plan_kernel, 1, 1024, st, (const GcBrDecResult*)(w->meta + oRes), (uint32_t)nChunks, (uint64_t)dstCap, (uint64_t*)(w->meta + oOffs), (uint64_t*)(w->meta + oTot));
        const uint32_t pieces = (maxHint + 65535u) >> 16;
        if (pieces) GC_LAUNCH(gc_brotli_dec_pack_kernel, (uint32_t)nChunks * pieces, 256, st, w->stage, (const GcBrDecChunk*)(w->meta + oChunks), (const GcBrDecResult*)(w->meta + oRes),
                              (const uint64_t*)(w->meta + oOffs), (const uint64_t*)(w->meta + oTot), pieces, d_dst);
        hipEventRecord((hipEvent_t)w->ev1, st);
        if (hipMemcpyAsync(tot, w->meta + oTot, 16, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = GC_ERR_HIP; break; }
        hipEventElapsedTime(&w->ms, (hipEvent_t)w->ev0, (hipEvent_t)w->ev1);
        if (tot[1] != BRD_LIMIT || w->nPages >= nChunks) break;   // (a second round with a page for every chunk)
    }
    free(hc);
    if (rc != GC_OK) return rc;
    switch (tot[1]) {
        case 0: *produced = (size_t)tot[0]; return GC_OK;
        case BRD_DST_SMALL + 16u: if (err) snprintf(err, errCap, "destination too small: need %llu bytes", (unsigned long long)tot[0]); return GC_ERR_DST_SMALL;
        case BRD_DICTIONARY: if (err) snprintf(err, errCap, "the brotli stream refers to the static dictionary of RFC 7932 and the host has not handed it over (gc_brotli_dec_set_dictionary)"); return GC_ERR_UNSUPPORTED;
        case BRD_LIMIT: if (err) snprintf(err, errCap, "a meta-block of the brotli stream holds more prefix codes than this decoder's arenas take"); return GC_ERR_UNSUPPORTED;
        default: if (err) snprintf(err, errCap, "damaged brotli stream (chunk status %u)", (unsigned)tot[1]); return GC_ERR_CORRUPT;   // (a chunk that outgrows its brotli-mt hint is one, brotli-mt_decompress.c:243)
    }
}
