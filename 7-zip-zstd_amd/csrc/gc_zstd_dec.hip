// gc_zstd_dec.hip -- zstd frame decoder on the device (the decoding half of the ZSTD method, what
// NCompress::NZSTD::CDecoder::CodeSpec does with ZSTD_decompressStream, CPP/7zip/Compress/ZstdDecoder.cpp:66-240).
//
// Four kernels per batch of frames:
//
//   index     one thread per frame walks its block headers and, for compressed blocks, the literals header and the sequence count:
//             the block table (GcZdBlock) with each block's place in the literal / sequence workspaces
//   literals  \  one wave per BLOCK each, two kernels on two streams -- blocks are entropy-coded independently of what they decode to,
//   sequences /  so a stream of one frame spreads over the chip just like a stream of a thousand:
//             literals  Huffman tree (direct or FSE-coded weights) -> 2^11-entry table in LDS; the 1 or 4 streams are decoded
//                       by 64 lanes at once: every lane starts at a guessed bit position inside its stream, Huffman codes
//                       re-synchronise after a few symbols, each lane then restarts where its predecessor really ended until
//                       nothing moves any more (usually one round), a prefix sum of the symbol counts places the output
//             sequences the three FSE tables (predefined / RLE / described / repeated from an earlier block: found by walking the
//                       block table backwards), then lane 0 decodes the sequence bitstream -- three interleaved FSE states, one
//                       serial chain by format -- from 4 KiB pieces the whole wave stages in LDS, and writes one 16-byte record
//                       per sequence.  Repeat offsets that reach into the history in front of the block stay symbolic
//                       (GC_ZD_SYM: "incoming offset k minus delta").
//   execute   one workgroup per FRAME, blocks in order, the block image in LDS:
//               pass 1  all threads, one sequence each: symbolic offsets get their values, literals -> image, and every match whose
//                       source lies in front of the block (copied from the frame's output in HBM)
//               pass 2  wave 0, sequence by sequence, 64 bytes per step: the matches that read the block itself
//                       (offset < 64: dst[k] = src[k mod offset], so overlapping copies are parallel as well)
//               flush   image -> HBM; XXH64 of the content (4 lanes = the 4 accumulators) when the frame carries a checksum
//
// Restated from the reference decoder (the format is normative, every rule has to match):
//   frame header, block headers        ZSTD_getFrameHeader_advanced zstd_decompress.c:447, ZSTD_decompressFrame :953, ZSTD_findFrameSizeInfo :734
//   literals section                   ZSTD_decodeLiteralsBlock zstd_decompress_block.c:134-340
//   Huffman weights / table / streams  HUF_readStats entropy_common.c:234, HUF_readDTableX1_wksp huf_decompress.c:385, HUF_decompress4X1 :602
//   NCount, FSE decoding tables        FSE_readNCount_body entropy_common.c:42, ZSTD_buildFSETable_body zstd_decompress_block.c:485
//   sequences header / decode / exec   ZSTD_decodeSeqHeaders :695, ZSTD_decodeSequence :1229, ZSTD_execSequence :1001
//   content checksum                   XXH64 (xxhash.h), zstd_decompress.c:1034-1056
// Not supported: dictionaries (a frame with a dictionary id is refused), legacy (v0.x) frames, offsets of 2 GiB and more.
#include "gpucodec.h"
#include "gc_zstd_dec.h"
#include "gc_device.h"
#ifdef HIPEMU
#include "hip_runtime_stub.h"
#else
#include <hip/hip_runtime.h>
#define GC_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif
#include <string.h>

// ---- format constants (RFC 8878 3.1.1.3.2.1.1: symbol -> baseline, extra bits) ----
__constant__ uint32_t kZdLLBase[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,128,256,512,1024,2048,4096,8192,16384,32768,65536 };
__constant__ uint8_t  kZdLLBits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
__constant__ uint32_t kZdMLBase[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                        35,37,39,41,43,47,51,59,67,83,99,131,259,515,1027,2051,4099,8195,16387,32771,65539 };
__constant__ uint8_t  kZdMLBits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
__constant__ int16_t  kZdLLNorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
__constant__ int16_t  kZdMLNorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
__constant__ int16_t  kZdOFNorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

#define ZD_HUF_LOG_MAX 11u
#define ZD_WIN_L 256u             // bytes staged from the start of a block: literals header, tree description (<= 5 + 129)
#define ZD_WIN_S 512u             // bytes staged from the symbol-modes byte on: modes + three NCount headers (well below 200)

struct __attribute__((aligned(8))) GcU2 { uint32_t x, y; };

// which of the three sequence symbol tables
enum { ZT_LL = 0, ZT_OF = 1, ZT_ML = 2 };
struct ZdConst {                  // the tables above, staged in LDS (a lane that walks them one entry at a time should not wait on HBM)
    uint32_t llBase[36], mlBase[53];
    uint8_t llBits[36], mlBits[53];
    int16_t llNorm[36], mlNorm[53], ofNorm[29];
};

// ---- backward bit reader over a stream in LDS (bits are consumed from the end; the highest set bit of the last byte is the end mark) ----
struct ZdBR { const uint8_t* p; int32_t off; };         // off = bits left below the read position; negative after an over-read
__device__ __forceinline__ bool zd_br_init(ZdBR& r, const uint8_t* p, uint32_t n)
{
    if (!n) return false;
    const uint32_t last = p[n - 1u];
    if (!last) return false;
    r.p = p; r.off = (int32_t)((n - 1u) * 8u + gc_hibit32(last));
    return true;
}
// the next nb bits (nb <= 32) as a number whose top bit is the one consumed first; bits in front of the stream read as zeros
__device__ __forceinline__ uint32_t zd_br_peek(const ZdBR& r, uint32_t nb)
{
    const int32_t pos = r.off - (int32_t)nb;
    uint64_t v;
    if (pos >= 0) v = gc_ld64(r.p + ((uint32_t)pos >> 3)) >> ((uint32_t)pos & 7u);
    else v = (pos <= -64) ? 0ull : (gc_ld64(r.p) << (uint32_t)(-pos));
    return (uint32_t)(v & ((1ull << nb) - 1ull));
}
__device__ __forceinline__ uint32_t zd_br_read(ZdBR& r, uint32_t nb) { const uint32_t v = zd_br_peek(r, nb); r.off -= (int32_t)nb; return v; }

// ---- NCount (forward, LSB first) from LDS.  Returns the bytes used, 0 on error. ----
__device__ uint32_t zd_read_ncount(const uint8_t* p, uint32_t n, int16_t* norm, uint32_t maxSymAllowed, uint32_t maxLog, uint32_t* maxSymOut, uint32_t* logOut)
{
    if (n < 1u) return 0;
    uint32_t bp = 0;
#define ZD_FW(nb) ((uint32_t)(gc_ld64(p + (bp >> 3)) >> (bp & 7u)) & ((1u << (nb)) - 1u))
    const uint32_t log = ZD_FW(4) + 5u; bp += 4u;
    if (log > maxLog) return 0;
    int32_t remaining = (int32_t)(1u << log) + 1;
    int32_t threshold = (int32_t)(1u << log);
    uint32_t nbBits = log + 1u, sym = 0;
    bool prev0 = false;
    while (remaining > 1 && sym <= maxSymAllowed) {
        if (bp > n * 8u) return 0;
        if (prev0) {
            uint32_t n0 = sym;
            for (;;) {
                if (bp > n * 8u) return 0;
                const uint32_t r = ZD_FW(2); bp += 2u;
                n0 += r;
                if (r != 3u) break;
            }
            if (n0 > maxSymAllowed + 1u) return 0;
            while (sym < n0) norm[sym++] = 0;
            if (sym > maxSymAllowed) break;
        }
        const int32_t mx = (2 * threshold - 1) - remaining;
        int32_t count;
        const int32_t low = (int32_t)ZD_FW(nbBits - 1u);
        if (low < mx) { count = low; bp += nbBits - 1u; }
        else {
            count = (int32_t)ZD_FW(nbBits);
            if (count >= threshold) count -= mx;
            bp += nbBits;
        }
        count--;                                   // -1: "less than one" probability
        remaining -= count < 0 ? -count : count;
        norm[sym++] = (int16_t)count;
        prev0 = count == 0;
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
    }
#undef ZD_FW
    if (remaining != 1 || sym == 0) return 0;
    const uint32_t bytes = (bp + 7u) >> 3;
    if (bytes > n) return 0;
    *maxSymOut = sym - 1u; *logOut = log;
    return bytes;
}

// ---- FSE decoding table of the Huffman weights.  First the symbol of every cell (tab[cell].x), then tab[state] = { newStateBase | nbBits << 16, symbol } ----
__device__ bool zd_fse_build(GcU2* tab, const int16_t* norm, uint32_t maxSym, uint32_t log, uint16_t* symNext)
{
    const uint32_t size = 1u << log, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    uint32_t high = size - 1u;
    for (uint32_t s = 0; s <= maxSym; s++) {
        if (norm[s] == -1) { tab[high--].x = s; symNext[s] = 1; }
        else symNext[s] = (uint16_t)norm[s];
    }
    uint32_t pos = 0;
    for (uint32_t s = 0; s <= maxSym; s++) {
        const int cnt = norm[s];
        for (int i = 0; i < cnt; i++) {
            tab[pos].x = s;
            do pos = (pos + step) & mask; while (pos > high);
        }
    }
    if (pos != 0) return false;
    for (uint32_t u = 0; u < size; u++) {
        const uint32_t s = tab[u].x;
        const uint32_t next = symNext[s]++;
        const uint32_t nb = log - gc_hibit32(next);
        GcU2 e; e.x = (((next << nb) - size) & 0xFFFFu) | (nb << 16); e.y = s;
        tab[u] = e;
    }
    return true;
}

// The same for the three symbol tables of the sequences section, 4 bytes per cell:
//   next-state base (10 bits) | state bits << 10 (4) | extra bits of the symbol << 14 (5) | symbol << 19 (6)
// (the baseline of a literal-length / match-length symbol comes from a 36- / 53-entry table, the offset's is 1 << symbol)
#define ZD_SEQ_ENTRY(base, nb, extra, sym) ((base) | ((nb) << 10) | ((extra) << 14) | ((sym) << 19))
__device__ bool zd_seq_fse_build(uint32_t* tab, const int16_t* norm, uint32_t maxSym, uint32_t log, uint16_t* symNext, const ZdConst* k, int which)
{
    const uint32_t size = 1u << log, mask = size - 1u, step = (size >> 1) + (size >> 3) + 3u;
    uint32_t high = size - 1u;
    for (uint32_t s = 0; s <= maxSym; s++) {
        if (norm[s] == -1) { tab[high--] = s; symNext[s] = 1; }
        else symNext[s] = (uint16_t)norm[s];
    }
    uint32_t pos = 0;
    for (uint32_t s = 0; s <= maxSym; s++) {
        const int cnt = norm[s];
        for (int i = 0; i < cnt; i++) {
            tab[pos] = s;
            do pos = (pos + step) & mask; while (pos > high);
        }
    }
    if (pos != 0) return false;
    for (uint32_t u = 0; u < size; u++) {
        const uint32_t s = tab[u];
        const uint32_t next = symNext[s]++;
        const uint32_t nb = log - gc_hibit32(next);
        const uint32_t extra = which == ZT_LL ? k->llBits[s] : (which == ZT_ML ? k->mlBits[s] : s);
        tab[u] = ZD_SEQ_ENTRY(((next << nb) - size) & 0x3FFu, nb, extra, s);
    }
    return true;
}

// ---- Huffman tree description (in LDS) -> sHuf[2^log] = symbol | nbBits << 8.  Returns the bytes used, 0 on error. ----
__device__ uint32_t zd_huf_read(const uint8_t* p, uint32_t n, uint16_t* sHuf, uint8_t* sW, int16_t* sNorm, uint16_t* sNext, GcU2* sFseW, uint32_t* logOut)
{
    if (n < 1u) return 0;
    const uint32_t hb = p[0];
    uint32_t nw, used;
    if (hb >= 128u) {                                    // 4-bit weights, high nibble first
        nw = hb - 127u; used = 1u + ((nw + 1u) >> 1);
        if (used > n) return 0;
        for (uint32_t i = 0; i < nw; i++) { const uint32_t b = p[1u + (i >> 1)]; sW[i] = (uint8_t)((i & 1u) ? (b & 15u) : (b >> 4)); }
    } else {                                             // FSE-coded weights, two interleaved states (FSE_decompress_usingDTable_generic)
        used = 1u + hb;
        if (hb < 2u || used > n) return 0;
        uint32_t maxSym = 0, log = 0;
        const uint32_t h = zd_read_ncount(p + 1u, hb, sNorm, 255u, 6u, &maxSym, &log);
        if (!h || h >= hb) return 0;
        if (!zd_fse_build(sFseW, sNorm, maxSym, log, sNext)) return 0;
        ZdBR r;
        if (!zd_br_init(r, p + 1u + h, hb - h)) return 0;
        uint32_t s1 = zd_br_read(r, log), s2 = zd_br_read(r, log);
        if (r.off < 0) return 0;
        nw = 0;
        for (;;) {
            if (nw > 253u) return 0;
            GcU2 e = sFseW[s1];
            sW[nw++] = (uint8_t)e.y;
            s1 = (e.x & 0xFFFFu) + zd_br_read(r, (e.x >> 16) & 0xFFu);
            if (r.off < 0) { sW[nw++] = (uint8_t)sFseW[s2].y; break; }
            if (nw > 253u) return 0;
            e = sFseW[s2];
            sW[nw++] = (uint8_t)e.y;
            s2 = (e.x & 0xFFFFu) + zd_br_read(r, (e.x >> 16) & 0xFFu);
            if (r.off < 0) { sW[nw++] = (uint8_t)sFseW[s1].y; break; }
        }
    }
    uint32_t total = 0, rank1 = 0;
    for (uint32_t i = 0; i < nw; i++) {
        const uint32_t w = sW[i];
        if (w > 12u) return 0;
        if (w) total += 1u << (w - 1u);
        rank1 += w == 1u;
    }
    if (!total) return 0;
    const uint32_t log = gc_hibit32(total) + 1u;
    if (log > ZD_HUF_LOG_MAX) return 0;
    const uint32_t rest = (1u << log) - total;
    if (rest & (rest - 1u)) return 0;                    // the implied last weight completes a power of two
    const uint32_t lastW = gc_hibit32(rest) + 1u;
    sW[nw] = (uint8_t)lastW; rank1 += lastW == 1u;
    if (rank1 < 2u || (rank1 & 1u)) return 0;            // HUF_readStats: by construction at least 2 symbols of weight 1, an even number
    const uint32_t nSym = nw + 1u;
    uint32_t pos = 0;
    for (uint32_t w = 1; w <= log; w++) {
        const uint32_t len = 1u << (w - 1u), nb = log + 1u - w;
        for (uint32_t s = 0; s < nSym; s++) {
            if (sW[s] != w) continue;
            const uint16_t e = (uint16_t)(s | (nb << 8));
            for (uint32_t i = 0; i < len; i++) sHuf[pos + i] = e;
            pos += len;
        }
    }
    *logOut = log;
    return used;
}

// ---- backward bit reader over a stream in HBM with a 64-bit register window ----
struct ZdGW { const uint8_t* p; uint64_t w; int32_t wb; };          // w = stream bits [wb, wb + 64)
__device__ __forceinline__ void zd_gw_fill(ZdGW& g, int32_t off)      // afterwards wb + 56 < off <= wb + 64 (for off > 0)
{
    const int32_t b = ((off + 7) >> 3) - 8;
    g.wb = b * 8;
    if (b >= 0) g.w = gc_ld64(g.p + b);
    else {
        uint64_t v = 0;
        for (int32_t i = 0; i < b + 8; i++) v |= (uint64_t)g.p[i] << (8 * i);
        g.w = b <= -8 ? 0ull : v << (uint32_t)(-b * 8);
    }
}
__device__ __forceinline__ uint32_t zd_gw_peek(ZdGW& g, int32_t off, uint32_t nb)       // nb <= 11
{
    if (off - (int32_t)nb < g.wb || off > g.wb + 64) zd_gw_fill(g, off);
    const int32_t sh = off - (int32_t)nb - g.wb;
    const uint64_t v = sh >= 0 ? g.w >> (uint32_t)sh : g.w << (uint32_t)(-sh);
    return (uint32_t)v & ((1u << nb) - 1u);
}

// Source of a repeated table: the nearest earlier block of the frame (compressed, with sequences) whose mode for this table is not
// "repeat".  Returns the block index or 0xFFFFFFFF.
__device__ uint32_t zd_find_table_source(const GcZdBlock* blocks, uint32_t first, uint32_t b, int which)
{
    const uint32_t sh = which == ZT_LL ? 6u : (which == ZT_OF ? 4u : 2u);
    for (uint32_t k = b; k > first; ) {
        k--;
        const uint32_t ty = blocks[k].type;
        if ((ty & 3u) != 2u || (ty & GC_ZD_B_BAD) || !blocks[k].nSeq) continue;
        if (((blocks[k].modes >> sh) & 3u) != 3u) return k;
    }
    return 0xFFFFFFFFu;
}

// One symbol table of the sequences section from the window `win` (modes byte at win[0]), p = read position inside the window.
// Returns the new read position, 0 on error.  mode 3 is resolved by the caller.
__device__ uint32_t zd_seq_table(uint32_t mode, const uint8_t* win, uint32_t winLen, uint32_t p, uint32_t* tab, uint32_t* logOut, int which,
                                 const ZdConst* k, int16_t* sNorm, uint16_t* sNext)
{
    const uint32_t maxSym = which == ZT_LL ? 35u : (which == ZT_OF ? 31u : 52u), maxLog = which == ZT_OF ? 8u : 9u;
    if (mode == 0u) {
        const int16_t* dn = which == ZT_LL ? k->llNorm : (which == ZT_OF ? k->ofNorm : k->mlNorm);
        const uint32_t dm = which == ZT_LL ? 35u : (which == ZT_OF ? 28u : 52u), dl = which == ZT_OF ? 5u : 6u;
        for (uint32_t s = 0; s <= dm; s++) sNorm[s] = dn[s];
        if (!zd_seq_fse_build(tab, sNorm, dm, dl, sNext, k, which)) return 0;
        *logOut = dl; return p;
    }
    if (mode == 1u) {
        if (p >= winLen) return 0;
        const uint32_t s = win[p];
        if (s > maxSym) return 0;
        const uint32_t extra = which == ZT_LL ? k->llBits[s] : (which == ZT_ML ? k->mlBits[s] : s);
        tab[0] = ZD_SEQ_ENTRY(0u, 0u, extra, s); *logOut = 0; return p + 1u;
    }
    uint32_t ms = 0, log = 0;
    if (p >= winLen) return 0;
    const uint32_t h = zd_read_ncount(win + p, winLen - p, sNorm, maxSym, maxLog, &ms, &log);
    if (!h) return 0;
    if (!zd_seq_fse_build(tab, sNorm, ms, log, sNext, k, which)) return 0;
    *logOut = log; return p + h;
}

// bytes of table `which` in a window (to skip it): 0 on error for mode 2
__device__ int zd_seq_table_skip(uint32_t mode, const uint8_t* win, uint32_t winLen, uint32_t p, int which, int16_t* sNorm)
{
    if (mode == 0u || mode == 3u) return 0;
    if (mode == 1u) return 1;
    const uint32_t maxSym = which == ZT_LL ? 35u : (which == ZT_OF ? 31u : 52u), maxLog = which == ZT_OF ? 8u : 9u;
    uint32_t ms = 0, log = 0;
    if (p >= winLen) return -1;
    const uint32_t h = zd_read_ncount(win + p, winLen - p, sNorm, maxSym, maxLog, &ms, &log);
    return h ? (int)h : -1;
}

// lane 0 only: n bytes from HBM (8 at a time, bounded by the end of the compressed stream) into LDS
__device__ void zd_load_window(uint8_t* win, const uint8_t* src, uint64_t srcSize, uint64_t from, uint32_t n)
{
    for (uint32_t i = 0; i < n; i += 8u) {
        uint64_t v = 0;
        if (from + i + 8u <= srcSize) v = gc_ld64(src + from + i);
        else for (uint32_t k = 0; k < 8u && from + i + k < srcSize; k++) v |= (uint64_t)src[from + i + k] << (8u * k);
        __builtin_memcpy(win + i, &v, 8);
    }
}

// =============================================================== index kernel ===============================================================
extern "C" __global__ void __launch_bounds__(64)
gc_zstd_dec_index_kernel(const uint8_t* __restrict__ src, const GcZdFrame* __restrict__ frames, uint32_t nFrames, GcZdBlock* blocks, uint64_t* frameTot)
{
    const uint32_t f = blockIdx.x * 64u + threadIdx.x;
    if (f >= nFrames) return;
    const GcZdFrame fr = frames[f];
    const uint8_t* const fsrc = src + fr.srcOff;
    const uint64_t srcEnd = fr.srcSize - ((fr.flags & GC_ZD_F_CHECKSUM) ? 4u : 0u);
    uint64_t ip = fr.hdrSize, lit = 0, seq = 0;
    bool broken = false;
    for (uint32_t k = 0; k < fr.nBlocks; k++) {
        GcZdBlock e;
        e.srcOff = 0; e.litOff = lit; e.seqOff = seq; e.size = 0; e.type = GC_ZD_B_BAD | GC_ZD_B_LAST; e.regen = 0; e.litInfo = 0; e.comp = 0; e.nSeq = 0; e.seqPos = 0; e.modes = 0;
        e.frame = f; e.status = GC_ZD_CORRUPT; e.outSize = 0; e.lposEnd = 0; e.dposEnd = 0; e.rep[0] = e.rep[1] = e.rep[2] = 0; e.litStatus = GC_ZD_OK;
        if (!broken && ip + 3u <= srcEnd) {
            const uint32_t h = (uint32_t)fsrc[ip] | ((uint32_t)fsrc[ip + 1] << 8) | ((uint32_t)fsrc[ip + 2] << 16);
            const uint32_t bt = (h >> 1) & 3u, bs = h >> 3, last = h & 1u;
            const uint64_t payload = bt == 1u ? 1u : bs;
            if (bt == 3u || bs > GC_ZSTD_BLOCK_MAX || ip + 3u + payload > srcEnd || (last != 0u) != (k + 1u == fr.nBlocks)) broken = true;
            else {
                e.srcOff = fr.srcOff + ip + 3u; e.size = (uint32_t)payload; e.type = bt | (last ? GC_ZD_B_LAST : 0u); e.status = GC_ZD_OK;
                if (bt < 2u) e.regen = bs;
                else {
                    const uint8_t* const b = fsrc + ip + 3u;
                    bool bad = bs < 2u;
                    uint32_t hdr = 0, regen = 0, comp = 0, nStreams = 1, lt = 0;
                    if (!bad) {
                        const uint32_t b0 = b[0], sf = (b0 >> 2) & 3u;
                        lt = b0 & 3u;
                        if (lt < 2u) {
                            if (!(sf & 1u)) { hdr = 1; regen = b0 >> 3; }
                            else if (sf == 1u) { hdr = 2; regen = (b0 | ((uint32_t)b[1] << 8)) >> 4; }
                            else { hdr = 3; if (bs < 3u) bad = true; else regen = (b0 | ((uint32_t)b[1] << 8) | ((uint32_t)b[2] << 16)) >> 4; }
                            comp = lt == 0u ? regen : 1u;
                        } else {
                            hdr = sf < 2u ? 3u : (sf == 2u ? 4u : 5u);
                            if (bs < hdr) bad = true;
                            else {
                                uint64_t v = 0;
                                for (uint32_t i = 0; i < hdr; i++) v |= (uint64_t)b[i] << (8u * i);
                                if (sf < 2u) { regen = (uint32_t)(v >> 4) & 0x3FFu; comp = (uint32_t)(v >> 14) & 0x3FFu; nStreams = sf == 0u ? 1u : 4u; }
                                else if (sf == 2u) { regen = (uint32_t)(v >> 4) & 0x3FFFu; comp = (uint32_t)(v >> 18) & 0x3FFFu; nStreams = 4; }
                                else { regen = (uint32_t)(v >> 4) & 0x3FFFFu; comp = (uint32_t)(v >> 22) & 0x3FFFFu; nStreams = 4; }
                                if (regen == 0u) bad = true;
                            }
                        }
                        if (regen > GC_ZSTD_BLOCK_MAX || (uint64_t)hdr + comp >= bs) bad = true;       // at least the sequence count follows
                    }
                    uint32_t nSeq = 0, p = hdr + comp;
                    if (!bad) {
                        const uint32_t b0 = b[p++];
                        if (b0 < 128u) nSeq = b0;
                        else if (b0 < 255u) { if (p >= bs) bad = true; else nSeq = ((b0 - 128u) << 8) + b[p++]; }
                        else { if (p + 2u > bs) bad = true; else { nSeq = (uint32_t)b[p] + ((uint32_t)b[p + 1] << 8) + 0x7F00u; p += 2u; } }
                    }
                    if (!bad) {
                        if (nSeq == 0u) { if (p != bs) bad = true; }
                        else if (p >= bs) bad = true;
                        else { e.modes = b[p]; if (e.modes & 3u) bad = true; }
                    }
                    if (bad) { e.type |= GC_ZD_B_BAD; e.status = GC_ZD_CORRUPT; }
                    else {
                        e.regen = regen; e.litInfo = lt | (nStreams << 2) | (hdr << 8); e.comp = comp; e.nSeq = nSeq; e.seqPos = p;
                        if (lt >= 2u) lit += (regen + 15u) & ~15ull;
                        seq += nSeq;
                    }
                }
                ip += 3u + payload;
            }
        } else broken = true;
        blocks[fr.blockBase + k] = e;
    }
    frameTot[2u * f] = lit + 64u; frameTot[2u * f + 1u] = seq + 1u;
}

// =============================================================== entropy kernel ===============================================================
// literal streams, wave 1.  G lanes per stream; every lane owns a slice of its stream's bit range.
__device__ bool zd_huf_parallel(const uint8_t* sp, uint32_t n, uint32_t count, uint8_t* out, const uint16_t* sHuf, uint32_t log, uint32_t G, uint32_t li, bool valid)
{
    // (all 64 lanes of the wave call this together; lanes of a stream that failed to open keep valid = false and only take part in the shuffles)
    ZdGW g; g.p = sp; g.w = 0; g.wb = 0x40000000;
    int32_t B = 0;
    if (valid) {
        const uint32_t last = n ? sp[n - 1u] : 0u;
        if (!last) valid = false; else B = (int32_t)((n - 1u) * 8u + gc_hibit32(last));
    }
    const int32_t S = valid ? ((B + (int32_t)G - 1) / (int32_t)G < 64 ? 64 : (B + (int32_t)G - 1) / (int32_t)G) : 64;
    int32_t lo = B - (int32_t)(li + 1u) * S; if (lo < 0 || li + 1u == G) lo = 0;
    int32_t start = B - (int32_t)li * S;                 // first guess; exact for lane 0
    int32_t e = start; uint32_t c = 0;
    int32_t chkOff = -1; uint32_t chkIdx = 0;            // a position the previous pass went through, and how many symbols it had decoded by then
    bool first = true;
    for (;;) {
        int32_t prevE = __shfl_up(e, 1u, (int)G);
        if (li == 0u) prevE = B;
        const bool changed = valid && (first || prevE != start);
        if (changed) {
            start = first ? start : prevE;
            int32_t off = start; uint32_t k = 0;
            const int32_t oldChk = chkOff; const uint32_t oldIdx = chkIdx, oldC = c; const int32_t oldE = e;
            chkOff = -1; chkIdx = 0;
            bool joined = false;
            while (off > lo) {
                const uint32_t en = sHuf[zd_gw_peek(g, off, log)];
                off -= (int32_t)(en >> 8); k++;
                if (k == 24u) { chkOff = off; chkIdx = k; }
                if (off == oldChk && off > lo) {         // from here on the previous pass saw the same bits
                    if (chkOff < 0) { chkOff = off; chkIdx = k; }
                    c = k + (oldC - oldIdx); e = oldE; joined = true; break;
                }
            }
            if (!joined) { c = k; e = off; }
        }
        first = false;
        if (!__any(changed)) break;
    }
    // placement: exclusive prefix of the symbol counts inside the group
    uint32_t incl = c;
    for (uint32_t d = 1; d < G; d <<= 1) { const uint32_t v = __shfl_up(incl, d, (int)G); if (li >= d) incl += v; }
    const uint32_t total = __shfl(incl, (int)(G - 1u), (int)G);
    const int32_t lastE = __shfl(e, (int)(G - 1u), (int)G);
    const bool ok = valid && total == count && lastE == 0;
    if (ok) {
        uint8_t* o = out + (incl - c);
        int32_t off = start;
        for (uint32_t k = 0; k < c; k++) {
            const uint32_t en = sHuf[zd_gw_peek(g, off, log)];
            o[k] = (uint8_t)en;
            off -= (int32_t)(en >> 8);
        }
    }
    return ok;
}

// ---- literals: one wave per block ----
extern "C" __global__ void __launch_bounds__(64)
gc_zstd_dec_lit_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, const GcZdFrame* __restrict__ frames, GcZdBlock* blocks, uint8_t* litWork, unsigned long long* prof,
                       const uint32_t* __restrict__ order, uint32_t* ready)
{
    __shared__ uint16_t sHuf[1u << ZD_HUF_LOG_MAX];
    __shared__ GcU2 sFseW[64];
    __shared__ __attribute__((aligned(16))) uint8_t sWinL[ZD_WIN_L + 8u];
    __shared__ int16_t sNormB[256];
    __shared__ uint16_t sNextB[256];
    __shared__ uint8_t sW[256];
    __shared__ uint32_t sV[4];        // 1 error, 2 huf log, 3 tree bytes

    const uint32_t lane = threadIdx.x, b = order[blockIdx.x];            // blocks are taken round by round over the frames (block 0 of every frame first)
    const uint32_t ty = blocks[b].type;
    if ((ty & 3u) != 2u || (ty & GC_ZD_B_BAD) || (blocks[b].litInfo & 3u) < 2u) return;
    const GcZdBlock e = blocks[b];
    const GcZdFrame fr = frames[e.frame];
    const uint8_t* const bsrc = src + e.srcOff;
    const uint32_t bs = e.size;
    for (uint32_t i = lane; i < ZD_WIN_L + 8u; i += 64u) sWinL[i] = i < bs ? bsrc[i] : (uint8_t)0;
    if (lane < 4u) sV[lane] = 0;
    gc_wave_sync();
    const unsigned long long pc0 = prof ? gc_clock() : 0ull;
    unsigned long long pc1 = 0;
    {
    // ------------------------------------------------ literals ------------------------------------------------
    const uint32_t lt = e.litInfo & 3u, nStreams = (e.litInfo >> 2) & 7u, hdr = e.litInfo >> 8;
    if (lt >= 2u) {
        if (lane == 0u) {
            uint32_t log = 0, tree = 0, err = 0;
            if (lt == 2u) {
                const uint32_t avail = e.comp < ZD_WIN_L - hdr ? e.comp : ZD_WIN_L - hdr;
                tree = zd_huf_read(sWinL + hdr, avail, sHuf, sW, sNormB, sNextB, sFseW, &log);
                if (!tree) err = 1;
            } else {                                   // treeless: the tree of the nearest earlier block that carries one
                uint32_t k = b; bool found = false;
                while (k > fr.blockBase) {
                    k--;
                    if ((blocks[k].type & 3u) == 2u && !(blocks[k].type & GC_ZD_B_BAD) && (blocks[k].litInfo & 3u) == 2u) { found = true; break; }
                }
                if (!found) err = 1;
                else {
                    const uint32_t h2 = blocks[k].litInfo >> 8, c2 = blocks[k].comp;
                    zd_load_window(sWinL, src, srcSize, blocks[k].srcOff + h2, ZD_WIN_L);
                    const uint32_t used = zd_huf_read(sWinL, c2 < ZD_WIN_L ? c2 : ZD_WIN_L, sHuf, sW, sNormB, sNextB, sFseW, &log);
                    if (!used) err = 1;
                }
            }
            sV[2] = log; sV[3] = tree; if (err) sV[1] = 1;
        }
        gc_wave_sync();
        pc1 = prof ? gc_clock() : 0ull;
        const uint32_t log = sV[2], tree = sV[3];
        bool ok = sV[1] == 0u;
        const uint32_t base = hdr + tree, avail = ok && e.comp >= tree ? e.comp - tree : 0u;
        const uint32_t G = nStreams == 4u ? 16u : 64u, q = nStreams == 4u ? lane >> 4 : 0u, li = lane & (G - 1u);
        uint32_t sOff = base, sLen = avail, cnt = e.regen, oOff = 0;
        if (nStreams == 4u) {
            if (avail < 10u || e.regen < 6u) ok = false;       // four streams regenerate at least 6 bytes (huf_decompress.c: HUF_decompress4X1_usingDTable_internal_body, dstSize < 6)
            else {
                const uint32_t s1 = (uint32_t)bsrc[base] | ((uint32_t)bsrc[base + 1] << 8), s2 = (uint32_t)bsrc[base + 2] | ((uint32_t)bsrc[base + 3] << 8),
                               s3 = (uint32_t)bsrc[base + 4] | ((uint32_t)bsrc[base + 5] << 8);
                const uint32_t seg = (e.regen + 3u) >> 2;
                if ((uint64_t)6u + s1 + s2 + s3 >= avail || 3u * seg > e.regen) ok = false;
                else {
                    sOff = base + 6u + (q > 0u ? s1 : 0u) + (q > 1u ? s2 : 0u) + (q > 2u ? s3 : 0u);
                    sLen = q == 0u ? s1 : (q == 1u ? s2 : (q == 2u ? s3 : avail - 6u - s1 - s2 - s3));
                    oOff = q * seg; cnt = q == 3u ? e.regen - 3u * seg : seg;
                }
            }
        } else if (!avail) ok = false;
        uint8_t* const out = litWork + fr.litBase + e.litOff + oOff;
        const bool good = zd_huf_parallel(bsrc + sOff, sLen, cnt, out, sHuf, log, G, li, ok);
        if (!good) sV[1] = 1;
    }
    }
    if (prof && lane == 0u) {
        const unsigned long long pc2 = gc_clock();
        atomicAdd(&prof[10], (pc1 ? pc1 : pc2) - pc0); atomicAdd(&prof[11], pc1 ? pc2 - pc1 : 0ull);
    }
    gc_wave_sync();
    if (lane == 0u) blocks[b].litStatus = sV[1] ? GC_ZD_CORRUPT : GC_ZD_OK;
    gc_signal_device(&ready[b]);                                           // the execution kernel may be waiting for this block
}

// ---- sequences: one wave per block (lane 0 decodes; 10 KB of LDS, so that many blocks are in flight per CU) ----
extern "C" __global__ void __launch_bounds__(64)
gc_zstd_dec_seq_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, const GcZdFrame* __restrict__ frames, GcZdBlock* blocks, GcU4* seqWork, unsigned long long* prof,
                       const uint32_t* __restrict__ order, uint32_t* ready)
{
    __shared__ uint32_t sLL[512], sML[512], sOF[256];
    __shared__ __attribute__((aligned(16))) uint8_t sBuf[(GC_ZD_CHUNK + 64u) > 2u * (ZD_WIN_S + 16u) ? (GC_ZD_CHUNK + 64u) : 2u * (ZD_WIN_S + 16u)];
    uint8_t* const sWinS = sBuf;                               // the two header windows are only needed while the tables are built:
    uint8_t* const sWinR = sBuf + ZD_WIN_S + 16u;              // they share their LDS with the bitstream pieces staged afterwards
    __shared__ int16_t sNormA[64];
    __shared__ uint16_t sNextA[64];
    __shared__ ZdConst sK;
    __shared__ uint32_t sV[16];       // 0 error, 4 seq stream start, 5..7 table logs, 8.. results

    const uint32_t lane = threadIdx.x, t = lane, b = order[blockIdx.x];
    const uint32_t ty = blocks[b].type;
    if ((ty & 3u) != 2u || (ty & GC_ZD_B_BAD)) return;
    const GcZdBlock e = blocks[b];
    const GcZdFrame fr = frames[e.frame];
    const uint8_t* const bsrc = src + e.srcOff;
    const uint32_t bs = e.size;
    for (uint32_t i = t; i < 36u; i += 64u) { sK.llBase[i] = kZdLLBase[i]; sK.llBits[i] = kZdLLBits[i]; sK.llNorm[i] = kZdLLNorm[i]; }
    for (uint32_t i = t; i < 53u; i += 64u) { sK.mlBase[i] = kZdMLBase[i]; sK.mlBits[i] = kZdMLBits[i]; sK.mlNorm[i] = kZdMLNorm[i]; }
    for (uint32_t i = t; i < 29u; i += 64u) sK.ofNorm[i] = kZdOFNorm[i];
    for (uint32_t i = t; i < ZD_WIN_S + 8u; i += 64u) sWinS[i] = e.seqPos + i < bs ? bsrc[e.seqPos + i] : (uint8_t)0;
    if (t < 16u) sV[t] = 0;
    gc_wave_sync();
    const unsigned long long pc0 = prof ? gc_clock() : 0ull;
    unsigned long long pc1 = 0;
    {
    // ------------------------------------------------ sequences ------------------------------------------------
    const uint32_t nSeq = e.nSeq;
    uint32_t err = 0, dpos = 0, lpos = 0;
    uint32_t rep0 = GC_ZD_SYM | 0u, rep1 = GC_ZD_SYM | 1u, rep2 = GC_ZD_SYM | 2u;
    if (nSeq) {
        if (lane == 0u) {
            uint32_t p = 1;                                // behind the modes byte
            const int order[3] = { ZT_LL, ZT_OF, ZT_ML };
            for (int i = 0; i < 3 && !err; i++) {
                const int which = order[i];
                uint32_t* const tab = which == ZT_LL ? sLL : (which == ZT_OF ? sOF : sML);
                const uint32_t mode = (e.modes >> (which == ZT_LL ? 6u : (which == ZT_OF ? 4u : 2u))) & 3u;
                if (mode != 3u) { p = zd_seq_table(mode, sWinS, ZD_WIN_S, p, tab, &sV[5 + which], which, &sK, sNormA, sNextA); if (!p) err = 1; }
                else {
                    const uint32_t k = zd_find_table_source(blocks, fr.blockBase, b, which);
                    if (k == 0xFFFFFFFFu) { err = 1; break; }
                    const uint32_t m2 = blocks[k].modes;
                    const uint32_t mk = (m2 >> (which == ZT_LL ? 6u : (which == ZT_OF ? 4u : 2u))) & 3u;
                    uint32_t q = 1;
                    if (mk != 0u) {
                        zd_load_window(sWinR, src, srcSize, blocks[k].srcOff + blocks[k].seqPos, ZD_WIN_S);
                        for (int y = 0; y < 3 && order[y] != which; y++) {
                            const int w2 = order[y];
                            const int sk = zd_seq_table_skip((m2 >> (w2 == ZT_LL ? 6u : (w2 == ZT_OF ? 4u : 2u))) & 3u, sWinR, ZD_WIN_S, q, w2, sNormA);
                            if (sk < 0) { err = 1; break; }
                            q += (uint32_t)sk;
                        }
                    }
                    if (!err && !zd_seq_table(mk, sWinR, ZD_WIN_S, q, tab, &sV[5 + which], which, &sK, sNormA, sNextA)) err = 1;
                }
            }
            if (!err && (p > ZD_WIN_S || e.seqPos + p >= bs)) err = 1;
            sV[4] = e.seqPos + p; if (err) sV[0] = 1;
        }
        gc_wave_sync();
        pc1 = prof ? gc_clock() : 0ull;
        err = sV[0];
        const uint32_t seqStart = sV[4];
        const uint8_t* const sp = bsrc + seqStart;
        const uint32_t n = err ? 1u : bs - seqStart;
        int32_t off = 0;
        if (!err) {
            const uint32_t last = sp[n - 1u];
            if (!last) err = 1; else off = (int32_t)((n - 1u) * 8u + gc_hibit32(last));
        }
        const uint32_t llLog = sV[5 + ZT_LL], ofLog = sV[5 + ZT_OF], mlLog = sV[5 + ZT_ML];
        uint32_t st = 0, j = 0;                               // st: the FSE state of my table (lanes 0..2)
        GcU4* const seq = seqWork + fr.seqBase + e.seqOff;
        bool first = true;
        while (!err) {
            // stage the piece of the bitstream the next sequences read: bytes [cLo, hi + 8) of the stream at sBuf + 16.  With 16 zero bytes in
            // front of byte 0 a read that reaches below the start of the stream (legal at the very end: the bits there read as zeros; an
            // over-read otherwise, caught right after) needs no special case: bit p lies in byte ((p + 128) >> 3) - cLo of sBuf.
            const uint32_t hi = (uint32_t)(off + 7) >> 3;
            const uint32_t cLo = hi > GC_ZD_CHUNK ? (hi - GC_ZD_CHUNK) & ~7u : 0u;
            if (lane < 2u) { const uint64_t z = 0; __builtin_memcpy(sBuf + 8u * lane, &z, 8); }
            for (uint32_t i = lane * 8u; i < hi + 8u - cLo; i += 512u) {
                uint64_t v = 0;
                if (cLo + i + 8u <= n) v = gc_ld64(sp + cLo + i);
                else for (uint32_t k = 0; k < 8u && cLo + i + k < n; k++) v |= (uint64_t)sp[cLo + i + k] << (8u * k);
                __builtin_memcpy(sBuf + 16u + i, &v, 8);
            }
            gc_wave_sync();
            {
#define ZD_RD64(p) (gc_ld64(sBuf + (((uint32_t)((p) + 128) >> 3) - cLo)) >> ((uint32_t)(p) & 7u))
#define ZD_RD32(p) (gc_ld32(sBuf + (((uint32_t)((p) + 128) >> 3) - cLo)) >> ((uint32_t)(p) & 7u))
                // Three lanes decode: lane 0 owns the literal-length state, lane 1 the match-length state, lane 2 the offset state (the other
                // lanes run along idle).  One table read, one read of the symbol's extra bits and one of the state bits serve all three; what
                // the three have to know of each other -- the bit counts, then the three values -- travels through v_readlane into scalar
                // registers, where the bit positions, the repeat-offset rules and the checks are computed once per wave.
                const uint32_t* const myTab = lane == 1u ? sML : (lane == 2u ? sOF : sLL);
                const uint32_t* const myBase = lane == 1u ? sK.mlBase : sK.llBase;
                if (first) {
                    const int32_t p1 = off - (int32_t)llLog, p2 = p1 - (int32_t)ofLog, p3 = p2 - (int32_t)mlLog;
                    if (p3 < 0) err = 1;
                    else {
                        const int32_t pm = lane == 0u ? p1 : (lane == 2u ? p2 : p3);
                        const uint32_t lg = lane == 0u ? llLog : (lane == 2u ? ofLog : mlLog);
                        st = lane < 3u ? ZD_RD32(pm) & ((1u << lg) - 1u) : 0u;
                        off = p3;
                    }
                }
                while (!err && j < nSeq && (cLo == 0u || ((uint32_t)off >> 3) >= cLo + 16u)) {
                    const uint32_t en = myTab[st];
                    const uint32_t more = j + 1u < nSeq ? 15u : 0u;              // the last sequence does not move the states on
                    const uint32_t eb = (en >> 14) & 31u, nb = (en >> 10) & more;
                    const uint32_t llb = gc_readlane(eb, 0), mlb = gc_readlane(eb, 1), ofb = gc_readlane(eb, 2);
                    const uint32_t nl = gc_readlane(nb, 0), nm = gc_readlane(nb, 1), no = gc_readlane(nb, 2);
                    const int32_t p1 = off - (int32_t)ofb, p2 = p1 - (int32_t)mlb, p3 = p2 - (int32_t)llb;      // extra bits: offset, match length, literal length
                    const int32_t p4 = p3 - (int32_t)nl, p5 = p4 - (int32_t)nm, p6 = p5 - (int32_t)no;          // state bits: literal length, match length, offset
                    const int32_t pe = lane == 0u ? p3 : (lane == 1u ? p2 : p1), ps = lane == 0u ? p4 : (lane == 1u ? p5 : p6);
                    const uint64_t we = ZD_RD64(pe);
                    const uint32_t ws = ZD_RD32(ps);
                    const uint32_t base = lane == 2u ? 1u << eb : myBase[lane < 2u ? en >> 19 : 0u];
                    const uint32_t val = base + ((uint32_t)we & ((1u << eb) - 1u));
                    st = lane < 3u ? (en & 0x3FFu) + (ws & ((1u << nb) - 1u)) : 0u;
                    const uint32_t ll = gc_readlane(val, 0), ml = gc_readlane(val, 1), ofv = gc_readlane(val, 2);
                    off = p6;
                    // offset value 1..3 = one of the three last offsets (shifted by one when the sequence has no literals; "4" = the first minus 1)
                    const bool isRep = ofv <= 3u;
                    const uint32_t idx = ofv - 1u + (ll == 0u ? 1u : 0u);
                    const uint32_t r0m1 = (rep0 & GC_ZD_SYM) ? rep0 + 4u : rep0 - 1u;
                    const uint32_t o = !isRep ? ofv - 3u : (idx == 0u ? rep0 : (idx == 1u ? rep1 : (idx == 2u ? rep2 : r0m1)));
                    const bool shift3 = !isRep || idx >= 2u, shift2 = isRep && idx == 1u;
                    rep2 = shift3 ? rep1 : rep2;
                    rep1 = (shift3 || shift2) ? rep0 : rep1;
                    rep0 = o;
                    const bool bad = off < 0 || o == 0u || lpos + ll > e.regen || dpos + ll + ml > GC_ZSTD_BLOCK_MAX;
                    if (ofb > 30u) { err = GC_ZD_UNSUPPORTED; break; }
                    if (bad) { err = 1; break; }
                    if (lane == 0u) { GcU4 rec; rec.x = ll | (ml << 18); rec.y = (ml >> 14) | (lpos << 4); rec.z = o; rec.w = dpos; seq[j] = rec; }
                    dpos += ll + ml; lpos += ll; j++;
                }
#undef ZD_RD64
#undef ZD_RD32
            }
            first = false;
            if (j >= nSeq) break;
        }
        if (!err && off != 0) err = 1;
        if (err && lane == 0u) sV[0] = err;
    }
    if (lane == 0u) { sV[8] = dpos; sV[9] = lpos; sV[10] = rep0; sV[11] = rep1; sV[12] = rep2; }
    }
    if (prof && lane == 0u) {
        const unsigned long long pc2 = gc_clock();
        atomicAdd(&prof[8], (pc1 ? pc1 : pc2) - pc0); atomicAdd(&prof[9], pc1 ? pc2 - pc1 : 0ull); atomicAdd(&prof[12], 1ull);
    }
    gc_wave_sync();
    if (t == 0) {
        const uint32_t dpos = sV[8], lpos = sV[9];
        uint32_t status = sV[0] ? (sV[0] == GC_ZD_UNSUPPORTED ? GC_ZD_UNSUPPORTED : GC_ZD_CORRUPT) : GC_ZD_OK;
        const uint32_t outSize = dpos + (e.regen - lpos);
        if (!status && outSize > GC_ZSTD_BLOCK_MAX) status = GC_ZD_CORRUPT;
        blocks[b].status = status; blocks[b].outSize = outSize; blocks[b].lposEnd = lpos; blocks[b].dposEnd = dpos;
        blocks[b].rep[0] = sV[10]; blocks[b].rep[1] = sV[11]; blocks[b].rep[2] = sV[12];
    }
    gc_signal_device(&ready[b]);
}

// ---- sequences, several blocks per wave ----
// The kernel above spends a whole wave on one block and lets three of its 64 lanes decode, the wave's scalar unit doing the bookkeeping: with 22
// waves per CU the kernel is bound by the ISSUE of those instructions (62 vector + 48 scalar per sequence, a vector instruction of a wave64 takes
// four cycles whatever the number of live lanes; measured 2 000 cycles per sequence and wave).  Here a wave takes ZV_G blocks, four lanes each
// (literal-length, match-length, offset state, one spare), and all bookkeeping is per lane: the same instructions now move ZV_G blocks on, and with
// one wave per SIMD the kernel is bound by the latency of the chain of one sequence (state -> table -> bit counts -> bit positions -> bits -> state).
// What the three lanes of a block tell each other travels through quad-permute DPP moves.  The bitstream pieces are still staged in LDS: when one
// block has used up its piece the whole wave fetches its next one (one 16-byte load per lane), the other blocks carry on from where they were.
#define ZV_G 6u
struct ZvBlock {
    uint32_t ll[512], ml[512], of[256];
    __attribute__((aligned(16))) uint8_t buf[(GC_ZD_CHUNK + 64u) > 2u * (ZD_WIN_S + 16u) ? (GC_ZD_CHUNK + 64u) : 2u * (ZD_WIN_S + 16u)];
    int16_t norm[64]; uint16_t next[64]; uint32_t v[16];                    // v: 0 error, 4 seq stream start, 5..7 table logs
};
// value of `v` in lane k of my group of four lanes
template <int K> __device__ __forceinline__ uint32_t zv_quad(uint32_t v)
{
#ifdef HIPEMU
    return __shfl(v, (int)((__lane_id() & 60u) + (unsigned)K));
#else
    // The move stays an instruction of its own: folded into the subtraction that uses it (v_subrev_u32_dpp ... quad_perm:[1,1,1,1], LLVM's DPP
    // combiner at -O1 and above) the MI355X returned the other lane's value as 0 -- every match length with extra bits came out wrong while the
    // emulator and the same code built with -mllvm -amdgpu-dpp-combine=false were right (tools/gpu_seqv_variants.py).  The empty asm is a use the
    // combiner cannot see through.
    uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55 /* quad_perm:[K,K,K,K] */, 0xF, 0xF, true);
#ifndef ZV_KEEP_DPP_FOLD
    asm volatile("" : "+v"(r));
#endif
    return r;
#endif
}

#ifdef ZV_STRONG_SYNC
#define ZV_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define ZV_SYNC() gc_wave_sync()
#endif
extern "C" __global__ void __launch_bounds__(64)
gc_zstd_dec_seqv_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, const GcZdFrame* __restrict__ frames, GcZdBlock* blocks, GcU4* seqWork,
                        const uint32_t* __restrict__ order, uint32_t nBlocks, uint32_t* ready, unsigned long long* dbg)
{
    __shared__ ZvBlock sB[ZV_G];
    __shared__ ZdConst sK;
#ifdef ZV_ZERO_LDS
    for (uint32_t i = threadIdx.x; i < sizeof(sB) / 4u; i += 64u) ((uint32_t*)sB)[i] = 0;
    gc_wave_sync();
#endif
    const uint32_t lane = threadIdx.x, role = lane & 3u, g = (lane >> 2) < ZV_G ? (lane >> 2) : 0u;
    const uint32_t idx = blockIdx.x * ZV_G + (lane >> 2);
    const bool have = (lane >> 2) < ZV_G && idx < nBlocks;
    const uint32_t b = have ? order[idx] : 0u;
    const uint32_t ty = have ? blocks[b].type : 0u;
    const bool mine = have && (ty & 3u) == 2u && !(ty & GC_ZD_B_BAD);       // my group of four lanes has a compressed block
    ZvBlock& M = sB[g];
    for (uint32_t i = lane; i < 36u; i += 64u) { sK.llBase[i] = kZdLLBase[i]; sK.llBits[i] = kZdLLBits[i]; sK.llNorm[i] = kZdLLNorm[i]; }
    for (uint32_t i = lane; i < 53u; i += 64u) { sK.mlBase[i] = kZdMLBase[i]; sK.mlBits[i] = kZdMLBits[i]; sK.mlNorm[i] = kZdMLNorm[i]; }
    for (uint32_t i = lane; i < 29u; i += 64u) sK.ofNorm[i] = kZdOFNorm[i];
    // block fields (every lane of the group holds them)
    uint64_t eSrcOff = 0, eSeqOff = 0; uint32_t eSize = 0, eSeqPos = 0, eNSeq = 0, eModes = 0, eRegen = 0, eFrame = 0;
    if (mine) { const GcZdBlock* e = blocks + b; eSrcOff = e->srcOff; eSeqOff = e->seqOff; eSize = e->size; eSeqPos = e->seqPos; eNSeq = e->nSeq; eModes = e->modes; eRegen = e->regen; eFrame = e->frame; }
    for (uint32_t gg = 0; gg < ZV_G; gg++) {                               // header windows, the whole wave per block
        if (!gc_readlane(mine ? 1u : 0u, gg * 4u)) continue;
        const uint64_t so = (uint64_t)gc_readlane((uint32_t)eSrcOff, gg * 4u) | ((uint64_t)gc_readlane((uint32_t)(eSrcOff >> 32), gg * 4u) << 32);
        const uint32_t sp0 = gc_readlane(eSeqPos, gg * 4u), bs0 = gc_readlane(eSize, gg * 4u);
        for (uint32_t i = lane; i < ZD_WIN_S + 8u; i += 64u) sB[gg].buf[i] = sp0 + i < bs0 ? src[so + sp0 + i] : (uint8_t)0;
        if (lane < 16u) sB[gg].v[lane] = 0;
    }
    ZV_SYNC();
    const GcZdFrame fr = frames[eFrame];
    const uint8_t* const bsrc = src + eSrcOff;
    if (mine && role == 0u && eNSeq) {                                     // symbol tables: one lane per block (the blocks side by side)
        uint8_t* const sWinS = M.buf; uint8_t* const sWinR = M.buf + ZD_WIN_S + 16u;
        uint32_t err = 0, p = 1;
        const int ord[3] = { ZT_LL, ZT_OF, ZT_ML };
        for (int i = 0; i < 3 && !err; i++) {
            const int which = ord[i];
            uint32_t* const tab = which == ZT_LL ? M.ll : (which == ZT_OF ? M.of : M.ml);
            const uint32_t mode = (eModes >> (which == ZT_LL ? 6u : (which == ZT_OF ? 4u : 2u))) & 3u;
            if (mode != 3u) { p = zd_seq_table(mode, sWinS, ZD_WIN_S, p, tab, &M.v[5 + which], which, &sK, M.norm, M.next); if (!p) err = 1; }
            else {
                const uint32_t k = zd_find_table_source(blocks, fr.blockBase, b, which);
                if (k == 0xFFFFFFFFu) { err = 1; break; }
                const uint32_t m2 = blocks[k].modes;
                const uint32_t mk = (m2 >> (which == ZT_LL ? 6u : (which == ZT_OF ? 4u : 2u))) & 3u;
                uint32_t q = 1;
                if (mk != 0u) {
                    zd_load_window(sWinR, src, srcSize, blocks[k].srcOff + blocks[k].seqPos, ZD_WIN_S);
                    for (int y = 0; y < 3 && ord[y] != which; y++) {
                        const int w2 = ord[y];
                        const int sk = zd_seq_table_skip((m2 >> (w2 == ZT_LL ? 6u : (w2 == ZT_OF ? 4u : 2u))) & 3u, sWinR, ZD_WIN_S, q, w2, M.norm);
                        if (sk < 0) { err = 1; break; }
                        q += (uint32_t)sk;
                    }
                }
                if (!err && !zd_seq_table(mk, sWinR, ZD_WIN_S, q, tab, &M.v[5 + which], which, &sK, M.norm, M.next)) err = 1;
            }
        }
        if (!err && (p > ZD_WIN_S || eSeqPos + p >= eSize)) err = 1;
        M.v[4] = eSeqPos + p; if (err) M.v[0] = 1;
    }
    ZV_SYNC();
    // ---- decode ----
    const uint32_t nSeq = eNSeq;
    uint32_t err = mine && nSeq ? M.v[0] : 0u;
    const uint32_t seqStart = M.v[4];
    const uint64_t spOff = eSrcOff + seqStart;                             // the bitstream of my block, absolute in the compressed stream
    const uint32_t n = (mine && nSeq && !err) ? eSize - seqStart : 1u;
    int32_t off = 0;
    if (mine && nSeq && !err) {
        const uint32_t last = src[spOff + n - 1u];
        if (!last) err = 1; else off = (int32_t)((n - 1u) * 8u + gc_hibit32(last));
    }
    const uint32_t llLog = M.v[5 + ZT_LL], ofLog = M.v[5 + ZT_OF], mlLog = M.v[5 + ZT_ML];
#ifndef ZV_DBG
#define ZV_DBG 0
#endif
    const bool dq = ZV_DBG >= 1 && dbg && blockIdx.x == 0u && lane < 4u;
    const bool dq2 = ZV_DBG >= 2 && dq;
    if (dq2 && role == 0u) { dbg[0] = (unsigned long long)M.v[0] | ((unsigned long long)M.v[4] << 32); dbg[1] = llLog | (ofLog << 8) | (mlLog << 16) | ((unsigned long long)n << 32); dbg[2] = (uint32_t)off | ((unsigned long long)err << 32); dbg[10] = (unsigned long long)mine | ((unsigned long long)nSeq << 32); dbg[11] = eSize | ((unsigned long long)eSeqPos << 32); }
    bool dfirst = true;
    const uint32_t* const myTab = role == 1u ? M.ml : (role == 2u ? M.of : M.ll);
    const uint32_t* const myBase = role == 1u ? sK.mlBase : sK.llBase;
    GcU4* const seq = seqWork + fr.seqBase + eSeqOff;
    uint32_t st = 0, j = 0, dpos = 0, lpos = 0, cLo = 0;
    uint32_t rep0 = GC_ZD_SYM | 0u, rep1 = GC_ZD_SYM | 1u, rep2 = GC_ZD_SYM | 2u;
    bool run = mine && nSeq && !err, staged = false, first = true;
    for (;;) {
        // the blocks whose piece is used up (or who have none yet): the whole wave stages the next piece of each
        const bool want = run && !staged;
        uint64_t m = __ballot(want && role == 0u);
        while (m) {
            const uint32_t L = gc_ctz64(m); m &= m - 1ull;
            const uint32_t gg = L >> 2;
            const int32_t offG = (int32_t)gc_readlane((uint32_t)off, L);
            const uint32_t nG = gc_readlane(n, L);
            const uint64_t spG = (uint64_t)gc_readlane((uint32_t)spOff, L) | ((uint64_t)gc_readlane((uint32_t)(spOff >> 32), L) << 32);
            const uint8_t* const sp = src + spG;
            const uint32_t hi = (uint32_t)(offG + 7) >> 3;
            const uint32_t lo = hi > GC_ZD_CHUNK ? (hi - GC_ZD_CHUNK) & ~7u : 0u;
            uint8_t* const buf = sB[gg].buf;
            if (lane < 2u) { const uint64_t z = 0; __builtin_memcpy(buf + 8u * lane, &z, 8); }
            for (uint32_t i = lane * 8u; i < hi + 8u - lo; i += 512u) {
                uint64_t v = 0;
                if (lo + i + 8u <= nG) v = gc_ld64(sp + lo + i);
                else for (uint32_t k = 0; k < 8u && lo + i + k < nG; k++) v |= (uint64_t)sp[lo + i + k] << (8u * k);
                __builtin_memcpy(buf + 16u + i, &v, 8);
            }
        }
        ZV_SYNC();
        if (want) { const uint32_t hi = (uint32_t)(off + 7) >> 3; cLo = hi > GC_ZD_CHUNK ? (hi - GC_ZD_CHUNK) & ~7u : 0u; staged = true; }
#define ZV_RD64(p, c) (gc_ld64(M.buf + (((uint32_t)((p) + 128) >> 3) - (c))) >> ((uint32_t)(p) & 7u))
#define ZV_RD32(p, c) (gc_ld32(M.buf + (((uint32_t)((p) + 128) >> 3) - (c))) >> ((uint32_t)(p) & 7u))
        if (want && first) {                                               // the three initial states
            const int32_t p1 = off - (int32_t)llLog, p2 = p1 - (int32_t)ofLog, p3 = p2 - (int32_t)mlLog;
            if (p3 < 0) { err = 1; run = false; }
            else {
                const int32_t pm = role == 0u ? p1 : (role == 2u ? p2 : p3);
                const uint32_t lg = role == 0u ? llLog : (role == 2u ? ofLog : mlLog);
                st = role < 3u ? ZV_RD32(pm, cLo) & ((1u << lg) - 1u) : 0u;
                off = p3;
            }
            first = false;
            if (dq2 && role < 3u) dbg[3 + role] = st | ((unsigned long long)(uint32_t)off << 32);
        }
        for (;;) {
            const bool can = run && (cLo == 0u || ((uint32_t)off >> 3) >= cLo + 16u);
            if (__ballot(run && !can)) break;                              // somebody needs the next piece of its stream
            if (!__ballot(can)) break;
            const uint32_t en = myTab[can ? st : 0u];
            const uint32_t more = j + 1u < nSeq ? 15u : 0u;                 // the last sequence does not move the states on
            const uint32_t eb = (en >> 14) & 31u, nb = (en >> 10) & more;
            const uint32_t llb = zv_quad<0>(eb), mlb = zv_quad<1>(eb), ofb = zv_quad<2>(eb);
            const uint32_t nl = zv_quad<0>(nb), nm = zv_quad<1>(nb), no = zv_quad<2>(nb);
            const int32_t p1 = off - (int32_t)ofb, p2 = p1 - (int32_t)mlb, p3 = p2 - (int32_t)llb;      // extra bits: offset, match length, literal length
            const int32_t p4 = p3 - (int32_t)nl, p5 = p4 - (int32_t)nm, p6 = p5 - (int32_t)no;          // state bits: literal length, match length, offset
            const int32_t pe = !can ? 0 : (role == 0u ? p3 : (role == 1u ? p2 : p1)), ps = !can ? 0 : (role == 0u ? p4 : (role == 1u ? p5 : p6));
            const uint32_t cc = can ? cLo : 0u;
            const uint64_t we = ZV_RD64(pe, cc);
            const uint32_t ws = ZV_RD32(ps, cc);
            const uint32_t base = role == 2u ? 1u << eb : myBase[role < 2u ? en >> 19 : 0u];
            const uint32_t val = base + ((uint32_t)we & ((1u << eb) - 1u));
            const uint32_t stNew = role < 3u ? (en & 0x3FFu) + (ws & ((1u << nb) - 1u)) : 0u;
            const uint32_t ll = zv_quad<0>(val), ml = zv_quad<1>(val), ofv = zv_quad<2>(val);
            if (dq2 && dfirst && can) { if (role < 3u) dbg[6 + role] = en | ((unsigned long long)val << 32); if (role == 0u) { dbg[9] = ll | ((unsigned long long)ml << 20) | ((unsigned long long)ofv << 40); dbg[12] = llb | (mlb << 8) | (ofb << 16) | ((unsigned long long)(nl | (nm << 8) | (no << 16)) << 32); dbg[13] = (uint32_t)p6 | ((unsigned long long)cLo << 32); } dfirst = false; }
            if (can) {
                st = stNew; off = p6;
                // offset value 1..3 = one of the three last offsets (shifted by one when the sequence has no literals; "4" = the first minus 1)
                const bool isRep = ofv <= 3u;
                const uint32_t ix = ofv - 1u + (ll == 0u ? 1u : 0u);
                const uint32_t r0m1 = (rep0 & GC_ZD_SYM) ? rep0 + 4u : rep0 - 1u;
                const uint32_t o = !isRep ? ofv - 3u : (ix == 0u ? rep0 : (ix == 1u ? rep1 : (ix == 2u ? rep2 : r0m1)));
                const bool shift3 = !isRep || ix >= 2u, shift2 = isRep && ix == 1u;
                rep2 = shift3 ? rep1 : rep2;
                rep1 = (shift3 || shift2) ? rep0 : rep1;
                rep0 = o;
                const bool bad = off < 0 || o == 0u || lpos + ll > eRegen || dpos + ll + ml > GC_ZSTD_BLOCK_MAX;
                if (ofb > 30u) { err = GC_ZD_UNSUPPORTED; run = false; }
                else if (bad) { err = 1; run = false; }
                else {
                    if (role == 0u) { GcU4 rec; rec.x = ll | (ml << 18); rec.y = (ml >> 14) | (lpos << 4); rec.z = o; rec.w = dpos; seq[j] = rec; }
                    dpos += ll + ml; lpos += ll; j++;
                    if (j >= nSeq) { run = false; if (off != 0) err = 1; }
                }
            }
        }
#undef ZV_RD64
#undef ZV_RD32
        if (run) { const bool can = cLo == 0u || ((uint32_t)off >> 3) >= cLo + 16u; if (!can) staged = false; }
        if (!__ballot(run)) break;
    }
    if (dq && role == 0u) { dbg[14] = err | ((unsigned long long)j << 32); dbg[15] = (uint32_t)off | ((unsigned long long)dpos << 32); }
    if (mine && role == 0u) {
        uint32_t status = err ? (err == GC_ZD_UNSUPPORTED ? GC_ZD_UNSUPPORTED : GC_ZD_CORRUPT) : GC_ZD_OK;
        const uint32_t outSize = dpos + (eRegen - lpos);
        if (!status && outSize > GC_ZSTD_BLOCK_MAX) status = GC_ZD_CORRUPT;
        blocks[b].status = status; blocks[b].outSize = outSize; blocks[b].lposEnd = lpos; blocks[b].dposEnd = dpos;
        blocks[b].rep[0] = rep0; blocks[b].rep[1] = rep1; blocks[b].rep[2] = rep2;
    }
#ifdef HIPEMU
    hipemu::wave_barrier();
    if (mine && role == 0u) __atomic_fetch_add(&ready[b], 1u, __ATOMIC_RELEASE);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (mine && role == 0u) __hip_atomic_fetch_add(&ready[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// =============================================================== execution kernel ===============================================================
// byte `si` of the block image, or of the frame's output in front of the block when si < 0.  (The LDS read is unconditional and the HBM read sits in
// its own branch: a select between an LDS and a global pointer makes this compiler emit an illegal compare against src_shared_base.)
__device__ __forceinline__ uint8_t zd_image_byte(const uint8_t* sOut, const uint8_t* bdst, int32_t si)
{
    uint8_t b = sOut[si < 0 ? 0 : si];
    if (si < 0) b = bdst[si];
    return b;
}

__device__ __forceinline__ uint64_t zd_rotl(uint64_t v, uint32_t r) { return (v << r) | (v >> (64u - r)); }
#define XP1 0x9E3779B185EBCA87ull
#define XP2 0xC2B2AE3D27D4EB4Full
#define XP3 0x165667B19E3779F9ull
#define XP4 0x85EBCA77C2B2AE63ull
#define XP5 0x27D4EB2F165667C5ull
__device__ __forceinline__ uint64_t zd_xround(uint64_t acc, uint64_t in) { return zd_rotl(acc + in * XP2, 31) * XP1; }

// n bytes from HBM to LDS; src may be read up to 7 bytes past src + n when that stays below srcLimit
__device__ __forceinline__ void zd_copy_in(uint8_t* d, const uint8_t* s, uint32_t n, const uint8_t* srcLimit)
{
    uint32_t k = 0;
    for (; k + 8u <= n; k += 8u) { const uint64_t v = gc_ld64(s + k); __builtin_memcpy(d + k, &v, 8); }
    if (k < n) {
        if (s + k + 8u <= srcLimit) { uint64_t v = gc_ld64(s + k); for (; k < n; k++) { d[k] = (uint8_t)v; v >>= 8; } }
        else for (; k < n; k++) d[k] = s[k];
    }
}

enum { XV_ERR = 0, XV_FRAME, XV_REP0, XV_REP1, XV_REP2, XV_COUNT };

extern "C" __global__ void __launch_bounds__(GC_ZD_T)
gc_zstd_dec_exec_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint8_t* dst, uint64_t dstCap, const GcZdFrame* __restrict__ frames, uint32_t nFrames,
                        const GcZdBlock* blocks, uint32_t* ticket, const uint8_t* litWork, uint64_t litWorkSize, GcU4* seqWork, uint64_t* result,
                        unsigned long long* prof, const uint32_t* ready)
{
    __shared__ __attribute__((aligned(16))) uint8_t sOut[GC_ZSTD_BLOCK_MAX + 32u];
    __shared__ uint32_t sV[XV_COUNT];
    __shared__ uint64_t sAcc[4];
    const uint32_t t = threadIdx.x, lane = t & 63u;

    for (;;) {
        if (t == 0) sV[XV_FRAME] = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t f = sV[XV_FRAME];
        __syncthreads();
        if (f >= nFrames) return;
        const GcZdFrame fr = frames[f];
        const uint8_t* const fsrc = src + fr.srcOff;
        uint8_t* const fdst = dst + fr.dstOff;
        const uint64_t cap = (fr.flags & GC_ZD_F_SIZE_KNOWN) ? fr.contentSize : (dstCap - fr.dstOff);
        const uint64_t srcEnd = fr.srcSize - ((fr.flags & GC_ZD_F_CHECKSUM) ? 4u : 0u);
        uint64_t produced = 0;
        if (t == 0) { sV[XV_ERR] = GC_ZD_OK; sV[XV_REP0] = 1; sV[XV_REP1] = 4; sV[XV_REP2] = 8; }
        __syncthreads();

        for (uint32_t bi = 0; bi < fr.nBlocks; bi++) {
            {   // the entropy kernels may still be at work on this block (they run beside this kernel): wait for its one or two workgroups
                const uint32_t ty = blocks[fr.blockBase + bi].type, li = blocks[fr.blockBase + bi].litInfo;       // (written by the index kernel: final)
                if ((ty & 3u) == 2u && !(ty & GC_ZD_B_BAD)) {
                    const uint32_t need = 1u + ((li & 3u) >= 2u ? 1u : 0u);
                    if (t == 0) {
                        uint32_t spins = 0;
                        while (gc_poll_device(&ready[fr.blockBase + bi]) < need) { gc_nap(); if (++spins > (1u << 24)) { sV[XV_ERR] = GC_ZD_CORRUPT; break; } }
                    }
                    __syncthreads();
                    gc_acquire_device();
                    if (sV[XV_ERR]) break;
                }
            }
            const GcZdBlock e = blocks[fr.blockBase + bi];
            const uint32_t bt = e.type & 3u;
            uint32_t fail = 0;
            if (e.type & GC_ZD_B_BAD) fail = GC_ZD_CORRUPT;
            else if (bt == 2u && (e.status || e.litStatus)) fail = e.status ? e.status : e.litStatus;
            else if (produced + (bt == 2u ? e.outSize : e.regen) > cap) fail = GC_ZD_DST_SMALL;
            if (fail) { if (t == 0) sV[XV_ERR] = fail; break; }
            const uint8_t* const bsrc = src + e.srcOff;
            uint8_t* const bdst = fdst + produced;
            if (bt == 0u) {
                for (uint32_t i = t; i < e.regen; i += GC_ZD_T) bdst[i] = bsrc[i];
                produced += e.regen;
            } else if (bt == 1u) {
                const uint8_t v = bsrc[0];
                for (uint32_t i = t; i < e.regen; i += GC_ZD_T) bdst[i] = v;
                produced += e.regen;
            } else {
                const uint32_t nSeq = e.nSeq, lt = e.litInfo & 3u, outSize = e.outSize;
                const uint8_t* const litSrc = lt >= 2u ? litWork + fr.litBase + e.litOff : bsrc + (e.litInfo >> 8);
                const uint8_t* const litLimit = lt >= 2u ? litWork + litWorkSize : src + srcSize;
                const uint8_t rleByte = lt == 1u ? bsrc[e.litInfo >> 8] : (uint8_t)0;
                GcU4* const seq = seqWork + fr.seqBase + e.seqOff;
                const uint32_t in0 = sV[XV_REP0], in1 = sV[XV_REP1], in2 = sV[XV_REP2];
                // ---- pass 1: offsets get their values; literals; matches that lie in front of the block ----
                const unsigned long long c0 = prof ? gc_clock() : 0ull;
                uint32_t bad = 0;
                GcU4 nextRec; nextRec.x = nextRec.y = nextRec.z = nextRec.w = 0;
                if (t < nSeq) nextRec = seq[t];
                for (uint32_t j = t; j < nSeq; j += GC_ZD_T) {
                    GcU4 rec = nextRec;
                    if (j + GC_ZD_T < nSeq) nextRec = seq[j + GC_ZD_T];
                    const uint32_t ll = rec.x & 0x3FFFFu, ml = (rec.x >> 18) | ((rec.y & 15u) << 14), lp = rec.y >> 4, dp = rec.w;
                    uint32_t off = rec.z;
                    if (off & GC_ZD_SYM) {
                        const uint32_t k = off & 3u, delta = (off & 0x7FFFFFFFu) >> 2, in = k == 0u ? in0 : (k == 1u ? in1 : in2);
                        if (in <= delta) { bad = 1; continue; }
                        off = in - delta; rec.z = off; seq[j] = rec;
                    }
                    const uint32_t d = dp + ll;
                    if ((uint64_t)off > produced + d) { bad = 1; continue; }
                    if (lt == 1u) for (uint32_t k = 0; k < ll; k++) sOut[dp + k] = rleByte;
                    else zd_copy_in(sOut + dp, litSrc + lp, ll, litLimit);
                    if (off >= d + ml) zd_copy_in(sOut + d, bdst + d - off, ml, bdst);      // the whole source lies in front of the block
                }
                {
                    const uint32_t lp = e.lposEnd, dp = e.dposEnd, rest = e.regen - lp;
                    if (lt == 1u) for (uint32_t k = t; k < rest; k += GC_ZD_T) sOut[dp + k] = rleByte;
                    else for (uint32_t k = t; k < rest; k += GC_ZD_T) sOut[dp + k] = litSrc[lp + k];
                }
                if (bad) sV[XV_ERR] = GC_ZD_CORRUPT;
                __threadfence();
                __syncthreads();
                if (sV[XV_ERR]) break;
                const unsigned long long c1 = prof ? gc_clock() : 0ull;
                // ---- pass 2: matches that read the block itself.  One lane per sequence, 64 sequences at a time; everything below the output
                //      position W of the first unfinished match of the group is final, so every match whose source ends at or below W can be
                //      copied now, all of them at once (their outputs are disjoint); the first one itself always can.  A few rounds per group
                //      instead of 64 dependent steps.  Long matches are copied by the whole wave, 64 bytes per step. ----
                if (t < 64u) {
                    uint32_t nGroups = 0, nRounds = 0, nSpecial = 0;
                    GcU4 ahead; ahead.x = ahead.y = ahead.z = ahead.w = 0;             // the records of the next group are fetched while this one is worked on
                    if (lane < nSeq) ahead = seq[lane];
                    for (uint32_t j0 = 0; j0 < nSeq; j0 += 64u) {
                        const GcU4 mine = ahead;
                        const bool have = j0 + lane < nSeq;
                        if (j0 + 64u + lane < nSeq) ahead = seq[j0 + 64u + lane];
                        const uint32_t ll = mine.x & 0x3FFFFu, ml0 = (mine.x >> 18) | ((mine.y & 15u) << 14), off0 = mine.z;
                        // what is left of my match: output position, length, distance.  A lane copies at most 16 bytes per round (so that one
                        // long match does not hold up the round); what remains is the same kind of match further on.
                        uint32_t d = mine.w + ll, ml = ml0, off = off0;
                        bool pending = have && off < d + ml;
                        // matches the lane-parallel path does not take: long ones, periods below 8 bytes, sources that start in front of the block.
                        // They wait until they are first, then the whole wave copies them (64 bytes per step).
                        const bool special = ml > 64u || off < 8u || off > d;
                        nGroups++;
                        for (;;) {
                            const uint64_t pm = __ballot(pending);
                            if (!pm) break;
                            nRounds++;
                            const uint32_t fl = gc_ctz64(pm);
                            const uint32_t W = gc_readlane(d, fl);
                            if (gc_readlane(special ? 1u : 0u, fl)) {
                                const uint32_t fml = gc_readlane(ml, fl), foff = gc_readlane(off, fl);
                                const int32_t fs0 = (int32_t)W - (int32_t)foff;
                                nSpecial++;
                                if (fs0 >= 0 && foff < 64u) {          // (the common cases stay inside LDS)
                                    for (uint32_t k = lane; k < fml; k += 64u) sOut[W + k] = sOut[(uint32_t)fs0 + k % foff];
                                    gc_wave_step();
                                } else if (fs0 >= 0) {
                                    for (uint32_t c = 0; c < fml; c += 64u) {
                                        const uint32_t k = c + lane;
                                        if (k < fml) sOut[W + k] = sOut[(uint32_t)fs0 + k];
                                        gc_wave_step();
                                    }
                                } else if (foff < 64u) {
                                    for (uint32_t k = lane; k < fml; k += 64u) {
                                        const int32_t si = fs0 + (int32_t)(k % foff);
                                        sOut[W + k] = zd_image_byte(sOut, bdst, si);
                                    }
                                    gc_wave_step();
                                } else {
                                    for (uint32_t c = 0; c < fml; c += 64u) {
                                        const uint32_t k = c + lane;
                                        if (k < fml) { const int32_t si = fs0 + (int32_t)k; sOut[W + k] = zd_image_byte(sOut, bdst, si); }
                                        gc_wave_step();
                                    }
                                }
                                if (lane == fl) pending = false;
                                continue;
                            }
                            const uint32_t s0 = d - off;
                            const uint32_t srcEnd = off >= ml ? s0 + ml : d;      // end of the part of the source that is not my own output
                            if (pending && !special && (lane == fl || srcEnd <= W)) {
#pragma unroll
                                for (int step = 0; step < 2; step++) {
                                    if (ml) {
                                        const uint64_t v = gc_ld64(sOut + (d - off));
                                        if (ml >= 8u) { __builtin_memcpy(sOut + d, &v, 8); d += 8u; ml -= 8u; }
                                        else {
                                            uint32_t lo = (uint32_t)v, k = d;
                                            if (ml & 4u) { __builtin_memcpy(sOut + k, &lo, 4); k += 4u; lo = (uint32_t)(v >> 32); }
                                            if (ml & 2u) { const uint16_t h = (uint16_t)lo; __builtin_memcpy(sOut + k, &h, 2); k += 2u; lo >>= 16; }
                                            if (ml & 1u) sOut[k] = (uint8_t)lo;
                                            d += ml; ml = 0;
                                        }
                                    }
                                }
                                pending = ml != 0u;
                            }
                            gc_wave_step();
                        }
                    }
                    if (prof && lane == 0u) { atomicAdd(&prof[4], (unsigned long long)nRounds); atomicAdd(&prof[5], (unsigned long long)nGroups); atomicAdd(&prof[6], (unsigned long long)nSpecial); }
                }
                if (t == 64u) {                                    // repeat offsets behind the block
                    uint32_t out[3];
                    for (int i = 0; i < 3; i++) {
                        uint32_t v = e.rep[i];
                        if (v & GC_ZD_SYM) { const uint32_t k = v & 3u, delta = (v & 0x7FFFFFFFu) >> 2, in = k == 0u ? in0 : (k == 1u ? in1 : in2); v = in > delta ? in - delta : 1u; }
                        out[i] = v;
                    }
                    sV[XV_REP0] = out[0]; sV[XV_REP1] = out[1]; sV[XV_REP2] = out[2];
                }
                __syncthreads();
                const unsigned long long c2 = prof ? gc_clock() : 0ull;
                // ---- flush ----
                for (uint32_t i = t * 16u; i < outSize; i += GC_ZD_T * 16u) {
                    if (i + 16u <= outSize) { GcU4 v; __builtin_memcpy(&v, sOut + i, 16); __builtin_memcpy(bdst + i, &v, 16); }
                    else for (uint32_t k = i; k < outSize; k++) bdst[k] = sOut[k];
                }
                produced += outSize;
                if (prof && t == 0) {
                    const unsigned long long c3 = gc_clock();
                    atomicAdd(&prof[0], c1 - c0); atomicAdd(&prof[1], c2 - c1); atomicAdd(&prof[2], c3 - c2); atomicAdd(&prof[3], 1ull);
                }
            }
            __threadfence();
            __syncthreads();
        }
        __syncthreads();
        uint32_t status = sV[XV_ERR];
        if (!status && fr.nBlocks == 0u) status = GC_ZD_CORRUPT;
        if (!status && (fr.flags & GC_ZD_F_SIZE_KNOWN) && produced != fr.contentSize) status = GC_ZD_SIZE;
        if (!status && (fr.flags & GC_ZD_F_CHECKSUM)) {
            // XXH64, seed 0: lanes 0..3 of wave 0 are the four accumulators
            const uint64_t nStripes = produced >> 5;
            if (t < 4u) {
                uint64_t acc = t == 0u ? XP1 + XP2 : (t == 1u ? XP2 : (t == 2u ? 0ull : 0ull - XP1));
                const uint8_t* q = fdst + 8u * t;
                for (uint64_t s = 0; s < nStripes; s++) acc = zd_xround(acc, gc_ld64(q + (s << 5)));
                sAcc[t] = acc;
            }
            __syncthreads();
            if (t == 0) {
                uint64_t h;
                if (produced >= 32u) {
                    h = zd_rotl(sAcc[0], 1) + zd_rotl(sAcc[1], 7) + zd_rotl(sAcc[2], 12) + zd_rotl(sAcc[3], 18);
                    for (int k = 0; k < 4; k++) h = (h ^ zd_xround(0, sAcc[k])) * XP1 + XP4;
                } else h = XP5;
                h += produced;
                uint64_t pos = nStripes << 5;
                while (pos + 8u <= produced) { h ^= zd_xround(0, gc_ld64(fdst + pos)); h = zd_rotl(h, 27) * XP1 + XP4; pos += 8u; }
                if (pos + 4u <= produced) { h ^= (uint64_t)gc_ld32(fdst + pos) * XP1; h = zd_rotl(h, 23) * XP2 + XP3; pos += 4u; }
                while (pos < produced) { h ^= (uint64_t)fdst[pos] * XP5; h = zd_rotl(h, 11) * XP1; pos++; }
                h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
                const uint32_t want = (uint32_t)fsrc[srcEnd] | ((uint32_t)fsrc[srcEnd + 1] << 8) | ((uint32_t)fsrc[srcEnd + 2] << 16) | ((uint32_t)fsrc[srcEnd + 3] << 24);
                sV[XV_ERR] = (uint32_t)h == want ? GC_ZD_OK : GC_ZD_CHECKSUM;
            }
            __syncthreads();
            status = sV[XV_ERR];
        }
        if (t == 0) result[f] = produced | ((uint64_t)status << 56);
        __syncthreads();
    }
}

// =============================================================== wide execution ===============================================================
// The execution kernel above copies one frame per workgroup, its blocks in order: the unit of parallelism is the frame, and a stream of few large
// frames (what the reference's single-threaded encoder writes) leaves the machine empty.  The wide path executes ALL blocks of ALL frames at once.
// A match byte cannot be copied before its source exists, so every content byte first gets a POINTER instead of a value:
//   place   one wave per frame walks its blocks (sizes and repeat offsets behind every block are known from the entropy stage): where each block's
//           content starts, which repeat offsets it starts with, whether the frame is still sound there;
//   spread  one workgroup per block: literal bytes go to their place (they are final), every match byte p gets ptr[p] = p - offset; a final
//           byte's entry is a MARK that carries the byte itself (0xFFFFFF00 | byte);
//   chase   rounds over all bytes that are not final: w = ptr[ptr[p]]; if w is a mark, the byte is known: store it and take the mark over;
//           else ptr[p] = w (pointer jumping: the distance to the literal at the end of the chain at least halves per round, so a chain of
//           length L takes at most log2 L rounds; a byte follows up to GC_ZW_HOPS links per round, which saves passes over the array).
//           Because a mark holds the value, whatever a lane reads is usable at once, also what another lane wrote a moment ago -- no ordering
//           inside a launch is needed, stale reads only cost a round.  Each XCD sweeps its own contiguous eighth of the content in address
//           order, so that a byte's source mostly lies where the same L2 has already seen it settle.  Pieces of 1 KiB that are final
//           everywhere are skipped from then on;
//   finish  per frame: size checks and the content checksum.
// ptr entries are positions relative to the first content byte of the batch.
#define GC_ZW_FINAL      0xFFFFFF00u              // ptr values from here on are marks: 0xFFFFFF00 | the byte
#define GC_ZW_PIECE      1024u                    // bytes per "all final" flag
#define GC_ZW_WG_BYTES   4096u                    // bytes per workgroup of the chase kernel
#define GC_ZW_HOPS       12                       // links a byte follows per round (1 GB of text: 1 link 7 rounds 51.9 ms, 4 links 3 rounds 18.5 ms, 12 links 2 rounds 13.7 ms)
#define GC_ZW_LANE_MAX   48u                      // literal runs / matches up to this length are written by the sequence's own lane

extern "C" __global__ void __launch_bounds__(64)
gc_zstd_dec_place_kernel(const GcZdFrame* __restrict__ frames, uint32_t nFrames, const GcZdBlock* __restrict__ blocks, uint64_t dstCap, GcZdPlace* __restrict__ place,
                         uint64_t* __restrict__ result, uint32_t* __restrict__ ferr)
{
    __shared__ uint32_t sIn[64][8];               // type, status, litStatus, size, rep[3]
    __shared__ GcZdPlace sOut[64];
    const uint32_t f = blockIdx.x, lane = threadIdx.x;
    if (f >= nFrames) return;
    const GcZdFrame fr = frames[f];
    const uint64_t cap = (fr.flags & GC_ZD_F_SIZE_KNOWN) ? fr.contentSize : (dstCap - fr.dstOff);
    uint64_t produced = 0;
    uint32_t r0 = 1, r1 = 4, r2 = 8, status = GC_ZD_OK;
    for (uint32_t b0 = 0; b0 < fr.nBlocks; b0 += 64u) {
        const uint32_t cnt = fr.nBlocks - b0 < 64u ? fr.nBlocks - b0 : 64u;
        if (lane < cnt) {
            const GcZdBlock* e = blocks + fr.blockBase + b0 + lane;
            const uint32_t ty = e->type;
            sIn[lane][0] = ty; sIn[lane][1] = e->status; sIn[lane][2] = e->litStatus; sIn[lane][3] = (ty & 3u) == 2u ? e->outSize : e->regen;
            sIn[lane][4] = e->rep[0]; sIn[lane][5] = e->rep[1]; sIn[lane][6] = e->rep[2];
        }
        __syncthreads();
        if (lane == 0) for (uint32_t k = 0; k < cnt; k++) {
            const uint32_t ty = sIn[k][0], bt = ty & 3u, size = sIn[k][3];
            if (!status) {
                if (ty & GC_ZD_B_BAD) status = GC_ZD_CORRUPT;
                else if (bt == 2u && (sIn[k][1] || sIn[k][2])) status = sIn[k][1] ? sIn[k][1] : sIn[k][2];
                else if (produced + size > cap) status = GC_ZD_DST_SMALL;
            }
            GcZdPlace pl; pl.dst = produced; pl.rep[0] = r0; pl.rep[1] = r1; pl.rep[2] = r2; pl.skip = status ? 1u : 0u;
            sOut[k] = pl;
            if (!status) {
                if (bt == 2u) {
                    uint32_t out[3];
                    for (int i = 0; i < 3; i++) {
                        uint32_t v = sIn[k][4 + i];
                        if (v & GC_ZD_SYM) { const uint32_t kk = v & 3u, delta = (v & 0x7FFFFFFFu) >> 2, in = kk == 0u ? r0 : (kk == 1u ? r1 : r2); v = in > delta ? in - delta : 1u; }
                        out[i] = v;
                    }
                    r0 = out[0]; r1 = out[1]; r2 = out[2];
                }
                produced += size;
            }
        }
        __syncthreads();
        if (lane < cnt) place[fr.blockBase + b0 + lane] = sOut[lane];
        __syncthreads();
        status = gc_readlane(status, 0); r0 = gc_readlane(r0, 0); r1 = gc_readlane(r1, 0); r2 = gc_readlane(r2, 0);
        produced = (uint64_t)gc_readlane((uint32_t)produced, 0) | ((uint64_t)gc_readlane((uint32_t)(produced >> 32), 0) << 32);
    }
    if (lane == 0) { result[f] = produced | ((uint64_t)status << 56); ferr[f] = 0; }
}

extern "C" __global__ void __launch_bounds__(256)
gc_zstd_dec_spread_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, uint8_t* __restrict__ dst, const GcZdFrame* __restrict__ frames, const GcZdBlock* __restrict__ blocks,
                          const GcZdPlace* __restrict__ place, const uint8_t* __restrict__ litWork, uint64_t litWorkSize, const GcU4* __restrict__ seqWork,
                          uint32_t* __restrict__ ptr, uint64_t batchBase, uint32_t* __restrict__ ferr)
{
    __shared__ uint32_t sList[256];
    __shared__ uint32_t sCount;
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    const GcZdPlace pl = place[b];
    if (pl.skip) return;
    const GcZdBlock e = blocks[b];
    const GcZdFrame fr = frames[e.frame];
    const uint32_t bt = e.type & 3u;
    const uint8_t* const bsrc = src + e.srcOff;
    uint8_t* const bdst = dst + fr.dstOff + pl.dst;
    const uint32_t P = (uint32_t)(fr.dstOff - batchBase + pl.dst);                 // the block's first byte as a ptr index
    uint32_t* const bptr = ptr + P;
    if (bt == 0u) { for (uint32_t i = t; i < e.regen; i += 256u) { const uint8_t v = bsrc[i]; bdst[i] = v; bptr[i] = GC_ZW_FINAL | v; } return; }
    if (bt == 1u) { const uint8_t v = bsrc[0]; for (uint32_t i = t; i < e.regen; i += 256u) { bdst[i] = v; bptr[i] = GC_ZW_FINAL | v; } return; }
    const uint32_t nSeq = e.nSeq, lt = e.litInfo & 3u;
    const uint8_t* const litSrc = lt >= 2u ? litWork + fr.litBase + e.litOff : bsrc + (e.litInfo >> 8);
    const uint8_t rleByte = lt == 1u ? bsrc[e.litInfo >> 8] : (uint8_t)0;
    const GcU4* const seq = seqWork + fr.seqBase + e.seqOff;
    uint32_t bad = 0;
    for (uint32_t j0 = 0; j0 < nSeq; j0 += 256u) {
        if (t == 0) sCount = 0;
        __syncthreads();
        const uint32_t j = j0 + t;
        if (j < nSeq) {
            const GcU4 rec = seq[j];
            const uint32_t ll = rec.x & 0x3FFFFu, ml = (rec.x >> 18) | ((rec.y & 15u) << 14), lp = rec.y >> 4, dp = rec.w;
            uint32_t off = rec.z;
            bool ok = true;
            if (off & GC_ZD_SYM) {
                const uint32_t k = off & 3u, delta = (off & 0x7FFFFFFFu) >> 2, in = pl.rep[k == 0u ? 0 : (k == 1u ? 1 : 2)];
                if (in <= delta) ok = false; else off = in - delta;
            }
            const uint32_t d = dp + ll;
            if (ok && (uint64_t)off > pl.dst + d) ok = false;
            if (!ok) bad = 1;
            else if (ll > GC_ZW_LANE_MAX || ml > GC_ZW_LANE_MAX) sList[atomicAdd(&sCount, 1u)] = j;
            else {
                for (uint32_t k = 0; k < ll; k++) { const uint8_t v = lt == 1u ? rleByte : litSrc[lp + k]; bdst[dp + k] = v; bptr[dp + k] = GC_ZW_FINAL | v; }
                const uint32_t q0 = P + d - off;
                for (uint32_t k = 0; k < ml; k++) bptr[d + k] = q0 + k;
            }
        }
        __syncthreads();
        const uint32_t nLong = sCount;
        for (uint32_t i = 0; i < nLong; i++) {                                     // long runs: the whole workgroup
            const GcU4 rec = seq[sList[i]];
            const uint32_t ll = rec.x & 0x3FFFFu, ml = (rec.x >> 18) | ((rec.y & 15u) << 14), lp = rec.y >> 4, dp = rec.w;
            uint32_t off = rec.z;
            if (off & GC_ZD_SYM) { const uint32_t k = off & 3u, delta = (off & 0x7FFFFFFFu) >> 2; off = pl.rep[k == 0u ? 0 : (k == 1u ? 1 : 2)] - delta; }
            const uint32_t d = dp + ll, q0 = P + d - off;
            for (uint32_t k = t; k < ll; k += 256u) { const uint8_t v = lt == 1u ? rleByte : litSrc[lp + k]; bdst[dp + k] = v; bptr[dp + k] = GC_ZW_FINAL | v; }
            for (uint32_t k = t; k < ml; k += 256u) bptr[d + k] = q0 + k;
        }
        __syncthreads();
    }
    {
        const uint32_t lp = e.lposEnd, dp = e.dposEnd, rest = e.regen - lp;
        for (uint32_t k = t; k < rest; k += 256u) { const uint8_t v = lt == 1u ? rleByte : litSrc[lp + k]; bdst[dp + k] = v; bptr[dp + k] = GC_ZW_FINAL | v; }
    }
    if (bad) atomicMax(&ferr[e.frame], (uint32_t)GC_ZD_CORRUPT);
}

extern "C" __global__ void __launch_bounds__(256)
gc_zstd_dec_chase_kernel(uint8_t* dst /* first content byte of the batch */, uint32_t* ptr, uint32_t n, uint32_t groupsPerXcd, uint32_t hops, uint8_t* pieceDone, uint32_t* counter)
{
    __shared__ uint32_t sLeft;
    const uint32_t t = threadIdx.x;
    // workgroups go round the XCDs (blockIdx % 8): XCD x takes the groups [x * groupsPerXcd, (x + 1) * groupsPerXcd) in order
    const uint32_t group = (blockIdx.x & 7u) * groupsPerXcd + (blockIdx.x >> 3);
    for (uint32_t it = 0; it < GC_ZW_WG_BYTES / GC_ZW_PIECE; it++) {
        const uint32_t piece = group * (GC_ZW_WG_BYTES / GC_ZW_PIECE) + it;
        const uint32_t p = piece * GC_ZW_PIECE + t * 4u;
        if ((uint64_t)piece * GC_ZW_PIECE >= n) break;
        if (pieceDone[piece]) continue;
        if (t == 0) sLeft = 0;
        __syncthreads();
        GcU4 v = *(const GcU4*)(ptr + p);                                          // (the array is padded with marks up to a whole workgroup)
        uint32_t x[4] = { v.x, v.y, v.z, v.w };
        uint32_t left = 0; bool changed = false;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t w = x[i];
            if (w >= GC_ZW_FINAL) continue;
            for (uint32_t hop = 0; hop < hops && w < GC_ZW_FINAL; hop++) w = ptr[w];      // a few links per round: fewer passes over the array
            x[i] = w; changed = true;
            if (w >= GC_ZW_FINAL) dst[p + (uint32_t)i] = (uint8_t)w; else left++;
        }
        if (changed) { v.x = x[0]; v.y = x[1]; v.z = x[2]; v.w = x[3]; *(GcU4*)(ptr + p) = v; }
        if (left) atomicAdd(&sLeft, left);
        __syncthreads();
        if (t == 0) { if (sLeft) atomicAdd(counter, sLeft); else pieceDone[piece] = 1; }
        __syncthreads();
    }
}

extern "C" __global__ void __launch_bounds__(64)
gc_zstd_dec_finish_kernel(const uint8_t* __restrict__ src, const uint8_t* __restrict__ dst, const GcZdFrame* __restrict__ frames, uint32_t nFrames, uint64_t* __restrict__ result,
                          const uint32_t* __restrict__ ferr)
{
    __shared__ uint64_t sAcc[4];
    const uint32_t f = blockIdx.x, t = threadIdx.x;
    if (f >= nFrames) return;
    const GcZdFrame fr = frames[f];
    const uint64_t r = result[f], produced = r & 0x00FFFFFFFFFFFFFFull;
    uint32_t status = (uint32_t)(r >> 56);
    if (!status && ferr[f]) status = ferr[f];
    if (!status && fr.nBlocks == 0u) status = GC_ZD_CORRUPT;
    if (!status && (fr.flags & GC_ZD_F_SIZE_KNOWN) && produced != fr.contentSize) status = GC_ZD_SIZE;
    if (!status && (fr.flags & GC_ZD_F_CHECKSUM)) {
        const uint8_t* const fdst = dst + fr.dstOff;
        const uint8_t* const fsrc = src + fr.srcOff;
        const uint64_t srcEnd = fr.srcSize - 4u;
        const uint64_t nStripes = produced >> 5;
        if (t < 4u) {                                                              // XXH64, seed 0: lanes 0..3 are the four accumulators
            uint64_t acc = t == 0u ? XP1 + XP2 : (t == 1u ? XP2 : (t == 2u ? 0ull : 0ull - XP1));
            const uint8_t* q = fdst + 8u * t;
            for (uint64_t s = 0; s < nStripes; s++) acc = zd_xround(acc, gc_ld64(q + (s << 5)));
            sAcc[t] = acc;
        }
        __syncthreads();
        if (t == 0) {
            uint64_t h;
            if (produced >= 32u) {
                h = zd_rotl(sAcc[0], 1) + zd_rotl(sAcc[1], 7) + zd_rotl(sAcc[2], 12) + zd_rotl(sAcc[3], 18);
                for (int k = 0; k < 4; k++) h = (h ^ zd_xround(0, sAcc[k])) * XP1 + XP4;
            } else h = XP5;
            h += produced;
            uint64_t pos = nStripes << 5;
            while (pos + 8u <= produced) { h ^= zd_xround(0, gc_ld64(fdst + pos)); h = zd_rotl(h, 27) * XP1 + XP4; pos += 8u; }
            if (pos + 4u <= produced) { h ^= (uint64_t)gc_ld32(fdst + pos) * XP1; h = zd_rotl(h, 23) * XP2 + XP3; pos += 4u; }
            while (pos < produced) { h ^= (uint64_t)fdst[pos] * XP5; h = zd_rotl(h, 11) * XP1; pos++; }
            h ^= h >> 33; h *= XP2; h ^= h >> 29; h *= XP3; h ^= h >> 32;
            const uint32_t want = (uint32_t)fsrc[srcEnd] | ((uint32_t)fsrc[srcEnd + 1] << 8) | ((uint32_t)fsrc[srcEnd + 2] << 16) | ((uint32_t)fsrc[srcEnd + 3] << 24);
            if ((uint32_t)h != want) status = GC_ZD_CHECKSUM;
        }
    }
    if (t == 0) result[f] = produced | ((uint64_t)status << 56);
}

// ---- host: frame scan (headers only) ----
static uint32_t zd_le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ZSTD_findFrameSizeInfo (zstd_decompress.c:734) + ZSTD_getFrameHeader_advanced (:447) for every frame of the input.
// partial: an input that ends inside a frame is not an error; *consumed = end of the last whole frame (or skippable frame).
static int zd_scan(const uint8_t* src, size_t n, gc_zstd_frame* out, size_t maxFrames, size_t* nFrames, uint64_t* contentTotal, bool partial, size_t* consumed)
{
    size_t pos = 0, cnt = 0;
    uint64_t total = 0; bool known = true;
    const int truncated = partial ? GC_OK : GC_ERR_CORRUPT;
#define ZD_NEED(cond) if (!(cond)) { rc = truncated; break; }
    int rc = GC_OK;
    while (pos < n) {
        ZD_NEED(n - pos >= 4u)
        const uint32_t magic = zd_le32(src + pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                   // skippable frame
            ZD_NEED(n - pos >= 8u)
            const uint64_t sz = zd_le32(src + pos + 4);
            ZD_NEED(sz <= n - pos - 8u)
            pos += 8u + (size_t)sz; continue;
        }
        if (magic != 0xFD2FB528u) return GC_ERR_CORRUPT;
        ZD_NEED(n - pos >= 6u)
        const size_t start = pos;
        const uint32_t fhd = src[pos + 4];
        const uint32_t fcsFlag = fhd >> 6, single = (fhd >> 5) & 1u, checksum = (fhd >> 2) & 1u, didFlag = fhd & 3u;
        if (fhd & 8u) return GC_ERR_CORRUPT;                           // reserved bit
        const uint32_t didSize = didFlag == 3u ? 4u : didFlag, fcsSize = fcsFlag == 0u ? single : (fcsFlag == 1u ? 2u : (fcsFlag == 2u ? 4u : 8u));
        const size_t hdr = 5u + (single ? 0u : 1u) + didSize + fcsSize;
        ZD_NEED(n - pos >= hdr)
        size_t p = pos + 5u;
        if (!single) { if ((src[p] >> 3) + 10u > 31u) return GC_ERR_PARAM; p++; }   // ZSTD_WINDOWLOG_MAX
        uint32_t did = 0;
        for (uint32_t i = 0; i < didSize; i++) did |= (uint32_t)src[p + i] << (8u * i);
        p += didSize;
        if (did) return GC_ERR_PARAM;                                  // dictionaries are not supported
        uint64_t fcs = 0;
        for (uint32_t i = 0; i < fcsSize; i++) fcs |= (uint64_t)src[p + i] << (8u * i);
        if (fcsSize == 2u) fcs += 256u;
        p += fcsSize;
        uint64_t nb = 0;
        bool whole = false;
        for (;;) {                                                     // blocks
            if (n - p < 3u) break;
            const uint32_t h = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
            const uint32_t bt = (h >> 1) & 3u, bs = h >> 3;
            if (bt == 3u) return GC_ERR_CORRUPT;
            const size_t payload = bt == 1u ? 1u : bs;
            if (n - p - 3u < payload) break;
            p += 3u + payload; nb++;
            if (h & 1u) { whole = true; break; }
        }
        if (whole && checksum) { if (n - p < 4u) whole = false; else p += 4u; }
        ZD_NEED(whole)
        if (nb > 0xFFFFFFFFull) return GC_ERR_PARAM;
        if (out) {
            if (cnt >= maxFrames) return GC_ERR_DST_SMALL;
            gc_zstd_frame& f = out[cnt];
            f.src_off = start; f.src_size = p - start; f.dst_off = known ? total : ~0ull; f.content_size = fcs;
            f.flags = (checksum ? GC_ZD_F_CHECKSUM : 0u) | (fcsSize ? GC_ZD_F_SIZE_KNOWN : 0u); f.header_size = (uint32_t)hdr;
            f.n_blocks = (uint32_t)nb; f.reserved = 0;
        }
        if (fcsSize) total += fcs; else known = false;
        cnt++; pos = p;
    }
#undef ZD_NEED
    if (rc != GC_OK) return rc;
    *nFrames = cnt;
    if (contentTotal) *contentTotal = known ? total : ~0ull;
    if (consumed) *consumed = pos;
    return GC_OK;
}

extern "C" int gc_zstd_scan_frames(const void* src, size_t n, gc_zstd_frame* out, size_t maxFrames, size_t* nFrames, uint64_t* contentTotal)
{
    if ((!src && n) || !nFrames) return GC_ERR_PARAM;
    return zd_scan((const uint8_t*)src, n, out, maxFrames, nFrames, contentTotal, false, nullptr);
}

extern "C" int gc_zstd_scan_prefix(const void* src, size_t n, gc_zstd_frame* out, size_t maxFrames, size_t* nFrames, uint64_t* contentTotal, size_t* consumed)
{
    if ((!src && n) || !nFrames || !consumed) return GC_ERR_PARAM;
    return zd_scan((const uint8_t*)src, n, out, maxFrames, nFrames, contentTotal, true, consumed);
}

// called by gc_api.hip
extern "C" void gc_zstd_dec_launch_index(hipStream_t st, const uint8_t* src, const GcZdFrame* frames, uint32_t nFrames, GcZdBlock* blocks, uint64_t* frameTot)
{
    GC_LAUNCH(gc_zstd_dec_index_kernel, (nFrames + 63u) / 64u, 64, st, src, frames, nFrames, blocks, frameTot);
}
// literals and sequences of all blocks: independent of each other (two streams), both in front of the execution kernel
extern "C" void gc_zstd_dec_launch_literals(hipStream_t st, const uint8_t* src, uint64_t srcSize, const GcZdFrame* frames, GcZdBlock* blocks, uint32_t nBlocks, uint8_t* litWork,
                                            unsigned long long* prof, const uint32_t* order, uint32_t* ready)
{
    if (nBlocks) GC_LAUNCH(gc_zstd_dec_lit_kernel, nBlocks, 64, st, src, srcSize, frames, blocks, litWork, prof, order, ready);
}
extern "C" void gc_zstd_dec_launch_sequences(hipStream_t st, const uint8_t* src, uint64_t srcSize, const GcZdFrame* frames, GcZdBlock* blocks, uint32_t nBlocks, void* seqWork,
                                             unsigned long long* prof, const uint32_t* order, uint32_t* ready, int several)
{
    if (!nBlocks) return;
    if (several) GC_LAUNCH(gc_zstd_dec_seqv_kernel, (nBlocks + ZV_G - 1u) / ZV_G, 64, st, src, srcSize, frames, blocks, (GcU4*)seqWork, order, nBlocks, ready, prof);
    else GC_LAUNCH(gc_zstd_dec_seq_kernel, nBlocks, 64, st, src, srcSize, frames, blocks, (GcU4*)seqWork, prof, order, ready);
}
extern "C" void gc_zstd_dec_launch_exec(hipStream_t st, const uint8_t* src, uint64_t srcSize, uint8_t* dst, uint64_t dstCap, const GcZdFrame* frames, uint32_t nFrames,
                                        GcZdBlock* blocks, uint32_t* ticket, uint8_t* litWork, uint64_t litWorkSize, void* seqWork, uint64_t* result, unsigned long long* prof,
                                        const uint32_t* ready)
{
    const uint32_t wg = nFrames < GC_ZD_MAX_WG ? nFrames : GC_ZD_MAX_WG;
    GC_LAUNCH(gc_zstd_dec_exec_kernel, wg, GC_ZD_T, st, src, srcSize, dst, dstCap, frames, nFrames, (const GcZdBlock*)blocks, ticket, (const uint8_t*)litWork, litWorkSize, (GcU4*)seqWork, result, prof, ready);
}

// ---- wide execution (all blocks of all frames at once; see the kernels) ----
extern "C" void gc_zstd_dec_launch_place(hipStream_t st, const GcZdFrame* frames, uint32_t nFrames, const GcZdBlock* blocks, uint64_t dstCap, GcZdPlace* place, uint64_t* result, uint32_t* ferr)
{
    if (nFrames) GC_LAUNCH(gc_zstd_dec_place_kernel, nFrames, 64, st, frames, nFrames, blocks, dstCap, place, result, ferr);
}
extern "C" void gc_zstd_dec_launch_spread(hipStream_t st, const uint8_t* src, uint64_t srcSize, uint8_t* dst, const GcZdFrame* frames, const GcZdBlock* blocks, uint32_t nBlocks,
                                          const GcZdPlace* place, const uint8_t* litWork, uint64_t litWorkSize, const void* seqWork, uint32_t* ptr, uint64_t batchBase, uint32_t* ferr)
{
    if (nBlocks) GC_LAUNCH(gc_zstd_dec_spread_kernel, nBlocks, 256, st, src, srcSize, dst, frames, blocks, place, litWork, litWorkSize, (const GcU4*)seqWork, ptr, batchBase, ferr);
}
extern "C" void gc_zstd_dec_launch_chase(hipStream_t st, uint8_t* dstBatch, uint32_t* ptr, uint32_t n, uint32_t hops, uint8_t* pieceDone, uint32_t* counter)
{
    const uint32_t groups = (uint32_t)(((uint64_t)n + GC_ZW_WG_BYTES - 1u) / GC_ZW_WG_BYTES), perXcd = (groups + 7u) / 8u;
    if (groups) GC_LAUNCH(gc_zstd_dec_chase_kernel, perXcd * 8u, 256, st, dstBatch, ptr, n, perXcd, hops ? hops : (uint32_t)GC_ZW_HOPS, pieceDone, counter);
}
extern "C" void gc_zstd_dec_launch_finish(hipStream_t st, const uint8_t* src, const uint8_t* dst, const GcZdFrame* frames, uint32_t nFrames, uint64_t* result, const uint32_t* ferr)
{
    if (nFrames) GC_LAUNCH(gc_zstd_dec_finish_kernel, nFrames, 64, st, src, dst, frames, nFrames, result, ferr);
}
