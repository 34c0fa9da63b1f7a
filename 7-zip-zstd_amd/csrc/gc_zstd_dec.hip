// gc_zstd_dec.hip -- zstd frame decoder on the device (the decoding half of the ZSTD method, what
// NCompress::NZSTD::CDecoder::CodeSpec does with ZSTD_decompressStream, CPP/7zip/Compress/ZstdDecoder.cpp:66-240).
//
// Unit of parallelism: the frame.  Frames are independent by format (own window, own repeat offsets), and this engine's encoder
// writes one frame per 8 MiB of input (one per 128 KiB at levels 1-2), so a stream of N MiB carries N/8 frames; a stream from the
// reference's single-threaded encoder is one frame and decodes on one workgroup.  One workgroup (256 threads) takes one frame at a
// time from a ticket counter and walks its blocks in order:
//
//   stage     the compressed block (<= 128 KiB) goes into LDS -- every bit of it is read through LDS from here on
//   tables    thread 0: literals header, Huffman weights (direct or FSE-coded) -> 2^11-entry decoding table in LDS;
//             sequences header, the three FSE decoding tables (predefined / RLE / described / repeated) in LDS
//   entropy   threads 64..67 decode the 1 or 4 Huffman streams into the workgroup's literal buffer in HBM while thread 0 decodes the
//             sequence bitstream (the three interleaved FSE states are one serial chain by format), resolves the repeat offsets and
//             writes (litLength, matchLength, offset, output position) records
//   pass 1    all threads, one sequence each: literals -> the block image in LDS (which replaces the staged input), and every match
//             whose source lies in front of the block (copied from the frame's output in HBM)
//   pass 2    wave 0, sequence by sequence in order, 64 bytes per step: the matches that read the block itself
//             (offset < 64: dst[k] = src[k mod offset], so overlapping copies are parallel as well)
//   flush     the block image -> HBM, fence, next block
//   checksum  XXH64 of the content (4 lanes = the 4 accumulators) when the frame carries one
//
// Restated from the reference decoder (the format is normative, every rule has to match):
//   frame header, block headers        ZSTD_getFrameHeader_advanced zstd_decompress.c:447, ZSTD_decompressFrame :953, ZSTD_findFrameSizeInfo :734
//   literals section                   ZSTD_decodeLiteralsBlock zstd_decompress_block.c:134-340
//   Huffman weights / table / streams  HUF_readStats entropy_common.c:234, HUF_readDTableX1_wksp huf_decompress.c:385, HUF_decompress4X1 :602
//   NCount, FSE decoding tables        FSE_readNCount_body entropy_common.c:42, ZSTD_buildFSETable_body zstd_decompress_block.c:485
//   sequences header / decode / exec   ZSTD_decodeSeqHeaders :695, ZSTD_decodeSequence :1229, ZSTD_execSequence :1001
//   content checksum                   XXH64 (xxhash.h), zstd_decompress.c:1034-1056
// Not supported: dictionaries (a frame with a dictionary id is refused), legacy (v0.x) frames.
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
#define ZD_PAD 32u

// shared scalars of the workgroup
enum { ZV_ERR = 0, ZV_FRAME, ZV_BTYPE, ZV_BSIZE, ZV_LAST, ZV_LITKIND, ZV_LITREGEN, ZV_LITOFF, ZV_NSTREAMS, ZV_HUFLOG, ZV_HUFOK,
       ZV_SEQPOS, ZV_NSEQ, ZV_LLLOG, ZV_OFLOG, ZV_MLLOG, ZV_FSEOK, ZV_OUTSIZE, ZV_LITEND, ZV_DPOSEND, ZV_STR0, ZV_STR1 = ZV_STR0 + 2, ZV_STR2 = ZV_STR0 + 4,
       ZV_STR3 = ZV_STR0 + 6, ZV_RLEBYTE = ZV_STR0 + 8, ZV_REP0, ZV_REP1, ZV_REP2, ZV_COUNT };
// literal kinds of a block
enum { ZL_RAW = 0, ZL_RLE = 1, ZL_HUF = 2 };

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

// ---- NCount (forward, LSB first).  Returns the bytes used, 0 on error. ----
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

// ---- FSE decoding table: tab[state] = newStateBase | nbBits << 16 | symbol << 24 ----
__device__ bool zd_fse_build(uint32_t* tab, const int16_t* norm, uint32_t maxSym, uint32_t log, uint16_t* symNext)
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
        tab[u] = (((next << nb) - size) & 0xFFFFu) | (nb << 16) | (s << 24);
    }
    return true;
}

// ---- Huffman tree description -> sHuf[2^log] = symbol | nbBits << 8.  Returns the bytes used, 0 on error. ----
__device__ uint32_t zd_huf_read(const uint8_t* p, uint32_t n, uint16_t* sHuf, uint8_t* sW, int16_t* sNorm, uint16_t* sNext, uint32_t* sFseW, uint32_t* logOut)
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
            uint32_t e = sFseW[s1];
            sW[nw++] = (uint8_t)(e >> 24);
            s1 = (e & 0xFFFFu) + zd_br_read(r, (e >> 16) & 0xFFu);
            if (r.off < 0) { sW[nw++] = (uint8_t)(sFseW[s2] >> 24); break; }
            if (nw > 253u) return 0;
            e = sFseW[s2];
            sW[nw++] = (uint8_t)(e >> 24);
            s2 = (e & 0xFFFFu) + zd_br_read(r, (e >> 16) & 0xFFu);
            if (r.off < 0) { sW[nw++] = (uint8_t)(sFseW[s1] >> 24); break; }
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

// one Huffman stream -> count bytes.  false = the stream does not end exactly where the symbols do
__device__ bool zd_huf_stream(const uint8_t* p, uint32_t n, const uint16_t* sHuf, uint32_t log, uint8_t* out, uint32_t count)
{
    ZdBR r;
    if (!zd_br_init(r, p, n)) return false;
    for (uint32_t i = 0; i < count; i++) {
        const uint32_t e = sHuf[zd_br_peek(r, log)];
        out[i] = (uint8_t)e;
        r.off -= (int32_t)(e >> 8);
    }
    return r.off == 0;
}

// ---- one of the three symbol tables of the sequences section.  Returns the bytes used (0 is valid for predefined / repeat), -1 on error ----
__device__ int zd_seq_table(uint32_t mode, const uint8_t* p, uint32_t n, uint32_t* tab, uint32_t* logVar, uint32_t maxSym, uint32_t maxLog,
                            const int16_t* defNorm, uint32_t defMaxSym, uint32_t defLog, bool haveOld, int16_t* sNorm, uint16_t* sNext)
{
    if (mode == 0u) {
        for (uint32_t s = 0; s <= defMaxSym; s++) sNorm[s] = defNorm[s];
        if (!zd_fse_build(tab, sNorm, defMaxSym, defLog, sNext)) return -1;
        *logVar = defLog; return 0;
    }
    if (mode == 1u) {
        if (n < 1u || p[0] > maxSym) return -1;
        tab[0] = (uint32_t)p[0] << 24; *logVar = 0; return 1;
    }
    if (mode == 2u) {
        uint32_t ms = 0, log = 0;
        const uint32_t h = zd_read_ncount(p, n, sNorm, maxSym, maxLog, &ms, &log);
        if (!h) return -1;
        if (!zd_fse_build(tab, sNorm, ms, log, sNext)) return -1;
        *logVar = log; return (int)h;
    }
    return haveOld ? 0 : -1;
}

__device__ __forceinline__ uint64_t zd_rotl(uint64_t v, uint32_t r) { return (v << r) | (v >> (64u - r)); }
#define XP1 0x9E3779B185EBCA87ull
#define XP2 0xC2B2AE3D27D4EB4Full
#define XP3 0x165667B19E3779F9ull
#define XP4 0x85EBCA77C2B2AE63ull
#define XP5 0x27D4EB2F165667C5ull
__device__ __forceinline__ uint64_t zd_xround(uint64_t acc, uint64_t in) { return zd_rotl(acc + in * XP2, 31) * XP1; }

extern "C" __global__ void __launch_bounds__(GC_ZD_T)
gc_zstd_dec_kernel(const uint8_t* __restrict__ src, uint8_t* dst, uint64_t dstCap, const GcZdFrame* __restrict__ frames, uint32_t nFrames,
                   uint32_t* ticket, uint8_t* litWork, GcU4* seqWork, uint32_t* lposWork, uint64_t* result)
{
    __shared__ __attribute__((aligned(16))) uint8_t sOut[GC_ZSTD_BLOCK_MAX + ZD_PAD];
    __shared__ uint16_t sHuf[1u << ZD_HUF_LOG_MAX];
    __shared__ uint32_t sLL[512], sOF[256], sML[512], sFseW[64];
    __shared__ int16_t sNorm[256];
    __shared__ uint16_t sNext[256];
    __shared__ uint8_t sW[256];
    __shared__ uint32_t sV[ZV_COUNT];
    __shared__ uint64_t sAcc[4];

    const uint32_t t = threadIdx.x, lane = t & 63u;
    uint8_t* const lit = litWork + (size_t)blockIdx.x * GC_ZD_LIT_STRIDE;
    GcU4* const seq = seqWork + (size_t)blockIdx.x * GC_ZD_MAX_SEQ;
    uint32_t* const lposA = lposWork + (size_t)blockIdx.x * GC_ZD_MAX_SEQ;

    for (;;) {
        if (t == 0) sV[ZV_FRAME] = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t f = sV[ZV_FRAME];
        __syncthreads();
        if (f >= nFrames) return;
        const GcZdFrame fr = frames[f];
        const uint8_t* const fsrc = src + fr.srcOff;
        uint8_t* const fdst = dst + fr.dstOff;
        const uint64_t cap = (fr.flags & GC_ZD_F_SIZE_KNOWN) ? fr.contentSize : (dstCap - fr.dstOff);
        const uint64_t srcEnd = fr.srcSize - ((fr.flags & GC_ZD_F_CHECKSUM) ? 4u : 0u);      // end of the blocks
        uint64_t ip = fr.hdrSize;          // frame-relative read position (uniform)
        uint64_t produced = 0;
        if (t == 0) { sV[ZV_ERR] = GC_ZD_OK; sV[ZV_HUFOK] = 0; sV[ZV_FSEOK] = 0; sV[ZV_REP0] = 1; sV[ZV_REP1] = 4; sV[ZV_REP2] = 8; }
        __syncthreads();

        for (;;) {      // blocks
            if (t == 0) {
                if (ip + 3u > srcEnd) sV[ZV_ERR] = GC_ZD_CORRUPT;
                else {
                    const uint32_t h = (uint32_t)fsrc[ip] | ((uint32_t)fsrc[ip + 1] << 8) | ((uint32_t)fsrc[ip + 2] << 16);
                    const uint32_t bt = (h >> 1) & 3u, bs = h >> 3;
                    sV[ZV_LAST] = h & 1u; sV[ZV_BTYPE] = bt; sV[ZV_BSIZE] = bs;
                    const uint64_t payload = bt == 1u ? 1u : bs;
                    if (bt == 3u || bs > GC_ZSTD_BLOCK_MAX || ip + 3u + payload > srcEnd) sV[ZV_ERR] = GC_ZD_CORRUPT;
                    else if (bt != 2u && produced + bs > cap) sV[ZV_ERR] = GC_ZD_DST_SMALL;
                }
            }
            __syncthreads();
            if (sV[ZV_ERR]) break;
            const uint32_t bt = sV[ZV_BTYPE], bs = sV[ZV_BSIZE], lastBlock = sV[ZV_LAST];
            const uint8_t* const bsrc = fsrc + ip + 3u;
            uint8_t* const bdst = fdst + produced;
            if (bt == 0u) {                                       // raw
                for (uint32_t i = t; i < bs; i += GC_ZD_T) bdst[i] = bsrc[i];
                ip += 3u + bs; produced += bs;
            } else if (bt == 1u) {                                // RLE
                const uint8_t b = bsrc[0];
                for (uint32_t i = t; i < bs; i += GC_ZD_T) bdst[i] = b;
                ip += 4u; produced += bs;
            } else {
                // ---- stage the compressed block ----
                for (uint32_t i = t * 8u; i < bs; i += GC_ZD_T * 8u) {
                    if (i + 8u <= bs) { const uint64_t v = gc_ld64(bsrc + i); __builtin_memcpy(sOut + i, &v, 8); }
                    else for (uint32_t k = i; k < bs; k++) sOut[k] = bsrc[k];
                }
                for (uint32_t i = t; i < ZD_PAD; i += GC_ZD_T) if (bs + i < GC_ZSTD_BLOCK_MAX + ZD_PAD) sOut[bs + i] = 0;
                __syncthreads();
                // ---- literals header + Huffman table ----
                if (t == 0) {
                    uint32_t err = 0;
                    if (bs < 1u) err = 1;       // a compressed block holds at least the two section headers (2 bytes); checked as we go
                    uint32_t hdr = 0, regen = 0, comp = 0, nStreams = 1, kind = ZL_RAW;
                    if (!err) {
                        const uint32_t b0 = sOut[0], lt = b0 & 3u, sf = (b0 >> 2) & 3u;
                        if (lt < 2u) {
                            if (!(sf & 1u)) { hdr = 1; regen = b0 >> 3; }
                            else if (sf == 1u) { hdr = 2; regen = ((uint32_t)sOut[0] | ((uint32_t)sOut[1] << 8)) >> 4; }
                            else { hdr = 3; regen = ((uint32_t)sOut[0] | ((uint32_t)sOut[1] << 8) | ((uint32_t)sOut[2] << 16)) >> 4; }
                            kind = lt == 0u ? ZL_RAW : ZL_RLE;
                            comp = lt == 0u ? regen : 1u;
                        } else {
                            const uint64_t v = gc_ld64(sOut);
                            if (sf < 2u) { hdr = 3; regen = (uint32_t)(v >> 4) & 0x3FFu; comp = (uint32_t)(v >> 14) & 0x3FFu; nStreams = sf == 0u ? 1u : 4u; }
                            else if (sf == 2u) { hdr = 4; regen = (uint32_t)(v >> 4) & 0x3FFFu; comp = (uint32_t)(v >> 18) & 0x3FFFu; nStreams = 4; }
                            else { hdr = 5; regen = (uint32_t)(v >> 4) & 0x3FFFFu; comp = (uint32_t)(v >> 22) & 0x3FFFFu; nStreams = 4; }
                            kind = ZL_HUF;
                            if (lt == 3u && !sV[ZV_HUFOK]) err = 1;          // treeless block without a table from an earlier block
                            if (regen == 0u) err = 1;
                        }
                        if (hdr > bs || regen > GC_ZSTD_BLOCK_MAX || (uint64_t)hdr + comp > bs) err = 1;
                        if (!err && kind == ZL_HUF) {
                            uint32_t treeBytes = 0;
                            if (lt == 2u) {
                                uint32_t log = 0;
                                treeBytes = zd_huf_read(sOut + hdr, comp, sHuf, sW, sNorm, sNext, sFseW, &log);
                                if (!treeBytes) err = 1; else { sV[ZV_HUFLOG] = log; sV[ZV_HUFOK] = 1; }
                            }
                            if (!err) {
                                const uint32_t base = hdr + treeBytes, avail = comp - treeBytes;
                                if (nStreams == 1u) { sV[ZV_STR0] = base; sV[ZV_STR0 + 1] = avail; if (!avail) err = 1; }
                                else {
                                    if (avail < 10u) err = 1;
                                    else {
                                        const uint32_t s1 = (uint32_t)sOut[base] | ((uint32_t)sOut[base + 1] << 8), s2 = (uint32_t)sOut[base + 2] | ((uint32_t)sOut[base + 3] << 8),
                                                       s3 = (uint32_t)sOut[base + 4] | ((uint32_t)sOut[base + 5] << 8);
                                        if ((uint64_t)6u + s1 + s2 + s3 >= avail) err = 1;
                                        else {
                                            sV[ZV_STR0] = base + 6u; sV[ZV_STR0 + 1] = s1;
                                            sV[ZV_STR1] = base + 6u + s1; sV[ZV_STR1 + 1] = s2;
                                            sV[ZV_STR2] = base + 6u + s1 + s2; sV[ZV_STR2 + 1] = s3;
                                            sV[ZV_STR3] = base + 6u + s1 + s2 + s3; sV[ZV_STR3 + 1] = avail - 6u - s1 - s2 - s3;
                                        }
                                    }
                                }
                            }
                        }
                        if (kind == ZL_RLE) sV[ZV_RLEBYTE] = sOut[hdr];
                    }
                    sV[ZV_LITKIND] = kind; sV[ZV_LITREGEN] = regen; sV[ZV_LITOFF] = hdr; sV[ZV_NSTREAMS] = nStreams;
                    sV[ZV_SEQPOS] = hdr + comp;
                    if (err) sV[ZV_ERR] = GC_ZD_CORRUPT;
                }
                __syncthreads();
                if (sV[ZV_ERR]) break;
                // ---- entropy stage: Huffman streams (threads 64..67) beside the sequences (thread 0) ----
                if (t >= 64u && t < 64u + sV[ZV_NSTREAMS] && sV[ZV_LITKIND] == ZL_HUF) {
                    const uint32_t q = t - 64u, regen = sV[ZV_LITREGEN], ns = sV[ZV_NSTREAMS];
                    const uint32_t seg = ns == 1u ? regen : (regen + 3u) >> 2;
                    const uint32_t o = q * seg;
                    bool ok = true;
                    uint32_t cnt = 0;
                    if (ns == 1u) cnt = regen;
                    else if (o > regen) ok = false;
                    else cnt = q == 3u ? regen - o : (seg <= regen - o ? seg : 0xFFFFFFFFu);
                    if (cnt == 0xFFFFFFFFu) ok = false;
                    if (ok) ok = zd_huf_stream(sOut + sV[ZV_STR0 + 2u * q], sV[ZV_STR0 + 2u * q + 1u], sHuf, sV[ZV_HUFLOG], lit + o, cnt);
                    if (!ok) sV[ZV_ERR] = GC_ZD_CORRUPT;
                }
                if (t == 0) {
                    uint32_t err = 0;
                    const uint32_t sp = sV[ZV_SEQPOS], regen = sV[ZV_LITREGEN];
                    uint32_t p = sp, nSeq = 0;
                    if (p >= bs) err = 1;
                    else {
                        const uint32_t b0 = sOut[p++];
                        if (b0 < 128u) nSeq = b0;
                        else if (b0 < 255u) { if (p >= bs) err = 1; else nSeq = ((b0 - 128u) << 8) + sOut[p++]; }
                        else { if (p + 2u > bs) err = 1; else { nSeq = (uint32_t)sOut[p] + ((uint32_t)sOut[p + 1] << 8) + 0x7F00u; p += 2u; } }
                    }
                    uint32_t dpos = 0, lpos = 0;
                    if (!err && nSeq == 0u) { if (p != bs) err = 1; }
                    else if (!err) {
                        if (p >= bs) err = 1;
                        else {
                            const uint32_t modes = sOut[p++];
                            if (modes & 3u) err = 1;
                            const bool old = sV[ZV_FSEOK] != 0u;
                            int h;
                            if (!err) { h = zd_seq_table(modes >> 6, sOut + p, bs - p, sLL, &sV[ZV_LLLOG], 35u, 9u, kZdLLNorm, 35u, 6u, old, sNorm, sNext); if (h < 0) err = 1; else p += (uint32_t)h; }
                            if (!err) { h = zd_seq_table((modes >> 4) & 3u, sOut + p, bs - p, sOF, &sV[ZV_OFLOG], 31u, 8u, kZdOFNorm, 28u, 5u, old, sNorm, sNext); if (h < 0) err = 1; else p += (uint32_t)h; }
                            if (!err) { h = zd_seq_table((modes >> 2) & 3u, sOut + p, bs - p, sML, &sV[ZV_MLLOG], 52u, 9u, kZdMLNorm, 52u, 6u, old, sNorm, sNext); if (h < 0) err = 1; else p += (uint32_t)h; }
                            if (!err) sV[ZV_FSEOK] = 1;
                        }
                        ZdBR r;
                        if (!err && (p >= bs || !zd_br_init(r, sOut + p, bs - p))) err = 1;
                        if (!err) {
                            const uint32_t llLog = sV[ZV_LLLOG], ofLog = sV[ZV_OFLOG], mlLog = sV[ZV_MLLOG];
                            uint32_t stLL = zd_br_read(r, llLog), stOF = zd_br_read(r, ofLog), stML = zd_br_read(r, mlLog);
                            uint32_t rep0 = sV[ZV_REP0], rep1 = sV[ZV_REP1], rep2 = sV[ZV_REP2];
                            const uint64_t frameBase = produced;              // bytes of the frame in front of this block
                            for (uint32_t j = 0; j < nSeq; j++) {
                                const uint32_t eLL = sLL[stLL], eOF = sOF[stOF], eML = sML[stML];
                                const uint32_t llc = eLL >> 24, ofc = eOF >> 24, mlc = eML >> 24;
                                if (ofc > 31u) { err = 1; break; }
                                const uint32_t ofv = (1u << ofc) + zd_br_read(r, ofc);
                                const uint32_t ml = kZdMLBase[mlc] + zd_br_read(r, kZdMLBits[mlc]);
                                const uint32_t ll = kZdLLBase[llc] + zd_br_read(r, kZdLLBits[llc]);
                                uint32_t off;
                                if (ofv > 3u) { off = ofv - 3u; rep2 = rep1; rep1 = rep0; rep0 = off; }
                                else {
                                    const uint32_t idx = ofv - 1u + (ll == 0u ? 1u : 0u);
                                    if (idx == 0u) off = rep0;
                                    else {
                                        off = idx == 1u ? rep1 : (idx == 2u ? rep2 : rep0 - 1u);
                                        if (idx != 1u) rep2 = rep1;
                                        rep1 = rep0; rep0 = off;
                                    }
                                }
                                if (j + 1u < nSeq) {
                                    stLL = (eLL & 0xFFFFu) + zd_br_read(r, (eLL >> 16) & 0xFFu);
                                    stML = (eML & 0xFFFFu) + zd_br_read(r, (eML >> 16) & 0xFFu);
                                    stOF = (eOF & 0xFFFFu) + zd_br_read(r, (eOF >> 16) & 0xFFu);
                                }
                                if (r.off < 0) { err = 1; break; }
                                if ((uint64_t)lpos + ll > regen || (uint64_t)dpos + ll + ml > GC_ZSTD_BLOCK_MAX) { err = 1; break; }
                                if (off == 0u || (uint64_t)off > frameBase + dpos + ll) { err = 1; break; }
                                GcU4 rec; rec.x = ll; rec.y = ml; rec.z = off; rec.w = dpos;
                                seq[j] = rec; lposA[j] = lpos;
                                dpos += ll + ml; lpos += ll;
                            }
                            if (!err && r.off != 0) err = 1;
                            sV[ZV_REP0] = rep0; sV[ZV_REP1] = rep1; sV[ZV_REP2] = rep2;
                        }
                    }
                    const uint32_t outSize = dpos + (regen - lpos);
                    if (!err && outSize > GC_ZSTD_BLOCK_MAX) err = 1;
                    sV[ZV_NSEQ] = nSeq; sV[ZV_LITEND] = lpos; sV[ZV_DPOSEND] = dpos; sV[ZV_OUTSIZE] = outSize;
                    if (err) sV[ZV_ERR] = GC_ZD_CORRUPT;
                    else if (produced + outSize > cap) sV[ZV_ERR] = GC_ZD_DST_SMALL;
                }
                __threadfence();
                __syncthreads();
                if (sV[ZV_ERR]) break;
                // ---- pass 1: literals and the matches that lie in front of the block ----
                const uint32_t nSeq = sV[ZV_NSEQ], kind = sV[ZV_LITKIND], outSize = sV[ZV_OUTSIZE];
                const uint8_t* const litSrc = kind == ZL_HUF ? lit : bsrc + sV[ZV_LITOFF];
                const uint8_t rleByte = (uint8_t)sV[ZV_RLEBYTE];
                for (uint32_t j = t; j < nSeq; j += GC_ZD_T) {
                    const GcU4 rec = seq[j];
                    const uint32_t lp = lposA[j];
                    if (kind == ZL_RLE) for (uint32_t k = 0; k < rec.x; k++) sOut[rec.w + k] = rleByte;
                    else for (uint32_t k = 0; k < rec.x; k++) sOut[rec.w + k] = litSrc[lp + k];
                    const uint32_t d = rec.w + rec.x;
                    if (rec.z >= d + rec.y) {                      // the whole source lies in front of the block
                        const uint8_t* const ms = bdst + d - rec.z;
                        for (uint32_t k = 0; k < rec.y; k++) sOut[d + k] = ms[k];
                    }
                }
                {
                    const uint32_t lp = sV[ZV_LITEND], dp = sV[ZV_DPOSEND], rest = sV[ZV_LITREGEN] - lp;
                    if (kind == ZL_RLE) for (uint32_t k = t; k < rest; k += GC_ZD_T) sOut[dp + k] = rleByte;
                    else for (uint32_t k = t; k < rest; k += GC_ZD_T) sOut[dp + k] = litSrc[lp + k];
                }
                __syncthreads();
                // ---- pass 2: matches that read the block itself, in order ----
                if (t < 64u) {
                    for (uint32_t j0 = 0; j0 < nSeq; j0 += 64u) {
                        GcU4 mine; mine.x = mine.y = mine.z = mine.w = 0;
                        if (j0 + lane < nSeq) mine = seq[j0 + lane];
                        const uint32_t cnt = nSeq - j0 < 64u ? nSeq - j0 : 64u;
                        for (uint32_t i = 0; i < cnt; i++) {
                            const uint32_t ll = __shfl(mine.x, (int)i), ml = __shfl(mine.y, (int)i), off = __shfl(mine.z, (int)i), dp = __shfl(mine.w, (int)i);
                            const uint32_t d = dp + ll;
                            if (off >= d + ml) continue;             // done in pass 1
                            const int32_t s0 = (int32_t)d - (int32_t)off;
                            if (off < 64u) {
                                for (uint32_t k = lane; k < ml; k += 64u) {
                                    const int32_t si = s0 + (int32_t)(k % off);
                                    sOut[d + k] = si < 0 ? bdst[si] : sOut[si];
                                }
                                gc_wave_step();
                            } else {
                                for (uint32_t c = 0; c < ml; c += 64u) {
                                    const uint32_t k = c + lane;
                                    if (k < ml) { const int32_t si = s0 + (int32_t)k; sOut[d + k] = si < 0 ? bdst[si] : sOut[si]; }
                                    gc_wave_step();
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                // ---- flush ----
                for (uint32_t i = t * 16u; i < outSize; i += GC_ZD_T * 16u) {
                    if (i + 16u <= outSize) { GcU4 v; __builtin_memcpy(&v, sOut + i, 16); __builtin_memcpy(bdst + i, &v, 16); }
                    else for (uint32_t k = i; k < outSize; k++) bdst[k] = sOut[k];
                }
                ip += 3u + bs; produced += outSize;
            }
            __threadfence();
            __syncthreads();
            if (lastBlock) break;
        }
        __syncthreads();
        uint32_t status = sV[ZV_ERR];
        if (!status && ip != srcEnd) status = GC_ZD_CORRUPT;
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
                sV[ZV_ERR] = (uint32_t)h == want ? GC_ZD_OK : GC_ZD_CHECKSUM;
            }
            __syncthreads();
            status = sV[ZV_ERR];
        }
        if (t == 0) result[f] = produced | ((uint64_t)status << 56);
        __syncthreads();
    }
}

// ---- host: frame scan (headers only) ----
static uint32_t zd_le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ZSTD_findFrameSizeInfo (zstd_decompress.c:734) + ZSTD_getFrameHeader_advanced (:447) for every frame of the input
extern "C" int gc_zstd_scan_frames(const void* srcv, size_t n, gc_zstd_frame* out, size_t maxFrames, size_t* nFrames, uint64_t* contentTotal)
{
    if ((!srcv && n) || !nFrames) return GC_ERR_PARAM;
    const uint8_t* src = (const uint8_t*)srcv;
    size_t pos = 0, cnt = 0;
    uint64_t total = 0; bool known = true;
    while (pos < n) {
        if (n - pos < 8u) return GC_ERR_CORRUPT;
        const uint32_t magic = zd_le32(src + pos);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {                   // skippable frame
            const uint64_t sz = zd_le32(src + pos + 4);
            if (sz > n - pos - 8u) return GC_ERR_CORRUPT;
            pos += 8u + (size_t)sz; continue;
        }
        if (magic != 0xFD2FB528u) return GC_ERR_CORRUPT;
        const size_t start = pos;
        const uint32_t fhd = src[pos + 4];
        const uint32_t fcsFlag = fhd >> 6, single = (fhd >> 5) & 1u, checksum = (fhd >> 2) & 1u, didFlag = fhd & 3u;
        if (fhd & 8u) return GC_ERR_CORRUPT;                           // reserved bit
        const uint32_t didSize = didFlag == 3u ? 4u : didFlag, fcsSize = fcsFlag == 0u ? single : (fcsFlag == 1u ? 2u : (fcsFlag == 2u ? 4u : 8u));
        const size_t hdr = 5u + (single ? 0u : 1u) + didSize + fcsSize;
        if (n - pos < hdr) return GC_ERR_CORRUPT;
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
        for (;;) {                                                     // blocks
            if (n - p < 3u) return GC_ERR_CORRUPT;
            const uint32_t h = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16);
            const uint32_t bt = (h >> 1) & 3u, bs = h >> 3;
            if (bt == 3u) return GC_ERR_CORRUPT;
            const size_t payload = bt == 1u ? 1u : bs;
            if (n - p - 3u < payload) return GC_ERR_CORRUPT;
            p += 3u + payload;
            if (h & 1u) break;
        }
        if (checksum) { if (n - p < 4u) return GC_ERR_CORRUPT; p += 4u; }
        if (out) {
            if (cnt >= maxFrames) return GC_ERR_DST_SMALL;
            gc_zstd_frame& f = out[cnt];
            f.src_off = start; f.src_size = p - start; f.dst_off = known ? total : ~0ull; f.content_size = fcs;
            f.flags = (checksum ? GC_ZD_F_CHECKSUM : 0u) | (fcsSize ? GC_ZD_F_SIZE_KNOWN : 0u); f.header_size = (uint32_t)hdr;
        }
        if (fcsSize) total += fcs; else known = false;
        cnt++; pos = p;
    }
    *nFrames = cnt;
    if (contentTotal) *contentTotal = known ? total : ~0ull;
    return GC_OK;
}

// called by gc_api.hip
extern "C" void gc_zstd_dec_launch(hipStream_t st, uint32_t grid, const uint8_t* src, uint8_t* dst, uint64_t dstCap, const GcZdFrame* frames, uint32_t nFrames,
                                   uint32_t* ticket, uint8_t* litWork, void* seqWork, uint32_t* lposWork, uint64_t* result)
{
    GC_LAUNCH(gc_zstd_dec_kernel, grid, GC_ZD_T, st, src, dst, dstCap, frames, nFrames, ticket, litWork, (GcU4*)seqWork, lposWork, result);
}
