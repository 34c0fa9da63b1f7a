// gc_bra.hip -- branch converters ("BCJ" pre-filters of executables) on data that lies in HBM (the 7z folder pipeline runs them in
// front of the compressor, CPP/7zip/Compress/BranchMisc.cpp:21-26 -> C/Bra.c).  They turn the relative targets of CALL instructions into absolute
// ones so that repeated calls of one function become repeated byte strings.
//
// The five converters built here are the ones whose decision is local: ARM64 (BL and ADRP), ARM (BL), PPC (bl), SPARC (call) look at one
// aligned 4-byte word (C/Bra.c:75-257); ARMT (Thumb BL) looks at a pair of 16-bit units, and since the second unit of a pair (top bits 11111) can
// never be the first unit of one (11110), pairs never overlap and every pair is decided on its own as well (C/Bra.c:260-340).  One thread per
// 16 bytes, one 16-byte load and one 16-byte store: 2 bytes of HBM traffic per byte, which is the algorithmic minimum.  IA64 and RISCV are not built.
//
// X86 (C/Bra86.c, "BCJ") is a state machine: whether an E8 / E9 byte is taken as CALL / JMP depends on the E8 / E9 bytes among the three bytes in
// front of it (the mask) and on whether it lies inside the operand of a converted instruction (those four bytes are skipped).  It is made parallel
// by restart points: an E8 / E9 byte with no E8 / E9 byte among the 7 bytes in front of it is reached with the same state (mask 0, not inside an
// operand) whatever came before.  Every lane owns 512 bytes, starts at the first restart point inside them (lane 0: at byte 0 with the caller's
// state) and runs the reference's control flow, restated, until it arrives at the first restart point at or behind the end of its bytes -- which
// is where the next lane that found one has started.  Decisions read the source only, conversions go to a copy, so the lanes do not see each
// other's writes; the one lane that reaches the end of the buffer reports the processed size and the state exactly as the reference's single
// pass does.  (A buffer without restart points -- kilobytes of E8 bytes -- degenerates to one lane; correct, slow.)
//
// Restated from the reference (same arithmetic, so encode and decode are bit-exact against C/Bra.c; `pc` = the virtual address of byte 0).
#include "gpucodec.h"
#include "gc_device.h"
#ifdef HIPEMU
#include "hip_runtime_stub.h"
#else
#include <hip/hip_runtime.h>
#define GC_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif

__device__ __forceinline__ uint32_t bra_bswap(uint32_t v) { return (v >> 24) | ((v >> 8) & 0xFF00u) | ((v << 8) & 0xFF0000u) | (v << 24); }

// one aligned word at byte offset o (v as loaded little-endian); returns the converted word
__device__ __forceinline__ uint32_t bra_word(uint32_t v, uint32_t kind, uint32_t pcAt, bool enc)
{
    if (kind == GC_BRA_ARM64) {
        if (((v - 0x94000000u) & 0xfc000000u) == 0u) {                 // BL: 26-bit word offset
            const uint32_t c = pcAt >> 2;
            v = enc ? v + c : v - c;
            return (v & 0x03ffffffu) | 0x94000000u;
        }
        const uint32_t flag = 1u << 20, mask = (1u << 24) - (flag << 1);
        uint32_t t = v - 0x90000000u;
        if ((t & 0x9f000000u) != 0u) return v;                          // not ADRP
        t += flag;
        if (t & mask) return v;                                         // only page offsets within +-1 GiB are converted
        uint32_t z = (t & 0xffffffe0u) | (t >> 26);
        const uint32_t c = (pcAt >> (12 - 3)) & ~7u;
        z = enc ? z + c : z - c;
        t &= 0x1fu; t |= 0x90000000u; t |= z << 26;
        t |= 0x00ffffe0u & ((z & ((flag << 1) - 1u)) - flag);
        return t;
    }
    if (kind == GC_BRA_ARM) {
        if ((v >> 24) != 0xebu) return v;                               // BL, always
        const uint32_t c = (pcAt + 8u) >> 2;
        v = enc ? v + c : v - c;
        return (v & 0x00ffffffu) | 0xeb000000u;
    }
    if (kind == GC_BRA_PPC) {
        uint32_t b = bra_bswap(v);
        if ((b & 0xfc000003u) != 0x48000001u) return v;                 // bl
        b = enc ? b + pcAt : b - pcAt;
        return bra_bswap((b & 0x03ffffffu) | 0x48000000u);
    }
    // SPARC: call with a displacement whose top bits are all 0 or all 1
    const uint32_t flag = 1u << 22;
    uint32_t t = bra_bswap(v);
    t += 5u << 29; t ^= 7u << 29; t += flag;
    if ((t & (0u - (flag << 1))) != 0u) return v;
    t <<= 2;
    t = enc ? t + pcAt : t - pcAt;
    t &= (flag << 3) - 1u;
    t -= flag << 2;
    t >>= 2;
    t |= 1u << 30;
    return bra_bswap(t);
}

extern "C" __global__ void __launch_bounds__(256)
gc_bra_kernel(const uint8_t* __restrict__ src, uint8_t* dst, uint64_t n, uint32_t pc, uint32_t kind, uint32_t encoding, uint32_t* lastPair)
{
    const uint64_t o = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 16u;
    if (o >= n) return;
    const bool enc = encoding != 0u;
    if (kind != GC_BRA_ARMT) {
        const uint64_t lim = n & ~3ull;
        if (o + 16u <= lim) {
            GcU4 v; __builtin_memcpy(&v, src + o, 16);
            v.x = bra_word(v.x, kind, pc + (uint32_t)o, enc); v.y = bra_word(v.y, kind, pc + (uint32_t)o + 4u, enc);
            v.z = bra_word(v.z, kind, pc + (uint32_t)o + 8u, enc); v.w = bra_word(v.w, kind, pc + (uint32_t)o + 12u, enc);
            __builtin_memcpy(dst + o, &v, 16);
        } else {
            for (uint64_t k = o; k < n && k < o + 16u; k += 4u) {
                if (k + 4u <= lim) { uint32_t v = gc_ld32(src + k); v = bra_word(v, kind, pc + (uint32_t)k, enc); __builtin_memcpy(dst + k, &v, 4); }
                else for (uint64_t b = k; b < n && b < k + 4u; b++) dst[b] = src[b];
            }
        }
        return;
    }
    // ARMT: 16-bit units h[0..7] of my 16 bytes, with one unit of context on either side.  A pair starts at unit j iff
    // (h[j] >> 11) == 0x1E and (h[j+1] >> 11) == 0x1F and the pair lies inside the processed range (byte offset <= size' - 4).
    const uint64_t sz = n & ~1ull;
    const bool whole = o + 16u <= sz;                      // all eight units of my chunk exist: one 16-byte load, one 16-byte store
    uint32_t h[10];
    if (whole) {
        GcU4 v; __builtin_memcpy(&v, src + o, 16);
        h[1] = v.x & 0xFFFFu; h[2] = v.x >> 16; h[3] = v.y & 0xFFFFu; h[4] = v.y >> 16; h[5] = v.z & 0xFFFFu; h[6] = v.z >> 16; h[7] = v.w & 0xFFFFu; h[8] = v.w >> 16;
        h[0] = o ? ((uint32_t)src[o - 2u] | ((uint32_t)src[o - 1u] << 8)) : 0u;
        h[9] = o + 18u <= sz ? ((uint32_t)src[o + 16u] | ((uint32_t)src[o + 17u] << 8)) : 0u;
    } else {
        for (int j = 0; j < 10; j++) {
            const uint64_t b = o + 2u * (uint64_t)j - 2u;    // (wraps for j = 0 of the first chunk: excluded by the test)
            h[j] = (j == 0 && o == 0u) || b + 2u > sz ? 0u : ((uint32_t)src[b] | ((uint32_t)src[b + 1u] << 8));
        }
    }
    uint32_t outw[8];
    for (int j = 1; j <= 8; j++) {
        const uint64_t b = o + 2u * (uint64_t)j - 2u;        // byte offset of unit j
        uint32_t out = h[j];
        if (b + 2u <= sz) {
            const bool first = (h[j] >> 11) == 0x1Eu && (h[j + 1] >> 11) == 0x1Fu && b + 4u <= sz;
            const bool second = (h[j - 1] >> 11) == 0x1Eu && (h[j] >> 11) == 0x1Fu && b >= 2u;
            if (first || second) {
                const uint64_t pb = first ? b : b - 2u;       // byte offset of the pair
                const uint32_t h0 = first ? h[j] : h[j - 1], h1 = first ? h[j + 1] : h[j];
                uint32_t v = (h0 << 11) | (h1 & 0x7FFu);
                const uint32_t c = (pc + (uint32_t)pb + 4u) >> 1;
                v = enc ? v + c : v - c;
                out = first ? (((v >> 11) & 0x7ffu) | 0xf000u) : ((v | 0xf800u) & 0xFFFFu);
                if (first && pb + 4u == sz) *lastPair = 1u;
            }
        }
        outw[j - 1] = out;
    }
    if (whole) {
        GcU4 v; v.x = outw[0] | (outw[1] << 16); v.y = outw[2] | (outw[3] << 16); v.z = outw[4] | (outw[5] << 16); v.w = outw[6] | (outw[7] << 16);
        __builtin_memcpy(dst + o, &v, 16);
    } else {
        for (int j = 0; j < 8; j++) {
            const uint64_t b = o + 2u * (uint64_t)j;
            if (b >= n) break;
            if (b + 2u > sz) { dst[b] = src[b]; break; }     // the odd byte at the end
            dst[b] = (uint8_t)outw[j]; dst[b + 1u] = (uint8_t)(outw[j] >> 8);
        }
    }
}

extern "C" int gc_bra_convert_device(int kind, const void* d_src, void* d_dst, size_t n, uint32_t pc, int encoding, size_t* processed)
{
    if (kind < GC_BRA_ARM64 || kind > GC_BRA_SPARC || (!d_src && n) || (!d_dst && n)) return GC_ERR_PARAM;
    if (kind == GC_BRA_ARMT && d_src == d_dst && n) return GC_ERR_PARAM;          // a Thumb pair can straddle two threads' chunks: out of place only
    if (processed) *processed = 0;
    if (!n) return GC_OK;
    uint32_t* dFlag = nullptr;
    uint32_t flag = 0;
    if (kind == GC_BRA_ARMT) {
        if (hipMalloc((void**)&dFlag, 4) != hipSuccess) return GC_ERR_NOMEM;
        if (hipMemcpy(dFlag, &flag, 4, hipMemcpyHostToDevice) != hipSuccess) { hipFree(dFlag); return GC_ERR_HIP; }
    }
    const uint64_t chunks = ((uint64_t)n + 15u) / 16u;
    GC_LAUNCH(gc_bra_kernel, (uint32_t)((chunks + 255u) / 256u), 256, (hipStream_t)0, (const uint8_t*)d_src, (uint8_t*)d_dst, (uint64_t)n, pc, (uint32_t)kind,
              (uint32_t)(encoding != 0), dFlag);
    bool ok = hipDeviceSynchronize() == hipSuccess;
    if (dFlag) { ok = ok && hipMemcpy(&flag, dFlag, 4, hipMemcpyDeviceToHost) == hipSuccess; hipFree(dFlag); }
    if (!ok) return GC_ERR_HIP;
    if (processed) {
        // what the reference's converter returns for one call on the whole buffer (the tail it leaves to the next call)
        if (kind == GC_BRA_ARMT) { const size_t sz = n & ~(size_t)1; *processed = sz <= 2u ? 0u : (flag ? sz : sz - 2u); }
        else *processed = n & ~(size_t)3;
    }
    return GC_OK;
}

// ---------------------------------------------------------------- X86 ----------------------------------------------------------------
#define BRA86_CHUNK 512u
__device__ __forceinline__ bool bra86_is(uint32_t b) { return (b & 0xFEu) == 0xE8u; }
__device__ __forceinline__ bool bra86_ms(uint32_t b) { return (((b) + 1u) & 0xFEu) == 0u; }        // 0x00 or 0xFF (BR86_NEED_CONV_FOR_MS_BYTE)
// no E8 / E9 among the 7 bytes in front of position c (c >= 7)
__device__ __forceinline__ bool bra86_restart(const uint8_t* src, uint64_t c)
{
    for (uint32_t k = 1; k <= 7u; k++) if (bra86_is(src[c - k])) return false;
    return true;
}

// The reference's control flow (Bra86.c:49-170) from position p.  atCand: p = candidate + 1 with mask 0 (label a3), else the top of the loop (label start).
// Stops (returns false) when it arrives at a candidate >= stopFrom that is a restart point; returns true when it reached the end: *outP, *outMask.
__device__ bool bra86_run(const uint8_t* __restrict__ src, uint8_t* dst, uint64_t n, uint32_t pc, bool enc, uint64_t p, uint32_t mask, bool atCand, uint64_t stopFrom,
                          uint64_t* outP, uint32_t* outMask)
{
    const uint64_t lim = n - 4u;
    enum { START, MAIN, A3, MLAB } mode = atCand ? A3 : START;
    for (;;) {
        if (mode == START) {
            if (p >= lim) break;
            const uint32_t w = gc_ld32(src + p) ^ 0xe8e8e8e8u;
            p += 4u;
            if ((w & 0xfeu) == 0u) { p -= 3u; mode = MLAB; continue; }
            mask >>= 1;
            if ((w & 0xfe00u) == 0u) { p -= 2u; mode = MLAB; continue; }
            mask >>= 1;
            if ((w & 0xfe0000u) == 0u) { p -= 1u; mode = MLAB; continue; }
            mask = 0;
            if ((w & 0xfe000000u) == 0u) { mode = A3; continue; }
            mode = MAIN; continue;
        }
        if (mode == MAIN) {
            if (p >= lim) break;
            bool found = false;
            for (;;) {
                const uint32_t w = gc_ld32(src + p) ^ 0xe8e8e8e8u;
                p += 4u;
                if ((w & 0xfeu) == 0u) { p -= 3u; found = true; break; }
                if ((w & 0xfe00u) == 0u) { p -= 2u; found = true; break; }
                if ((w & 0xfe0000u) == 0u) { p -= 1u; found = true; break; }
                if ((w & 0xfe000000u) == 0u) { found = true; break; }
                if (p >= lim) break;
            }
            if (!found) break;
            mode = A3; continue;
        }
        if (mode == MLAB) {
            if (mask == 0u) { mode = A3; continue; }
            if (p > lim) { p--; break; }
            if (mask > 4u || mask == 3u) { mask = (mask >> 1) | 4u; mode = START; continue; }
            mask >>= 1;
            if (bra86_ms(src[p + mask])) { mask |= 4u; mode = START; continue; }
            uint32_t v = gc_ld32(src + p) + (1u << 24);
            if (v & 0xfe000000u) { mask |= 4u; mode = START; continue; }
            const uint32_t c = pc + 4u + (uint32_t)p;
            v = enc ? v + c : v - c;
            const uint32_t sh = mask << 3;
            if (bra86_ms((v >> sh) & 0xFFu)) { v ^= (0x100u << sh) - 1u; v = enc ? v + c : v - c; }
            mask = 0;
            v &= (1u << 25) - 1u; v -= 1u << 24;
            __builtin_memcpy(dst + p, &v, 4);
            p += 4u; mode = MAIN; continue;
        }
        // A3: p = candidate + 1, mask = 0
        {
            const uint64_t cand = p - 1u;
            if (cand >= stopFrom && cand < lim && cand >= 7u && bra86_restart(src, cand)) return false;      // the next lane's start
            if (p > lim) { p--; break; }
            uint32_t v = gc_ld32(src + p) + (1u << 24);
            if (v & 0xfe000000u) { mask = 4u; mode = START; continue; }
            const uint32_t c = pc + 4u + (uint32_t)p;
            v = enc ? v + c : v - c;
            v &= (1u << 25) - 1u; v -= 1u << 24;
            __builtin_memcpy(dst + p, &v, 4);
            p += 4u; mode = MAIN; continue;
        }
    }
    *outP = p; *outMask = mask;
    return true;
}

extern "C" __global__ void __launch_bounds__(256)
gc_bra86_kernel(const uint8_t* __restrict__ src, uint8_t* dst, uint64_t n, uint32_t pc, uint32_t encoding, uint32_t stateIn, uint64_t* result)
{
    const uint64_t lane = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t lim = n - 4u, c0 = lane * BRA86_CHUNK, c1 = c0 + BRA86_CHUNK;
    if (c0 >= lim) return;
    uint64_t p = 0; uint32_t mask = stateIn; bool atCand = false;
    if (lane) {
        // my start: the first candidate in [max(c0, 7), min(c1, lim)) with no candidate among the 7 bytes in front of it
        const uint64_t hi = c1 < lim ? c1 : lim;
        uint64_t c = c0 < 7u ? 7u : c0;
        uint32_t quiet = 0;                                  // non-candidate bytes seen in a row, counting from c0 - 7
        for (uint64_t k = c - 7u; k < c; k++) quiet = bra86_is(src[k]) ? 0u : quiet + 1u;
        bool found = false;
        for (; c < hi; c++) {
            const bool is = bra86_is(src[c]);
            if (is && quiet >= 7u) { found = true; break; }
            quiet = is ? 0u : quiet + 1u;
        }
        if (!found) return;                                  // the lane in front of me runs through my bytes
        p = c + 1u; mask = 0; atCand = true;
    }
    uint64_t outP = 0; uint32_t outMask = 0;
    if (bra86_run(src, dst, n, pc, encoding != 0u, p, mask, atCand, c1, &outP, &outMask)) { result[0] = outP; result[1] = outMask; }
}

extern "C" int gc_bra_x86_convert_device(const void* d_src, void* d_dst, size_t n, uint32_t pc, int encoding, uint32_t* state, size_t* processed)
{
    if ((!d_src && n) || (!d_dst && n) || !state || (d_src == d_dst && n)) return GC_ERR_PARAM;
    if (processed) *processed = 0;
    if (!n) return GC_OK;
    if (hipMemcpy(d_dst, d_src, n, hipMemcpyDeviceToDevice) != hipSuccess) return GC_ERR_HIP;
    if (n < 5u) return GC_OK;                                // Bra86.c:52: nothing is processed, the state stays
    uint64_t* dRes = nullptr; uint64_t res[2] = { 0, 0 };
    if (hipMalloc((void**)&dRes, 16) != hipSuccess) return GC_ERR_NOMEM;
    const uint64_t lanes = ((uint64_t)n - 4u + BRA86_CHUNK - 1u) / BRA86_CHUNK;
    GC_LAUNCH(gc_bra86_kernel, (uint32_t)((lanes + 255u) / 256u), 256, (hipStream_t)0, (const uint8_t*)d_src, (uint8_t*)d_dst, (uint64_t)n, pc, (uint32_t)(encoding != 0), *state, dRes);
    const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(res, dRes, 16, hipMemcpyDeviceToHost) == hipSuccess;
    hipFree(dRes);
    if (!ok) return GC_ERR_HIP;
    *state = (uint32_t)res[1];
    if (processed) *processed = (size_t)res[0];
    return GC_OK;
}
