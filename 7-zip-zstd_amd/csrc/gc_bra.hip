// gc_bra.hip -- branch converters ("BCJ" pre-filters of executables) on data that lies in HBM (the 7z folder pipeline runs them in
// front of the compressor, CPP/7zip/Compress/BranchMisc.cpp:21-26 -> C/Bra.c).  They turn the relative targets of CALL instructions into absolute
// ones so that repeated calls of one function become repeated byte strings.
//
// The five converters built here are the ones whose decision is local: ARM64 (BL and ADRP), ARM (BL), PPC (bl), SPARC (call) look at one
// aligned 4-byte word (C/Bra.c:75-257); ARMT (Thumb BL) looks at a pair of 16-bit units, and since the second unit of a pair (top bits 11111) can
// never be the first unit of one (11110), pairs never overlap and every pair is decided on its own as well (C/Bra.c:260-340).  One thread per
// 16 bytes, one 16-byte load and one 16-byte store: 2 bytes of HBM traffic per byte, which is the algorithmic minimum.  IA64 decides per 16-byte bundle (one thread each).
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
#include "gc_host_stream.h"

#define BRA86_CHUNK 512u           // bytes per lane of the two converters that run a state machine (X86, RISCV)

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
    if (kind == GC_BRA_IA64) {
        // one 16-byte bundle: the template (low 5 bits) says which of the three 41-bit slots hold branch instructions (C/Bra.c:343-420)
        if (o + 16u > (n & ~15ull)) { for (uint64_t b = o; b < n && b < o + 16u; b++) dst[b] = src[b]; return; }
        uint8_t q[20];
        { GcU4 v; __builtin_memcpy(&v, src + o, 16); __builtin_memcpy(q, &v, 16); q[16] = q[17] = q[18] = q[19] = 0; }
        uint32_t m = (0x334b0000u >> (q[0] & 0x1eu)) & 3u;
        if (m) {
            const uint32_t pcv = ((pc - 16u) >> 3) + 2u * (uint32_t)(o >> 4) + 2u;
            uint32_t at = 5u * m - 4u;                               // byte offset of the slot's first byte inside the bundle
            do {
                const uint32_t t = (uint32_t)q[at] | ((uint32_t)q[at + 1u] << 8);
                uint32_t z = ((uint32_t)q[at + 1u] | ((uint32_t)q[at + 2u] << 8) | ((uint32_t)q[at + 3u] << 16) | ((uint32_t)q[at + 4u] << 24)) >> m;
                if (((t >> m) & (0x70u << 1)) == 0u && ((z - (0x5000000u << 1)) & (0xf000000u << 1)) == 0u) {
                    uint32_t v = ((0x8fffffu << 1) | 1u) & z;
                    z ^= v;
                    const uint32_t low = (0x1fffffu << 1) | 1u;
                    v = enc ? v + (pcv & low) : v - (pcv | ~low);
                    v &= ~(0x600000u << 1);
                    v += 0x700000u << 1;
                    v &= (0x8fffffu << 1) | 1u;
                    z |= v;
                    z <<= m;
                    q[at + 1u] = (uint8_t)z; q[at + 2u] = (uint8_t)(z >> 8); q[at + 3u] = (uint8_t)(z >> 16); q[at + 4u] = (uint8_t)(z >> 24);
                }
                at += 5u; m++;
            } while (m &= 3u);
        }
        { GcU4 v; __builtin_memcpy(&v, q, 16); __builtin_memcpy(dst + o, &v, 16); }
        return;
    }
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


// ---------------------------------------------------------------- RISCV ----------------------------------------------------------------
// C/Bra.c:428-709.  JAL and AUIPC(+ the instruction behind it) at 2-byte granularity; after a candidate the scan goes on 2, 4, 6 or 8 bytes
// further, so whether a 16-bit unit is looked at depends on what came before -- the same situation as X86, solved the same way: a candidate
// unit with no candidate among the three units in front of it is reached by every history (restart point); lanes of 512 bytes start at
// their first restart point and run the reference's loop, restated, to the next lane's; decisions read the source, conversions go to a copy.
__device__ __forceinline__ bool rv_is(uint32_t hw) { return ((((hw) ^ 0x10u) + 1u) & 0x77u) == 0u; }
__device__ __forceinline__ uint32_t rv_ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ void rv_st32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
#define RV_REG_VAL (2u << 7)
#define RV_CMD_VAL 3u
#define RV_CHECK_1(v, b) (((((b) - RV_CMD_VAL) ^ ((v) << 8)) & (0xf8000u + RV_CMD_VAL)) == 0u)
#define RV_CHECK_2(v, r) ((((v) - ((RV_CMD_VAL << 12) | RV_REG_VAL | 8u)) << 18) < ((r) & 0x1du))

// from position p (a candidate unit when atCand) to the end or to the first restart candidate >= stopFrom; true = reached the end, *outP = the reference's return value
__device__ bool rv_run(const uint8_t* __restrict__ src, uint8_t* dst, uint64_t n, uint32_t pc, bool enc, uint64_t p, bool atCand, uint64_t stopFrom, uint64_t* outP)
{
    const uint64_t lim = (n & ~1ull) - 6u;
    for (;;) {
        uint32_t a = 0;
        if (atCand) { a = (rv_ld16(src + p) ^ 0x10u) + 1u; atCand = false; }
        else {
            bool found = false;
            for (;;) {
                if (p >= lim) { *outP = p; return true; }
                a = (rv_ld16(src + p) ^ 0x10u) + 1u;
                if ((a & 0x77u) == 0u) { found = true; break; }
                a = (rv_ld16(src + p + 2u) ^ 0x10u) + 1u;
                p += 4u;
                if ((a & 0x77u) == 0u) { p -= 2u; if (p >= lim) { *outP = p; return true; } found = true; break; }
            }
            (void)found;
            if (p >= stopFrom && p >= 6u && !rv_is(rv_ld16(src + p - 2u)) && !rv_is(rv_ld16(src + p - 4u)) && !rv_is(rv_ld16(src + p - 6u))) return false;   // the next lane's start
        }
        const uint32_t pcAt = pc + (uint32_t)p;
        uint32_t v = a;
        a = gc_ld32(src + p);
        if (enc) {
            if ((v & 8u) == 0u) {                                   // JAL
                if ((v - 0x100u) & 0xd80u) { p += 2u; continue; }
                v = ((a & (1u << 31)) >> 11) | ((a & (0x3ffu << 21)) >> 20) | ((a & (1u << 20)) >> 9) | (a & (0xffu << 12));
                v += pcAt;
                dst[p + 1u] = (uint8_t)(((v >> 13) & 0xf0u) | ((a >> 8) & 0xfu));
                dst[p + 2u] = (uint8_t)(v >> 9);
                dst[p + 3u] = (uint8_t)(v >> 1);
                p += 4u; continue;
            }
            if (v & 0xe80u) {                                       // AUIPC, rd neither x0 nor x2
                const uint32_t b = gc_ld32(src + p + 4u);
                if (RV_CHECK_1(v, b)) {
                    rv_st32(dst + p, (b << 12) | (0x17u + RV_REG_VAL));
                    a &= 0xfffff000u;
                    a += (uint32_t)((int32_t)b >> 20);
                    a += pcAt;
                    rv_st32(dst + p + 4u, bra_bswap(a));
                    p += 8u;
                } else p += 6u;
            } else {
                uint32_t r = a >> 27;
                if (RV_CHECK_2(v, r)) {
                    v = gc_ld32(src + p + 4u);
                    r = (r << 7) + 0x17u + (v & 0xfffff000u);
                    a = (a >> 12) | (v << 20);
                    rv_st32(dst + p, r); rv_st32(dst + p + 4u, a);
                    p += 8u;
                } else p += 4u;
            }
        } else {
            if ((v & 8u) == 0u) {                                   // JAL (v holds the transformed low 16 bits)
                uint32_t t = v - 0x100u + 0x7fu;
                if (t & 0xd80u) { p += 2u; continue; }
                const uint32_t a_old = (t + (0xefu - 0x7fu)) & 0xfffu;
                uint32_t x = ((uint32_t)src[p + 3u] << 1) | ((uint32_t)src[p + 2u] << 9) | ((t & 0xf000u) << 5);
                x -= pcAt;
                const uint32_t w = a_old | ((x << 11) & (1u << 31)) | ((x << 20) & (0x3ffu << 21)) | ((x << 9) & (1u << 20)) | (x & (0xffu << 12));
                rv_st32(dst + p, w);
                p += 4u; continue;
            }
            if ((v & 0xe80u) == 0u) {                               // x0 / x2
                const uint32_t r = a >> 27;
                if (RV_CHECK_2(v, r)) {
                    uint32_t b = bra_bswap(gc_ld32(src + p + 4u));
                    uint32_t x = a >> 12;
                    b -= pcAt;
                    uint32_t w = (r << 7) + 0x17u;
                    w += (b + 0x800u) & 0xfffff000u;
                    x |= b << 20;
                    rv_st32(dst + p, w); rv_st32(dst + p + 4u, x);
                    p += 8u;
                } else p += 4u;
            } else {
                const uint32_t b = gc_ld32(src + p + 4u);
                if (!RV_CHECK_1(v, b)) p += 6u;
                else {
                    const uint32_t x = (a & 0xfffff000u) | (b >> 20);
                    const uint32_t w = (b << 12) | (0x17u + RV_REG_VAL);
                    rv_st32(dst + p, w); rv_st32(dst + p + 4u, x);
                    p += 8u;
                }
            }
        }
    }
}

extern "C" __global__ void __launch_bounds__(256)
gc_bra_riscv_kernel(const uint8_t* __restrict__ src, uint8_t* dst, uint64_t n, uint32_t pc, uint32_t encoding, uint64_t* result)
{
    const uint64_t lane = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t lim = (n & ~1ull) - 6u, c0 = lane * BRA86_CHUNK, c1 = c0 + BRA86_CHUNK;
    if (c0 >= lim) return;
    uint64_t p = 0; bool atCand = false;
    if (lane) {
        const uint64_t hi = c1 < lim ? c1 : lim;
        uint64_t c = c0 < 6u ? 6u : c0;
        uint32_t quiet = 0;                                  // non-candidate units in a row in front of c
        for (uint64_t k = c - 6u; k < c; k += 2u) quiet = rv_is(rv_ld16(src + k)) ? 0u : quiet + 1u;
        bool found = false;
        for (; c < hi; c += 2u) {
            const bool is = rv_is(rv_ld16(src + c));
            if (is && quiet >= 3u) { found = true; break; }
            quiet = is ? 0u : quiet + 1u;
        }
        if (!found) return;
        p = c; atCand = true;
    }
    uint64_t outP = 0;
    if (rv_run(src, dst, n, pc, encoding != 0u, p, atCand, c1, &outP)) result[0] = outP;
}

static int bra_riscv(const void* d_src, void* d_dst, size_t n, uint32_t pc, int encoding, size_t* processed)
{
    if (processed) *processed = 0;
    if (!n) return GC_OK;
    if (gc_copy_sync(d_dst, d_src, n, hipMemcpyDeviceToDevice) != hipSuccess) return GC_ERR_HIP;
    if ((n & ~(size_t)1) <= 6u) return GC_OK;
    uint64_t* dRes = nullptr; uint64_t res = 0;
    if (gc_scratch_alloc((void**)&dRes, 8) != hipSuccess) return GC_ERR_NOMEM;
    if (hipMemsetAsync(dRes, 0xFF, 8, gc_tls_stream) != hipSuccess) { gc_scratch_free(dRes); return GC_ERR_HIP; }       // a mark the kernel must overwrite
    const uint64_t lanes = (((uint64_t)n & ~1ull) - 6u + BRA86_CHUNK - 1u) / BRA86_CHUNK;
    GC_LAUNCH(gc_bra_riscv_kernel, (uint32_t)((lanes + 255u) / 256u), 256, gc_tls_stream, (const uint8_t*)d_src, (uint8_t*)d_dst, (uint64_t)n, pc, (uint32_t)(encoding != 0), dRes);
    const bool ok = hipStreamSynchronize(gc_tls_stream) == hipSuccess && gc_copy_sync(&res, dRes, 8, hipMemcpyDeviceToHost) == hipSuccess;
    gc_scratch_free(dRes);
    if (!ok || res > (uint64_t)n) return GC_ERR_HIP;
    if (processed) *processed = (size_t)res;
    return GC_OK;
}

extern "C" int gc_bra_convert_device(int kind, const void* d_src, void* d_dst, size_t n, uint32_t pc, int encoding, size_t* processed)
{
    if (kind < GC_BRA_ARM64 || kind > GC_BRA_RISCV || (!d_src && n) || (!d_dst && n)) return GC_ERR_PARAM;
    if ((kind == GC_BRA_ARMT || kind == GC_BRA_RISCV) && d_src == d_dst && n) return GC_ERR_PARAM;          // a Thumb pair can straddle two threads' chunks: out of place only
    if (kind == GC_BRA_RISCV) return bra_riscv(d_src, d_dst, n, pc, encoding, processed);
    if (processed) *processed = 0;
    if (!n) return GC_OK;
    uint32_t* dFlag = nullptr;
    uint32_t flag = 0;
    if (kind == GC_BRA_ARMT) {
        if (gc_scratch_alloc((void**)&dFlag, 4) != hipSuccess) return GC_ERR_NOMEM;
        if (gc_copy_sync(dFlag, &flag, 4, hipMemcpyHostToDevice) != hipSuccess) { gc_scratch_free(dFlag); return GC_ERR_HIP; }
    }
    const uint64_t chunks = ((uint64_t)n + 15u) / 16u;
    GC_LAUNCH(gc_bra_kernel, (uint32_t)((chunks + 255u) / 256u), 256, gc_tls_stream, (const uint8_t*)d_src, (uint8_t*)d_dst, (uint64_t)n, pc, (uint32_t)kind,
              (uint32_t)(encoding != 0), dFlag);
    bool ok = hipStreamSynchronize(gc_tls_stream) == hipSuccess;
    if (dFlag) { ok = ok && gc_copy_sync(&flag, dFlag, 4, hipMemcpyDeviceToHost) == hipSuccess; gc_scratch_free(dFlag); }
    if (!ok) return GC_ERR_HIP;
    if (processed) {
        // what the reference's converter returns for one call on the whole buffer (the tail it leaves to the next call)
        if (kind == GC_BRA_ARMT) { const size_t sz = n & ~(size_t)1; *processed = sz <= 2u ? 0u : (flag ? sz : sz - 2u); }
        else *processed = kind == GC_BRA_IA64 ? n & ~(size_t)15 : n & ~(size_t)3;
    }
    return GC_OK;
}

// ---------------------------------------------------------------- X86 ----------------------------------------------------------------
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
    if (gc_copy_sync(d_dst, d_src, n, hipMemcpyDeviceToDevice) != hipSuccess) return GC_ERR_HIP;
    if (n < 5u) return GC_OK;                                // Bra86.c:52: nothing is processed, the state stays
    uint64_t* dRes = nullptr; uint64_t res[2] = { 0, 0 };
    if (gc_scratch_alloc((void**)&dRes, 16) != hipSuccess) return GC_ERR_NOMEM;
    if (hipMemsetAsync(dRes, 0xFF, 16, gc_tls_stream) != hipSuccess) { gc_scratch_free(dRes); return GC_ERR_HIP; }      // a mark the kernel must overwrite (the lane that reaches the end stores position and state)
    const uint64_t lanes = ((uint64_t)n - 4u + BRA86_CHUNK - 1u) / BRA86_CHUNK;
    GC_LAUNCH(gc_bra86_kernel, (uint32_t)((lanes + 255u) / 256u), 256, gc_tls_stream, (const uint8_t*)d_src, (uint8_t*)d_dst, (uint64_t)n, pc, (uint32_t)(encoding != 0), *state, dRes);
    const bool ok = hipStreamSynchronize(gc_tls_stream) == hipSuccess && gc_copy_sync(res, dRes, 16, hipMemcpyDeviceToHost) == hipSuccess;
    gc_scratch_free(dRes);
    if (!ok || res[0] > (uint64_t)n) return GC_ERR_HIP;             // (the mark is still there, or a position behind the buffer: never report bytes as converted on such a result)
    *state = (uint32_t)res[1];
    if (processed) *processed = (size_t)res[0];
    return GC_OK;
}
