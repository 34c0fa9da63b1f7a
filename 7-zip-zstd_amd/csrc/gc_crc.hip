// gc_crc.hip -- CRC-32 of a buffer that already lies in HBM (the 7z folder pipeline computes the CRC of every unpacked stream,
// CPP/7zip/Archive/7z/7zUpdate.cpp via C/7zCrc.c CrcUpdate; with the data on the device for compression anyway, the checksum can come from
// there instead of from a serial pass over the same bytes on one host core).
//
// CRC-32/ISO-HDLC as C/7zCrc.c computes it: reflected polynomial 0xEDB88320, initial value and final XOR 0xFFFFFFFF.  The remainder is linear
// over GF(2): with R(M) = the register after M starting from 0,  R(A || B) = shift(R(A), |B|) ^ R(B), where shift multiplies by x^(8 |B|)
// modulo the polynomial.  The kernel computes R for every 4 KiB slice (one lane per slice, byte table in LDS) and folds the 256 slices of a
// 1 MiB chunk in LDS with the eight fixed shift operators x^(8 * 4096 * 2^k) (32 x 32 bit matrices built on the host); the host folds the chunk
// values (one matrix-vector product per MiB), the tail below 1 MiB and the initial value's own shift.  No MFMA: a GF(2) matrix-vector product
// per MiB is not a contraction worth the name.
#include "gpucodec.h"
#include "gc_device.h"
#ifdef HIPEMU
#include "hip_runtime_stub.h"
#else
#include <hip/hip_runtime.h>
#define GC_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif
#include <stdlib.h>
#include <string.h>
#include "gc_host_stream.h"

#define CRC_SLICE 4096u
#define CRC_T     256u                 // slices per chunk
#define CRC_CHUNK (CRC_SLICE * CRC_T)  // 1 MiB

struct GcCrcOps { uint32_t table[256]; uint32_t shift[8][32]; };      // shift[k][i] = image of bit i under "append 4096 * 2^k zero bytes"

__device__ __forceinline__ uint32_t crc_apply(const uint32_t* m, uint32_t v)
{
    uint32_t r = 0;
#pragma unroll
    for (uint32_t i = 0; i < 32u; i++) r ^= ((v >> i) & 1u) ? m[i] : 0u;
    return r;
}

extern "C" __global__ void __launch_bounds__(CRC_T)
gc_crc32_chunk_kernel(const uint8_t* __restrict__ src, uint32_t nChunks, const GcCrcOps* __restrict__ ops, uint32_t* __restrict__ out)
{
    __shared__ uint32_t sTab[256];
    __shared__ uint32_t sShift[8][32];
    __shared__ uint32_t sR[CRC_T];
    const uint32_t t = threadIdx.x, c = blockIdx.x;
    if (c >= nChunks) return;
    sTab[t] = ops->table[t];
    sShift[t >> 5][t & 31u] = ops->shift[t >> 5][t & 31u];
    __syncthreads();
    const GcU4* p = (const GcU4*)(src + (uint64_t)c * CRC_CHUNK + (uint64_t)t * CRC_SLICE);
    uint32_t r = 0;
    GcU4 nxt = p[0];
    for (uint32_t i = 0; i < CRC_SLICE / 16u; i++) {
        const GcU4 v = nxt;
        if (i + 1u < CRC_SLICE / 16u) nxt = p[i + 1u];
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            r ^= w[k];                                            // four bytes at a time through the byte table
            r = sTab[r & 0xFFu] ^ (r >> 8); r = sTab[r & 0xFFu] ^ (r >> 8); r = sTab[r & 0xFFu] ^ (r >> 8); r = sTab[r & 0xFFu] ^ (r >> 8);
        }
    }
    sR[t] = r;
    __syncthreads();
    for (uint32_t k = 0; k < 8u; k++) {                           // fold pairs: left value shifted over the right one's 4096 * 2^k bytes
        const uint32_t stride = 1u << k;
        uint32_t v = 0;
        const bool on = (t & (2u * stride - 1u)) == 0u;
        if (on) v = crc_apply(sShift[k], sR[t]) ^ sR[t + stride];
        __syncthreads();
        if (on) sR[t] = v;
        __syncthreads();
    }
    if (t == 0) out[c] = sR[0];
}

// ---------------------------------------------------------------------------------------------------- host side
#ifndef __HIP_DEVICE_COMPILE__
thread_local hipStream_t gc_tls_stream = nullptr;          // (gc_device.h)
thread_local GcScratchSlot gc_tls_scratch[4] = { { nullptr, 0, false }, { nullptr, 0, false }, { nullptr, 0, false }, { nullptr, 0, false } };   // (gc_host_stream.h)
#endif
static uint32_t crc_byte_table_entry(uint32_t i) { uint32_t r = i; for (int k = 0; k < 8; k++) r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1u))); return r; }
static uint32_t crc_mat_apply(const uint32_t m[32], uint32_t v) { uint32_t r = 0; for (int i = 0; i < 32; i++) if ((v >> i) & 1u) r ^= m[i]; return r; }
static void crc_mat_square(uint32_t out[32], const uint32_t m[32]) { for (int i = 0; i < 32; i++) out[i] = crc_mat_apply(m, m[i]); }
// operator "append n zero bytes" (n >= 1) by repeated squaring from the one-zero-byte operator
static void crc_shift_op(uint32_t out[32], uint64_t n)
{
    uint32_t sq[32], tmp[32], acc[32]; bool have = false;
    for (int i = 0; i < 32; i++) { const uint32_t v = 1u << i; uint32_t r = v; for (int k = 0; k < 8; k++) r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1u))); sq[i] = r; }
    for (; n; n >>= 1) {
        if (n & 1u) { if (!have) { memcpy(acc, sq, sizeof(acc)); have = true; } else { for (int i = 0; i < 32; i++) tmp[i] = crc_mat_apply(sq, acc[i]); memcpy(acc, tmp, sizeof(acc)); } }
        crc_mat_square(tmp, sq); memcpy(sq, tmp, sizeof(sq));
    }
    memcpy(out, acc, sizeof(acc));
}

// CRC-32 of n bytes in device memory; on the calling thread's stream (gc_device.h: the null stream unless a context set its own).  0 on success.
extern "C" int gc_crc32_device(const void* d_src, size_t n, uint32_t* crc)
{
    if ((!d_src && n) || !crc) return GC_ERR_PARAM;
    static GcCrcOps hostOps; static bool built = false;
    if (!built) {
        for (uint32_t i = 0; i < 256u; i++) hostOps.table[i] = crc_byte_table_entry(i);
        for (uint32_t k = 0; k < 8u; k++) crc_shift_op(hostOps.shift[k], (uint64_t)CRC_SLICE << k);
        built = true;
    }
    const uint32_t nChunks = (uint32_t)(n / CRC_CHUNK);
    const size_t tail = n - (size_t)nChunks * CRC_CHUNK;
    uint32_t reg = 0;                                             // R(bytes so far), register started from 0
    uint8_t* tailBytes = (uint8_t*)malloc(tail ? tail : 1);
    if (!tailBytes) return GC_ERR_NOMEM;
    if (nChunks) {
        GcCrcOps* dOps = nullptr; uint32_t* dOut = nullptr;
        uint32_t* hOut = (uint32_t*)malloc((size_t)nChunks * 4u);
        if (!hOut || gc_scratch_alloc((void**)&dOps, sizeof(GcCrcOps)) != hipSuccess || gc_scratch_alloc((void**)&dOut, (size_t)nChunks * 4u) != hipSuccess) { free(hOut); free(tailBytes); gc_scratch_free(dOps); gc_scratch_free(dOut); return GC_ERR_NOMEM; }
        bool ok = gc_copy_sync(dOps, &hostOps, sizeof(GcCrcOps), hipMemcpyHostToDevice) == hipSuccess;
        if (ok) { GC_LAUNCH(gc_crc32_chunk_kernel, nChunks, CRC_T, gc_tls_stream, (const uint8_t*)d_src, nChunks, (const GcCrcOps*)dOps, dOut); }
        ok = ok && gc_copy_sync(hOut, dOut, (size_t)nChunks * 4u, hipMemcpyDeviceToHost) == hipSuccess;
        gc_scratch_free(dOps); gc_scratch_free(dOut);
        if (!ok) { free(hOut); free(tailBytes); return GC_ERR_HIP; }
        uint32_t chunkOp[32]; crc_shift_op(chunkOp, CRC_CHUNK);
        for (uint32_t c = 0; c < nChunks; c++) reg = crc_mat_apply(chunkOp, reg) ^ hOut[c];
        free(hOut);
    }
    if (tail) {
        if (gc_copy_sync(tailBytes, (const uint8_t*)d_src + (size_t)nChunks * CRC_CHUNK, tail, hipMemcpyDeviceToHost) != hipSuccess) { free(tailBytes); return GC_ERR_HIP; }
        for (size_t i = 0; i < tail; i++) reg = hostOps.table[(reg ^ tailBytes[i]) & 0xFFu] ^ (reg >> 8);
    }
    free(tailBytes);
    // the initial value 0xFFFFFFFF rides along as R-linear term: shift(0xFFFFFFFF, n); then the final XOR
    uint32_t init = 0xFFFFFFFFu;
    if (n) { uint32_t op[32]; crc_shift_op(op, n); init = crc_mat_apply(op, init); }
    *crc = reg ^ init ^ 0xFFFFFFFFu;
    return GC_OK;
}
