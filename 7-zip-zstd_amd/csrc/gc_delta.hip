// gc_delta.hip -- the Delta filter (C/Delta.c: out[i] = in[i] - in[i - delta], delta 1..256; method 3, CPP/7zip/Compress/DeltaFilter.cpp) on data
// that lies in HBM.  Encoding is elementwise.  Decoding, out[i] = in[i] + out[i - delta], is `delta` interleaved running sums
// modulo 256: every chain is cut into segments, one thread per (chain, segment) sums its segment, the segment totals are scanned, and the segments
// are then rebuilt from their starting values -- inside a 64 KiB chunk that is staged in LDS, with the chunk totals scanned by one small kernel
// in between (three launches: chunk totals, scan of the totals, rebuild).  state = the `delta` original bytes in front of the buffer, as in
// Delta_Encode / Delta_Decode (zero at the start of a stream).
#include "gpucodec.h"
#include "gc_device.h"
#ifdef HIPEMU
#include "hip_runtime_stub.h"
#else
#include <hip/hip_runtime.h>
#define GC_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif
#include <string.h>
#include "gc_host_stream.h"

#define DL_CHUNK 65536u
#define DL_T     256u

struct GcDeltaState { uint8_t b[256]; };

extern "C" __global__ void __launch_bounds__(256)
gc_delta_enc_kernel(const uint8_t* __restrict__ src, uint8_t* dst, uint64_t n, uint32_t delta, GcDeltaState st)
{
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 8u;
    if (i0 >= n) return;
    if (i0 + 8u <= n && i0 >= delta) {                    // eight bytes at once: bytewise subtraction inside a 64-bit word
        const uint64_t a = gc_ld64(src + i0), b = gc_ld64(src + i0 - delta);
        const uint64_t H = 0x8080808080808080ull;
        const uint64_t d = ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);
        __builtin_memcpy(dst + i0, &d, 8);
        return;
    }
    for (uint64_t i = i0; i < n && i < i0 + 8u; i++) dst[i] = (uint8_t)(src[i] - (i >= delta ? src[i - delta] : st.b[i]));
}

// chain r (positions = r mod delta) of chunk c, segment s: local chain indices [s * segLen, (s + 1) * segLen)
struct DlGeom { uint32_t nSeg, segLen; };
__device__ __forceinline__ DlGeom dl_geom(uint32_t delta)
{
    DlGeom g; g.nSeg = DL_T / delta; if (!g.nSeg) g.nSeg = 1u;
    const uint32_t chainMax = (DL_CHUNK + delta - 1u) / delta;
    g.segLen = (chainMax + g.nSeg - 1u) / g.nSeg;
    return g;
}

// totals[c][r] = sum of the bytes of chunk c on chain r (r = global position mod delta)
extern "C" __global__ void __launch_bounds__(DL_T)
gc_delta_totals_kernel(const uint8_t* __restrict__ src, uint64_t n, uint32_t delta, uint8_t* totals)
{
    __shared__ __attribute__((aligned(16))) uint8_t sBuf[DL_CHUNK];
    __shared__ uint32_t sTot[256];
    const uint32_t t = threadIdx.x, c = blockIdx.x;
    const uint64_t base = (uint64_t)c * DL_CHUNK;
    const uint32_t len = n - base < DL_CHUNK ? (uint32_t)(n - base) : DL_CHUNK;
    for (uint32_t i = t * 16u; i < len; i += DL_T * 16u) {
        if (i + 16u <= len) { GcU4 v; __builtin_memcpy(&v, src + base + i, 16); __builtin_memcpy(sBuf + i, &v, 16); }
        else for (uint32_t k = i; k < len; k++) sBuf[k] = src[base + k];
    }
    sTot[t] = 0;
    __syncthreads();
    const DlGeom g = dl_geom(delta);
    const uint32_t r = t % delta, s = t / delta;           // my chain (by LOCAL residue) and segment
    if (s < g.nSeg) {
        uint32_t sum = 0;
        for (uint32_t m = 0; m < g.segLen; m++) { const uint32_t j = r + delta * (s * g.segLen + m); if (j < len) sum += sBuf[j]; }
        atomicAdd(&sTot[r], sum);
    }
    __syncthreads();
    if (t < delta) totals[(uint64_t)c * 256u + (uint32_t)((base + t) % delta)] = (uint8_t)sTot[t];     // stored by GLOBAL residue
}

// carry[c][r] = value of out[] on chain r just in front of chunk c  (state + totals of the chunks in front)
extern "C" __global__ void __launch_bounds__(256)
gc_delta_scan_kernel(uint8_t* totals, uint32_t nChunks, uint32_t delta, GcDeltaState st)
{
    const uint32_t r = threadIdx.x;
    if (r >= delta) return;
    uint32_t acc = st.b[r];                                 // state[r] = original byte at position r - delta
    for (uint32_t c = 0; c < nChunks; c++) { const uint32_t v = totals[(uint64_t)c * 256u + r]; totals[(uint64_t)c * 256u + r] = (uint8_t)acc; acc += v; }
}

extern "C" __global__ void __launch_bounds__(DL_T)
gc_delta_dec_kernel(const uint8_t* __restrict__ src, uint8_t* dst, uint64_t n, uint32_t delta, const uint8_t* __restrict__ carry)
{
    __shared__ __attribute__((aligned(16))) uint8_t sBuf[DL_CHUNK];
    __shared__ uint8_t sSeg[256];                          // total of (chain, segment), then its starting value
    const uint32_t t = threadIdx.x, c = blockIdx.x;
    const uint64_t base = (uint64_t)c * DL_CHUNK;
    const uint32_t len = n - base < DL_CHUNK ? (uint32_t)(n - base) : DL_CHUNK;
    for (uint32_t i = t * 16u; i < len; i += DL_T * 16u) {
        if (i + 16u <= len) { GcU4 v; __builtin_memcpy(&v, src + base + i, 16); __builtin_memcpy(sBuf + i, &v, 16); }
        else for (uint32_t k = i; k < len; k++) sBuf[k] = src[base + k];
    }
    __syncthreads();
    const DlGeom g = dl_geom(delta);
    const uint32_t r = t % delta, s = t / delta;
    const bool mine = s < g.nSeg;
    if (mine) {
        uint32_t sum = 0;
        for (uint32_t m = 0; m < g.segLen; m++) { const uint32_t j = r + delta * (s * g.segLen + m); if (j < len) sum += sBuf[j]; }
        sSeg[s * delta + r] = (uint8_t)sum;
    }
    __syncthreads();
    if (t < delta) {                                       // starting value of every segment of chain t
        uint32_t acc = carry[(uint64_t)c * 256u + (uint32_t)((base + t) % delta)];
        for (uint32_t k = 0; k < g.nSeg; k++) { const uint32_t v = sSeg[k * delta + t]; sSeg[k * delta + t] = (uint8_t)acc; acc += v; }
    }
    __syncthreads();
    if (mine) {
        uint32_t acc = sSeg[s * delta + r];
        for (uint32_t m = 0; m < g.segLen; m++) { const uint32_t j = r + delta * (s * g.segLen + m); if (j < len) { acc += sBuf[j]; sBuf[j] = (uint8_t)acc; } }
    }
    __syncthreads();
    for (uint32_t i = t * 16u; i < len; i += DL_T * 16u) {
        if (i + 16u <= len) { GcU4 v; __builtin_memcpy(&v, sBuf + i, 16); __builtin_memcpy(dst + base + i, &v, 16); }
        else for (uint32_t k = i; k < len; k++) dst[base + k] = sBuf[k];
    }
}

extern "C" int gc_delta_convert_device(const void* d_src, void* d_dst, size_t n, unsigned delta, int encoding, unsigned char state[256])
{
    if (delta < 1u || delta > 256u || !state || (!d_src && n) || (!d_dst && n) || (encoding && d_src == d_dst && n)) return GC_ERR_PARAM;
    if (!n) return GC_OK;
    GcDeltaState st; memcpy(st.b, state, 256);
    uint8_t tail[256]; const size_t tl = n < delta ? n : delta;
    if (encoding) {
        GC_LAUNCH(gc_delta_enc_kernel, (uint32_t)(((n + 7u) / 8u + 255u) / 256u), 256, gc_tls_stream, (const uint8_t*)d_src, (uint8_t*)d_dst, (uint64_t)n, (uint32_t)delta, st);
        if (hipStreamSynchronize(gc_tls_stream) != hipSuccess || gc_copy_sync(tail, (const uint8_t*)d_src + n - tl, tl, hipMemcpyDeviceToHost) != hipSuccess) return GC_ERR_HIP;
    } else {
        const uint32_t nChunks = (uint32_t)((n + DL_CHUNK - 1u) / DL_CHUNK);
        uint8_t* dTot = nullptr;
        if (gc_scratch_alloc((void**)&dTot, (size_t)nChunks * 256u) != hipSuccess) return GC_ERR_NOMEM;
        GC_LAUNCH(gc_delta_totals_kernel, nChunks, DL_T, gc_tls_stream, (const uint8_t*)d_src, (uint64_t)n, (uint32_t)delta, dTot);
        GC_LAUNCH(gc_delta_scan_kernel, 1, 256, gc_tls_stream, dTot, nChunks, (uint32_t)delta, st);
        GC_LAUNCH(gc_delta_dec_kernel, nChunks, DL_T, gc_tls_stream, (const uint8_t*)d_src, (uint8_t*)d_dst, (uint64_t)n, (uint32_t)delta, (const uint8_t*)dTot);
        const bool ok = hipStreamSynchronize(gc_tls_stream) == hipSuccess && gc_copy_sync(tail, (const uint8_t*)d_dst + n - tl, tl, hipMemcpyDeviceToHost) == hipSuccess;
        gc_scratch_free(dTot);
        if (!ok) return GC_ERR_HIP;
    }
    // the state behind the buffer: the last `delta` original bytes (older ones shift down when the buffer is shorter than delta)
    uint8_t ns[256];
    for (size_t k = 0; k < delta; k++) ns[k] = k + tl < delta ? state[k + tl] : tail[k + tl - delta];
    memcpy(state, ns, delta);
    return GC_OK;
}
