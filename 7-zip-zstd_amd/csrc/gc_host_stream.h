// gc_host_stream.h -- which stream the stand-alone device entry points (CRC-32, branch converters, Delta) work on.  Include after the HIP runtime (or its emulator stand-in).
#pragma once

// The stand-alone device entry points (gc_crc.hip, gc_bra.hip, gc_delta.hip) take no context, so they used the
// null stream and waited for the WHOLE device (hipDeviceSynchronize, hipMalloc / hipFree) on every call -- also when a context calls them for the pre-filter + CRC of a piece
// while another context's kernels run (two contexts per GPU in the host scheduler).  A context now names its own stream for the calls it makes on the calling thread
// (GcStreamScope in gc_api.hip); the kernels, copies and the stream-ordered scratch allocations go there and only that stream is waited for.  Public calls: the null stream, as before.
#ifdef __HIP_DEVICE_COMPILE__
extern hipStream_t gc_tls_stream;             // (host variable: the device pass only has to parse the host functions that name it)
#else
extern thread_local hipStream_t gc_tls_stream;
#endif
static inline hipError_t gc_copy_sync(void* d, const void* s, size_t n, hipMemcpyKind kind)
{
    const hipError_t e = hipMemcpyAsync(d, s, n, kind, gc_tls_stream);
    return e != hipSuccess ? e : hipStreamSynchronize(gc_tls_stream);
}
// Scratch memory of a call: a few buffers kept per calling thread and handed out again (grown with hipMalloc when a call needs more).  Every entry point that takes one waits for its
// stream before it returns, so a buffer is never in use on the device when the next call on the thread gets it.
// (Rounds 5-6 took these from the stream-ordered pool, hipMallocAsync / hipFreeAsync.  Run final5 of round 6: `7z a -m0=BCJGPU` wrote a wrong archive about once in ten runs -- the
// x86 converter's 16-byte result, read back from pool memory behind kernel and hipStreamSynchronize, was the PREVIOUS call's (the pool hands out the same address again; the
// kernel's store was not what the copy saw), and the filter advanced its program counter by 2 MiB instead of 155 KB (tools/gpu_diag_bcj.py).  Plain allocations, as everywhere
// else in the library, and the converters now put a mark into the result that the kernel must overwrite.)
struct GcScratchSlot { void* p; size_t cap; bool used; };
#ifdef __HIP_DEVICE_COMPILE__
extern GcScratchSlot gc_tls_scratch[4];
#else
extern thread_local GcScratchSlot gc_tls_scratch[4];
#endif
static inline hipError_t gc_scratch_alloc(void** p, size_t n)
{
    if (!n) n = 1;
    int pick = -1;
    for (int i = 0; i < 4; i++) if (!gc_tls_scratch[i].used && gc_tls_scratch[i].cap >= n && (pick < 0 || gc_tls_scratch[i].cap < gc_tls_scratch[pick].cap)) pick = i;
    if (pick < 0) {
        for (int i = 0; i < 4; i++) if (!gc_tls_scratch[i].used && (pick < 0 || gc_tls_scratch[i].cap < gc_tls_scratch[pick].cap)) pick = i;      // the smallest free one grows
        if (pick < 0) return (hipError_t)2;                   // (hipErrorOutOfMemory: four live scratch buffers on one thread -- no caller takes more than two)
        GcScratchSlot& s = gc_tls_scratch[pick];
        if (s.p) { (void)hipFree(s.p); s.p = nullptr; s.cap = 0; }
        const size_t cap = n < 4096 ? 4096 : n;
        const hipError_t e = hipMalloc(&s.p, cap);
        if (e != hipSuccess) { s.p = nullptr; return e; }
        s.cap = cap;
    }
    gc_tls_scratch[pick].used = true;
    *p = gc_tls_scratch[pick].p;
    return hipSuccess;
}
static inline void gc_scratch_free(void* p) { if (p) for (int i = 0; i < 4; i++) if (gc_tls_scratch[i].p == p) gc_tls_scratch[i].used = false; }
