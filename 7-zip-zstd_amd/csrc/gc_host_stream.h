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
static inline hipError_t gc_scratch_alloc(void** p, size_t n) { return hipMallocAsync(p, n ? n : 1, gc_tls_stream); }
static inline void gc_scratch_free(void* p) { if (p) (void)hipFreeAsync(p, gc_tls_stream); }
