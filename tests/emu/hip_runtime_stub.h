// tests/emu/hip_runtime_stub.h -- TEST INFRASTRUCTURE ONLY: host-side HIP runtime stand-ins so that the
// product's host code (gc_api.hip) can be exercised together with the emulated kernels.  "Device memory"
// is plain malloc memory; streams and events are no-ops / wall clocks.
#pragma once
#include "hipemu.h"
#include <stdlib.h>
#include <string.h>
#include <chrono>
typedef int hipError_t;
typedef void* hipStream_t;
struct hipemu_event { std::chrono::steady_clock::time_point tp; };
typedef hipemu_event* hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; };
// HIPEMU_DEVICES=N makes the emulated machine report N devices (they share the one CPU): the multi-device host scheduler can then be exercised
static inline hipError_t hipGetDeviceCount(int* n) { const char* e = getenv("HIPEMU_DEVICES"); const int v = e ? atoi(e) : 1; *n = v >= 1 && v <= 64 ? v : 1; return 0; }
static inline hipError_t hipSetDevice(int) { return 0; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { strcpy(p->gcnArchName, "gfx950:emu"); p->multiProcessorCount = 256; return 0; }
// GC_EMU_POISON=<byte>: every device allocation starts filled with that byte (a negative value: with random bytes of that seed) (what a recycled allocation of a long-lived process looks like on the device;
// malloc's fresh pages are zero, which hides reads of words no kernel has written)
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); if (*p) { const char* e = getenv("GC_EMU_POISON"); if (e) { const int v = atoi(e); if (v >= 0) memset(*p, v, n ? n : 1); else { unsigned long long x = 0x9E3779B97F4A7C15ull * (unsigned long long)(-v); unsigned char* q = (unsigned char*)*p; for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; q[i] = (unsigned char)(x >> 32); } } } } return *p ? 0 : 2; }
static inline hipError_t hipFree(void* p) { free(p); return 0; }
typedef int hipMemcpyKind;
static inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
static inline hipError_t hipFreeAsync(void* p, hipStream_t) { free(p); return 0; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
static inline hipError_t hipHostFree(void* p) { free(p); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return 0; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emulated"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->tp = std::chrono::steady_clock::now(); return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->tp - a->tp).count(); return 0; }
#define GC_LAUNCH(kernel, grid, block, stream, ...) HIPEMU_LAUNCH(kernel, dim3(grid), dim3(block), __VA_ARGS__)
