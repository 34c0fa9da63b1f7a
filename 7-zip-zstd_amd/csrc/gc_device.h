// gc_device.h -- small device-side helpers (gfx950).  Under tests/emu (HIPEMU) the wave intrinsics are
// supplied by the SIMT emulator; nothing here selects between GPU back-ends.
#pragma once
#include <stdint.h>
#ifndef HIPEMU
#include <hip/hip_runtime.h>
#endif

// Unaligned little-endian loads.  gfx950 global/LDS accesses tolerate any byte alignment (unaligned access
// mode is on under ROCm), and the compiler turns the memcpy into a single global_load_dwordx2 / dword.
__device__ __forceinline__ uint64_t gc_ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint32_t gc_ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// 8 bytes at src[pos..], zero-filled past `limit` (end of the whole input buffer)
__device__ __forceinline__ uint64_t gc_ld64_guard(const uint8_t* src, uint64_t pos, uint64_t limit)
{
    if (pos + 8 <= limit) return gc_ld64(src + pos);
    uint64_t v = 0;
    for (int i = 0; i < 8; i++) if (pos + i < limit) v |= (uint64_t)src[pos + i] << (8 * i);
    return v;
}

struct __attribute__((aligned(16))) GcU4 { uint32_t x, y, z, w; };     // one global_load_dwordx4

__device__ __forceinline__ uint32_t gc_hibit32(uint32_t v) { return 31u - (uint32_t)__clz((int)v); }   // v != 0
__device__ __forceinline__ uint32_t gc_ctz64(uint64_t v) { return (uint32_t)__ffsll((long long)v) - 1u; } // v != 0

// LZMA distance slot of dist = distance - 1 (C/LzmaEnc.c GetPosSlot): 2 * floor(log2 dist) + next bit
__device__ __forceinline__ uint32_t gc_dist_slot(uint32_t dist)
{
    if (dist < 4u) return dist;
    const uint32_t hb = 31u - (uint32_t)__clz((int)dist);
    return 2u * hb + ((dist >> (hb - 1u)) & 1u);
}

// value known to be identical in every lane of the wave -> keep it in an SGPR
__device__ __forceinline__ uint32_t gc_uniform(uint32_t v)
{
#ifdef HIPEMU
    return v;
#else
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#endif
}

// make LDS writes of this wave visible to its other lanes (wave-synchronous code: no workgroup barrier needed).
// The fence is restricted to the LDS address space: a fence over all address spaces makes the wave wait for every global
// load and store it has in flight (s_waitcnt vmcnt(0)), which serialises software-pipelined loops.  Global memory written by
// one lane and read by another lane of the same wave needs gc_wave_sync_global().
__device__ __forceinline__ void gc_wave_sync()
{
#ifdef HIPEMU
    hipemu::wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
#endif
}
// Marks the end of one step of a wave-synchronous loop whose steps communicate only through LDS operations.  The hardware
// executes the LDS operations of a wave in program order, so nothing is needed there; the emulator runs its lanes one after
// another and needs the rendezvous to keep the steps of different lanes from overtaking each other.
__device__ __forceinline__ void gc_wave_step()
{
#ifdef HIPEMU
    hipemu::wave_barrier();
#else
    __builtin_amdgcn_wave_barrier();
#endif
}
__device__ __forceinline__ void gc_wave_sync_global()
{
#ifdef HIPEMU
    hipemu::wave_barrier();
#else
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
}

// Kernels that run beside each other: the producer makes what it wrote visible to the whole device and counts itself in; the consumer polls the
// counter and, once it has seen the value it waits for, drops what its caches may hold of the producer's lines (agent scope: across the XCDs' L2s).
__device__ __forceinline__ void gc_signal_device(uint32_t* counter)         // called by every lane of the wave; one lane counts
{
#ifdef HIPEMU
    hipemu::wave_barrier();
    if ((__lane_id() & 63u) == 0u) __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ uint32_t gc_poll_device(const uint32_t* counter)
{
#ifdef HIPEMU
    return __atomic_load_n(counter, __ATOMIC_ACQUIRE);
#else
    return __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void gc_acquire_device()
{
#ifndef HIPEMU
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
__device__ __forceinline__ void gc_nap()
{
#ifndef HIPEMU
    __builtin_amdgcn_s_sleep(32);
#endif
}

// shader cycle counter (s_memtime); only used by the optional phase profile
__device__ __forceinline__ unsigned long long gc_clock()
{
#ifdef HIPEMU
    return __builtin_ia32_rdtsc();
#else
    return (unsigned long long)clock64();
#endif
}

// value of `v` in lane `lane`, where `lane` is wave-uniform: a single v_readlane_b32 (no LDS round trip)
__device__ __forceinline__ uint32_t gc_readlane(uint32_t v, uint32_t lane)
{
#ifdef HIPEMU
    return __shfl(v, (int)lane);
#else
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)lane));
#endif
}

// Wave-wide shift by one lane towards lane 0: lane t receives the value of lane t + 1, lane 63 receives `fill`.
// One DPP move (wave_shl:1, a GFX9 control) -- no LDS crossbar, no index registers.
__device__ __forceinline__ uint32_t gc_wave_shl1(uint32_t v, uint32_t fill)
{
#ifdef HIPEMU
    const uint32_t o = __shfl_down(v, 1);
    return (__lane_id() & 63u) == 63u ? fill : o;
#else
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
#endif
}
// write the wave-uniform value `s` into lane `lane` (wave-uniform) of `v`: one v_writelane_b32
__device__ __forceinline__ uint32_t gc_writelane(uint32_t v, uint32_t s, uint32_t lane)
{
#ifdef HIPEMU
    return (__lane_id() & 63u) == (lane & 63u) ? s : v;
#else
    // (this compiler has no writelane builtin; the lane select travels in M0 because a VOP3 instruction reads one SGPR only)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane((int)s)), "s"(__builtin_amdgcn_readfirstlane((int)lane)) : "m0");
    return v;
#endif
}

// same with a compile-time lane (inline constant: no M0 traffic)
template <int LANE> __device__ __forceinline__ uint32_t gc_writelane_c(uint32_t v, uint32_t s)
{
#ifdef HIPEMU
    return (__lane_id() & 63u) == (uint32_t)LANE ? s : v;
#else
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane((int)s)), "n"(LANE));
    return v;
#endif
}

__device__ __forceinline__ uint64_t gc_lanemask_lt() { return (1ull << (__lane_id() & 63)) - 1ull; }

// inclusive wave scan (sum) over 64 lanes
__device__ __forceinline__ uint32_t gc_wave_incl_sum(uint32_t v)
{
    uint32_t lane = __lane_id();
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d); if (lane >= (uint32_t)d) v += o; }
    return v;
}
__device__ __forceinline__ uint32_t gc_wave_incl_max(uint32_t v)
{
    uint32_t lane = __lane_id();
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(v, d); if (lane >= (uint32_t)d) v = v > o ? v : o; }
    return v;
}
__device__ __forceinline__ uint32_t gc_wave_sum(uint32_t v)
{
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ uint32_t gc_wave_max(uint32_t v)
{
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(v, d); v = v > o ? v : o; }
    return v;
}

// Wave "publish / peek": every lane contributes one 32-bit value, afterwards any lane's value can be read with a
// wave-uniform index.  On gfx950 this is a plain VGPR + v_readlane_b32 (result in an SGPR, so everything computed from it
// stays on the scalar unit); under the emulator the values are copied out once per publish instead of once per peek.
#ifdef HIPEMU
struct GcPub { uint32_t v[64]; };
__device__ __forceinline__ void gc_publish(GcPub& pub, uint32_t mine)
{
    hipemu::WaveCtx& w = hipemu::wv();
    hipemu::wave_barrier();
    w.xchg[hipemu::lane()] = mine;
    hipemu::wave_barrier();
    for (int i = 0; i < 64; i++) pub.v[i] = (uint32_t)w.xchg[i];
    hipemu::wave_barrier();          // nobody reuses the exchange slots before every lane has copied them out
}
__device__ __forceinline__ uint32_t gc_peek(const GcPub& pub, uint32_t lane) { return pub.v[lane & 63u]; }
#else
struct GcPub { uint32_t mine; };
__device__ __forceinline__ void gc_publish(GcPub& pub, uint32_t mine) { pub.mine = mine; }
__device__ __forceinline__ uint32_t gc_peek(const GcPub& pub, uint32_t lane) { return gc_readlane(pub.mine, lane); }
#endif

