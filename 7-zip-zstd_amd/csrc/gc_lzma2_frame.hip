// gc_lzma2_frame.hip -- L4/L5 of the FLZMA2 path: LZMA2 chunk headers and stream assembly.
//
// Chunk layout (C/fast-lzma2/lzma2_enc.c:94-102,2040-2075; decoder C/Lzma2Dec.c:97-220):
//   0x00                                   end of stream
//   0x01 / 0x02, u16be (size-1), bytes     stored chunk (0x01 also resets the dictionary)
//   0x80 | reset<<5 | (usize-1)>>16, u16be (usize-1), u16be (csize-1), [props]   LZMA chunk; reset 0 = nothing (the chunk
//                                          continues the model of the previous one, only the range coder restarts), 2 = state +
//                                          props, 3 = state + props + dictionary.
// A 4 KiB rc chunk -- or a group of neighbouring rc chunks of one model segment that L2 merged (GC_LZMA_RC_MERGE_WORDS: the leader
// carries the group, the other members are marked absent) -- becomes one LZMA2 chunk.  The first chunk of a model segment resets the
// coder state (0xC0; 0xE0 for the first chunk of the stream) and carries the props byte, the others continue (0x80).  A segment whose
// LZMA chunks would not be smaller than the data is stored whole (one stored chunk per rc chunk); the segment behind it starts with a state reset like
// every segment, which is also what the decoder demands after a dictionary-resetting stored chunk.  Round 5: a model segment may span several 128 KiB blocks
// (segProps entry 0xFF = "continues the block in front"): only its first chunk resets, and it is stored or coded as a whole (gc_lzma2_segkind_kernel).
#include "gc_common.h"
#include "gc_device.h"
#include "gc_lzma2.h"

// Is a model segment coded as LZMA chunks, or stored?  One thread per segment ENTRY (props[sg]: the props byte of a segment's first block, 0xFF = the block
// continues the model of the block in front -- a segment over several blocks, gc_lzma2_model_kernel, round 5); the leader's thread decides for the whole segment:
// stored if one of its chunks overflowed its staging (or was never modelled), or if the LZMA chunks are not smaller than stored ones.  kind[sg] = 1 LZMA, 0 stored.
extern "C" __global__ void __launch_bounds__(256)
gc_lzma2_segkind_kernel(const GcLzmaChunkInfo* __restrict__ cinfo, const uint8_t* __restrict__ props, uint32_t nSegs, uint32_t segLog, uint8_t* __restrict__ kind)
{
    const uint32_t sg = blockIdx.x * 256u + threadIdx.x;
    if (sg >= nSegs || props[sg] == 0xFFu) return;
    const uint32_t rcPerSeg = 1u << (segLog - GC_LZMA_RC_LOG);
    uint32_t end = sg + 1u;
    while (end < nSegs && props[end] == 0xFFu) end++;
    uint32_t lz = 0, raw = 0; bool ok = true;
    for (uint32_t c = sg * rcPerSeg; c < end * rcPerSeg && ok; c++) {
        const GcLzmaChunkInfo ci = cinfo[c];
        if (ci.usize == 0u) continue;
        if (ci.csize == 0xFFFFFFFFu || ci.csize == 0u) { ok = false; break; }
        lz += (c == sg * rcPerSeg ? 6u : 5u) + ci.csize; raw += 3u + ci.usize;
    }
    const uint8_t k = (ok && lz < raw) ? 1u : 0u;
    for (uint32_t q = sg; q < end; q++) kind[q] = k;
}

// L4: one workgroup; exclusive scan of chunk sizes.  flags bit0: no end marker (more shards follow)
extern "C" __global__ void __launch_bounds__(1024)
gc_lzma2_plan_kernel(const GcLzmaChunkInfo* __restrict__ cinfo, uint32_t nRc, uint32_t segLog, uint64_t dstCap, uint32_t flags,
                     GcLzmaPlan* __restrict__ plan, uint64_t* __restrict__ result /* [0]=total bytes, [1]=error */,
                     const uint8_t* __restrict__ segProps, const uint8_t* __restrict__ segKind)
{
    __shared__ uint32_t sWave[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t rcPerSeg = 1u << (segLog - GC_LZMA_RC_LOG);
    uint64_t carry = 0;
    for (uint32_t tb = 0; tb < nRc; tb += 1024u) {
        const uint32_t c = tb + t;
        uint32_t size = 0, kind = 0;
        if (c < nRc) {
            const GcLzmaChunkInfo ci = cinfo[c];
            if (ci.usize) {
                kind = segKind[c / rcPerSeg] ? 1u : 2u;
                size = kind == 1u ? (((c & (rcPerSeg - 1u)) == 0u && segProps[c / rcPerSeg] != 0xFFu) ? 6u : 5u) + ci.csize : 3u + ci.usize;
            }
        }
        uint32_t incl = gc_wave_incl_sum(size);
        if (lane == 63u) sWave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 16u; w++) { uint32_t v = sWave[w]; if (w < wave) before += v; all += v; }
        __syncthreads();
        if (c < nRc) { GcLzmaPlan p; p.off = carry + before + incl - size; p.size = size; p.kind = kind; plan[c] = p; }
        carry += all;
    }
    if (t == 0) {
        const uint64_t total = carry + ((flags & 1u) ? 0u : 1u);
        result[0] = total; result[1] = total > dstCap ? 1u : 0u;
    }
}

// L5: one workgroup per rc chunk (+ one extra workgroup for the end marker)
extern "C" __global__ void __launch_bounds__(256)
gc_lzma2_emit_kernel(const uint8_t* __restrict__ src, uint32_t segLog, const uint8_t* __restrict__ rcOut,
                     const GcLzmaChunkInfo* __restrict__ cinfo, const GcLzmaPlan* __restrict__ plan, uint32_t nRc,
                     uint32_t flags, const uint64_t* __restrict__ result, uint8_t* __restrict__ dst, const uint8_t* __restrict__ segProps /* L2: props byte per model segment */)
{
    if (result[1]) return;
    const uint32_t t = threadIdx.x, c = blockIdx.x;
    if (c == nRc) { if (t == 0 && !(flags & 1u)) dst[result[0] - 1u] = 0x00; return; }
    const GcLzmaPlan p = plan[c];
    if (p.kind == 0u) return;
    const GcLzmaChunkInfo ci = cinfo[c];
    uint8_t* o = dst + p.off;
    const uint32_t u1 = ci.usize - 1u;
    const uint32_t rcPerSeg = 1u << (segLog - GC_LZMA_RC_LOG);
    if (p.kind == 1u) {
        const bool segFirst = (c & (rcPerSeg - 1u)) == 0u && segProps[c / rcPerSeg] != 0xFFu;       // (0xFF: the block continues the segment of the block in front -- an ordinary 0x80 chunk)
        const uint32_t hdr = segFirst ? 6u : 5u;
        if (t == 0) {
            const uint32_t c1 = ci.csize - 1u;
            o[0] = (uint8_t)(0x80u | ((segFirst ? (c == 0u ? 3u : 2u) : 0u) << 5) | (u1 >> 16));
            o[1] = (uint8_t)(u1 >> 8); o[2] = (uint8_t)u1; o[3] = (uint8_t)(c1 >> 8); o[4] = (uint8_t)c1;
            if (segFirst) o[5] = segProps[c / rcPerSeg];
        }
        const uint8_t* s = rcOut + (uint64_t)c * GC_LZMA_RC_STRIDE;
        for (uint32_t i = t; i < ci.csize; i += 256u) o[hdr + i] = s[i];
    } else {
        if (t == 0) { o[0] = c == 0u ? 0x01 : 0x02; o[1] = (uint8_t)(u1 >> 8); o[2] = (uint8_t)u1; }
        const uint8_t* s = src + ((uint64_t)c << GC_LZMA_RC_LOG);
        for (uint32_t i = t; i < ci.usize; i += 256u) o[3u + i] = s[i];
    }
}
