// gc_zstd_frame.hip -- K4/K5: frame planning and assembly.
//
// Every run of `frameBlocks` blocks becomes one single-segment zstd frame (magic, frame header descriptor, frame content
// size, then per block a 3-byte header + payload) -- ZSTD_writeFrameHeader (C/zstd/zstd_compress.c:4695-4745), block header
// `last | type<<1 | size<<3` (:4655-4658), raw-block fallback when the sections do not beat the input
// (ZSTD_noCompressBlock, zstd_compress_internal.h:650; decision :3033-3035).  frameBlocks = 1 (block-local match finder):
// every block is its own frame.  The reference decoder accepts any number of concatenated frames
// (CPP/7zip/Compress/ZstdDecoder.cpp:145-158), which is what makes frames independent units for the GPU and, one level
// up, for range-splitting across GPUs.
#include "gc_common.h"
#include "gc_device.h"

#define FRAME_T 256u

struct GcFramePlan { uint64_t off; uint32_t size; uint32_t compressed; };

__device__ __forceinline__ uint32_t frame_hdr_size(uint32_t frameLen) { return 5u + (frameLen < 256u ? 1u : (frameLen < 65536u + 256u ? 2u : 4u)); }
// content length of the frame that block b belongs to
__device__ __forceinline__ uint32_t frame_len_of(uint32_t b, uint32_t frameBlocks, uint64_t srcSize)
{
    const uint64_t fb = (uint64_t)frameBlocks * GC_ZSTD_BLOCK_MAX;
    const uint64_t start = (uint64_t)(b / frameBlocks) * fb;
    return (uint32_t)((srcSize - start) < fb ? (srcSize - start) : fb);
}

// K4: one workgroup; exclusive scan of frame sizes
extern "C" __global__ void __launch_bounds__(1024)
gc_zstd_plan_kernel(const GcSectionInfo* __restrict__ info, uint32_t nBlocks, uint64_t srcSize, uint64_t dstCap, uint32_t frameBlocks,
                    GcFramePlan* __restrict__ plan, uint64_t* __restrict__ result /* [0]=total bytes, [1]=error */,
                    uint32_t seekTable /* 1: a seek table (skippable frame) follows the last frame */)
{
    __shared__ uint32_t sWave[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    uint64_t carry = 0;
    for (uint32_t tb = 0; tb < nBlocks; tb += 1024u) {
        const uint32_t b = tb + t;
        uint32_t size = 0, comp = 0;
        if (b < nBlocks) {
            const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
            const uint32_t blockLen = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
            const GcSectionInfo si = info[b];
            const uint64_t payload = (uint64_t)si.litSecSize + si.seqSecSize;
            comp = (si.seqSecSize != 0xFFFFFFFFu && payload < blockLen) ? 1u : 0u;
            const uint32_t hdr = (b % frameBlocks) == 0u ? frame_hdr_size(frame_len_of(b, frameBlocks, srcSize)) : 0u;
            size = hdr + 3u + (comp ? (uint32_t)payload : blockLen);
        }
        uint32_t incl = gc_wave_incl_sum(size);
        if (lane == 63u) sWave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
        for (uint32_t w = 0; w < 16u; w++) { uint32_t c = sWave[w]; if (w < wave) before += c; all += c; }
        __syncthreads();
        if (b < nBlocks) { GcFramePlan p; p.off = carry + before + incl - size; p.size = size; p.compressed = comp; plan[b] = p; }
        carry += all;
    }
    if (t == 0) {
        const uint32_t nFrames = (nBlocks + frameBlocks - 1u) / frameBlocks;
        const uint64_t total = carry + (seekTable ? 8ull + 8ull * nFrames + 9ull : 0ull);
        result[0] = total; result[1] = total > dstCap ? 1u : 0u;
    }
}

// K5: one workgroup per block writes its frame; with a seek table one more workgroup (blockIdx == nBlocks) writes it:
//   0x184D2A5E | size of the rest | per frame { compressed size, decompressed size } | number of frames | descriptor 0 | 0x8F92EAB1
// (zstd seekable format, contrib/seekable_format/zstd_seekable_compression_format.md; little endian, no per-frame checksums)
extern "C" __global__ void __launch_bounds__(FRAME_T)
gc_zstd_emit_kernel(const uint8_t* __restrict__ src, uint64_t srcSize, const uint8_t* __restrict__ litSec,
                    const uint8_t* __restrict__ seqSec, const GcSectionInfo* __restrict__ info,
                    const GcFramePlan* __restrict__ plan, const uint64_t* __restrict__ result, uint32_t nBlocks, uint32_t frameBlocks,
                    uint8_t* __restrict__ dst)
{
    if (result[1]) return;                       // output buffer too small: write nothing
    const uint32_t t = threadIdx.x, b = blockIdx.x;
    if (b == nBlocks) {                          // the seek table (only launched when one is wanted)
        const uint32_t nFrames = (nBlocks + frameBlocks - 1u) / frameBlocks;
        const uint64_t tableBytes = 8ull + 8ull * nFrames + 9ull, framesEnd = result[0] - tableBytes;
        uint8_t* o = dst + framesEnd;
        auto put32 = [](uint8_t* q, uint32_t v) { q[0] = (uint8_t)v; q[1] = (uint8_t)(v >> 8); q[2] = (uint8_t)(v >> 16); q[3] = (uint8_t)(v >> 24); };
        if (t == 0) { put32(o, 0x184D2A5Eu); put32(o + 4, (uint32_t)(tableBytes - 8ull)); }
        for (uint32_t f = t; f < nFrames; f += FRAME_T) {
            const uint32_t b0 = f * frameBlocks, b1 = b0 + frameBlocks;
            const uint64_t end = b1 < nBlocks ? plan[b1].off : framesEnd;
            put32(o + 8u + 8ull * f, (uint32_t)(end - plan[b0].off));
            put32(o + 12u + 8ull * f, frame_len_of(b0, frameBlocks, srcSize));
        }
        if (t == 0) { uint8_t* ft = o + 8u + 8ull * nFrames; put32(ft, nFrames); ft[4] = 0; put32(ft + 5, 0x8F92EAB1u); }
        return;
    }
    const uint64_t base = (uint64_t)b * GC_ZSTD_BLOCK_MAX;
    const uint32_t blockLen = (uint32_t)((srcSize - base) < GC_ZSTD_BLOCK_MAX ? (srcSize - base) : GC_ZSTD_BLOCK_MAX);
    const GcFramePlan p = plan[b];
    const GcSectionInfo si = info[b];
    uint8_t* o = dst + p.off;
    const bool first = (b % frameBlocks) == 0u, last = (b % frameBlocks) == frameBlocks - 1u || b == nBlocks - 1u;
    const uint32_t frameLen = frame_len_of(b, frameBlocks, srcSize);
    const uint32_t hs = first ? frame_hdr_size(frameLen) : 0u;
    if (t == 0) {
        if (first) {
            o[0] = 0x28; o[1] = 0xB5; o[2] = 0x2F; o[3] = 0xFD;                     // ZSTD_MAGICNUMBER 0xFD2FB528
            const uint32_t fcsCode = hs == 6u ? 0u : (hs == 7u ? 1u : 2u);
            o[4] = (uint8_t)((fcsCode << 6) | (1u << 5));                            // single segment, no checksum, no dictID
            if (hs == 6u) o[5] = (uint8_t)frameLen;
            else if (hs == 7u) { uint32_t v = frameLen - 256u; o[5] = (uint8_t)v; o[6] = (uint8_t)(v >> 8); }
            else { o[5] = (uint8_t)frameLen; o[6] = (uint8_t)(frameLen >> 8); o[7] = (uint8_t)(frameLen >> 16); o[8] = (uint8_t)(frameLen >> 24); }
        }
        const uint32_t bsz = p.compressed ? si.litSecSize + si.seqSecSize : blockLen;
        const uint32_t bh = (last ? 1u : 0u) | ((p.compressed ? 2u : 0u) << 1) | (bsz << 3);  // last block, type, size
        o[hs] = (uint8_t)bh; o[hs + 1u] = (uint8_t)(bh >> 8); o[hs + 2u] = (uint8_t)(bh >> 16);
    }
    uint8_t* pay = o + hs + 3u;
    if (p.compressed) {
        const uint8_t* ls = litSec + (uint64_t)b * GC_LITSEC_STRIDE;
        const uint8_t* ss = seqSec + (uint64_t)b * GC_SEQSEC_STRIDE;
        for (uint32_t i = t; i < si.litSecSize; i += FRAME_T) pay[i] = ls[i];
        pay += si.litSecSize;
        for (uint32_t i = t; i < si.seqSecSize; i += FRAME_T) pay[i] = ss[i];
    } else {
        const uint8_t* s = src + base;
        for (uint32_t i = t; i < blockLen; i += FRAME_T) pay[i] = s[i];
    }
}
