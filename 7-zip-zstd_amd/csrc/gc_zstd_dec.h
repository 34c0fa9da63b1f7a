// gc_zstd_dec.h -- shared definitions of the zstd frame decoder on the device.
#pragma once
#include <stdint.h>
#include "gc_common.h"

// One entry per zstd frame of the compressed input (skippable frames are dropped by the scan).  The scan runs on the host: it only
// walks frame and block headers (3 bytes per 128 KiB block), all entropy decoding happens in the kernel.
struct GcZdFrame {
    uint64_t srcOff;        // first byte of the frame (its magic number)
    uint64_t srcSize;       // whole frame, checksum included
    uint64_t dstOff;        // where its content goes in the output
    uint64_t contentSize;   // Frame_Content_Size, valid if flags & GC_ZD_F_SIZE_KNOWN
    uint32_t flags;
    uint32_t hdrSize;       // bytes in front of the first block header
};
#define GC_ZD_F_CHECKSUM   1u
#define GC_ZD_F_SIZE_KNOWN 2u

#define GC_ZD_T         256u                          // threads per workgroup (one workgroup decodes one frame at a time)
#define GC_ZD_MAX_SEQ   98304u                        // the sequence count field holds at most 0x7F00 + 0xFFFF
#define GC_ZD_LIT_STRIDE (GC_ZSTD_BLOCK_MAX + 64u)    // regenerated literals of one block
#define GC_ZD_MAX_WG    256u                          // frames in flight (141 KB of LDS: one workgroup per CU)

// per frame result word: produced bytes | status << 56
#define GC_ZD_OK          0u
#define GC_ZD_CORRUPT     1u
#define GC_ZD_DST_SMALL   2u
#define GC_ZD_UNSUPPORTED 3u      // dictionary id
#define GC_ZD_CHECKSUM    4u
#define GC_ZD_SIZE        5u      // content size field does not match
