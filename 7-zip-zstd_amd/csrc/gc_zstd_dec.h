// gc_zstd_dec.h -- shared definitions of the zstd frame decoder on the device.
#pragma once
#include <stdint.h>
#include "gc_common.h"

// One entry per zstd frame of the compressed input (skippable frames are dropped by the scan).  The scan runs on the host: it only
// walks frame and block headers (3 bytes per block); everything else happens in the kernels.
struct GcZdFrame {
    uint64_t srcOff;        // first byte of the frame (its magic number)
    uint64_t srcSize;       // whole frame, checksum included
    uint64_t dstOff;        // where its content goes in the output
    uint64_t contentSize;   // Frame_Content_Size, valid if flags & GC_ZD_F_SIZE_KNOWN
    uint32_t flags;
    uint32_t hdrSize;       // bytes in front of the first block header
    uint32_t nBlocks;
    uint32_t blockBase;     // index of its first block in the block table of the batch
    uint64_t litBase;       // start of its regenerated literals / sequence records in the workspaces
    uint64_t seqBase;
};
#define GC_ZD_F_CHECKSUM   1u
#define GC_ZD_F_SIZE_KNOWN 2u

// One entry per block, written by the index kernel, completed by the two entropy kernels (literals, sequences).
struct GcZdBlock {
    uint64_t srcOff;        // payload (behind the 3-byte block header), absolute in the compressed stream
    uint64_t litOff;        // frame-relative offsets into the literal (bytes) and sequence (records) workspaces
    uint64_t seqOff;
    uint32_t size;          // payload bytes (RLE: 1)
    uint32_t type;          // 0 raw, 1 RLE, 2 compressed | GC_ZD_B_LAST | GC_ZD_B_BAD
    uint32_t regen;         // raw / RLE: content bytes; compressed: regenerated literal bytes
    uint32_t litInfo;       // literals type (bits 0-1) | streams << 2 | literals header bytes << 8
    uint32_t comp;          // literals payload bytes (tree description + streams; raw: regen; RLE: 1)
    uint32_t nSeq;
    uint32_t seqPos;        // offset of the symbol-modes byte inside the block (behind the sequence count)
    uint32_t modes;
    uint32_t frame;
    // results of the entropy kernel (compressed blocks)
    uint32_t status;        // GC_ZD_*
    uint32_t outSize;       // content bytes of the block
    uint32_t lposEnd;       // literals consumed by the sequences (the rest goes behind the last match)
    uint32_t dposEnd;       // content bytes covered by the sequences
    uint32_t rep[3];        // repeat offsets behind the block, possibly symbolic (GC_ZD_SYM)
    uint32_t litStatus;     // GC_ZD_* of the literals kernel (blocks with Huffman-coded literals)
};
#define GC_ZD_B_LAST 4u
#define GC_ZD_B_BAD  8u
// A repeat offset that refers to the history in front of the block: GC_ZD_SYM | delta << 2 | k  =  (incoming repeat offset k) - delta.
// The entropy kernel decodes every block without knowing what came before; the execution kernel walks the blocks of a frame in order
// and puts the real values in.
#define GC_ZD_SYM 0x80000000u

// wide execution: where a block's content starts inside its frame, the repeat offsets it starts with, and whether the frame has already failed there
struct GcZdPlace { uint64_t dst; uint32_t rep[3]; uint32_t skip; };
#define GC_ZD_BATCH_BYTES (2ull << 30)                // content bytes decoded per batch of launches (a frame that is larger is a batch of its own)
#define GC_ZD_WIDE_MAX  0xFFF00000ull                 // content bytes of a batch the 32-bit positions of the wide path can address

#define GC_ZD_T         1024u                         // execution kernel
#define GC_ZD_MAX_WG    256u                          // frames in execution at a time (129 KB of LDS: one workgroup per CU)
#define GC_ZD_CHUNK     1024u                         // bytes of the sequence bitstream staged in LDS at a time

// per frame result word: produced bytes | status << 56
#define GC_ZD_OK          0u
#define GC_ZD_CORRUPT     1u
#define GC_ZD_DST_SMALL   2u
#define GC_ZD_UNSUPPORTED 3u      // dictionary id, offsets of 2 GiB and more
#define GC_ZD_CHECKSUM    4u
#define GC_ZD_SIZE        5u      // content size field does not match
