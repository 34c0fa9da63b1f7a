/* include/gpucodec.h -- C ABI of the MI355X block-parallel compression engine (libgpucodec.so).
 *
 * This is the boundary the reference's codec wrappers bind.  Each entry point names the reference
 * interface it replaces (paths relative to /root/reference):
 *
 *   gc_zstd_compress_host   <->  the ZSTD_compressStream2() loop inside NCompress::NZSTD::CEncoder::Code
 *                                (CPP/7zip/Compress/ZstdEncoder.cpp:398-461; library entry
 *                                C/zstd/zstd_compress.c:6447).  Same contract: bytes in, a zstd stream out
 *                                that ZSTD_decompressStream / NCompress::NZSTD::CDecoder
 *                                (CPP/7zip/Compress/ZstdDecoder.cpp:66-175) regenerates bit-exactly.
 *   gc_zstd_compress_device <->  same, for callers that already hold the input in HBM (bench, multi-GPU
 *                                sharding: one context per GPU, host range-splits the input).
 *   gc_zstd_compress_bound  <->  ZSTD_compressBound (C/zstd/zstd_compress.c:69).
 *   gc_ctx_create/destroy   <->  ZSTD_createCCtx / ZSTD_freeCCtx (ZstdEncoder.cpp:262, :44).
 *
 * The 7-Zip plugin surface (GetNumberOfMethods, GetMethodProperty, CreateEncoder, CreateDecoder,
 * CreateObject, GetModuleProp + the ICompressCoder vtable; CPP/7zip/Compress/CodecExports.cpp:153-378)
 * lives in lib7zgpucodec.so, which is a thin C++ layer over this ABI.
 *
 * Plain C types only; no torch / HIP types cross this boundary.  All functions return GC_OK (0) or a
 * negative GC_ERR_* code; there is NO CPU fallback: without a usable gfx950 device gc_ctx_create fails.
 */
#ifndef GPUCODEC_H
#define GPUCODEC_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GC_OK               0
#define GC_ERR_NO_DEVICE   -1   /* no HIP device / wrong architecture */
#define GC_ERR_HIP         -2   /* a HIP runtime call failed; see gc_last_error_message */
#define GC_ERR_NOMEM       -3
#define GC_ERR_DST_SMALL   -4   /* dstCapacity < compressed size (cf. ZSTD_error_dstSize_tooSmall) */
#define GC_ERR_PARAM       -5
#define GC_ERR_CORRUPT     -6   /* decoder: the compressed data is damaged (cf. ZSTD_error_corruption_detected, checksum_wrong) */
#define GC_ERR_UNSUPPORTED -7   /* decoder: a valid stream that needs what this decoder does not hold (brotli: the static dictionary has not been handed over) */

typedef struct gc_ctx gc_ctx;

int         gc_device_count(void);
int         gc_ctx_create(gc_ctx** out, int device);
void        gc_ctx_destroy(gc_ctx* ctx);
const char* gc_last_error_message(const gc_ctx* ctx);

/* worst-case compressed size for n input bytes (every block stored raw + frame overhead) */
size_t      gc_zstd_compress_bound(size_t n);

/* Compress n bytes already resident in device memory into device memory.  Asynchronous on the context's
 * stream; call gc_zstd_finish to synchronise and fetch the size.  `level` follows the reference's -mx
 * scale (1..22) and selects the configuration of the GPU path: levels 1-5 the windowed finder over 8 MiB frames with a one-step lazy parse, 6+ lazy2 and link following, 7+ the far pass
 * (16- / 12-byte keys), 10+ the short pass and the price-based parse. */
int         gc_zstd_compress_device(gc_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t dstCapacity, int level);
int         gc_zstd_finish(gc_ctx* ctx, size_t* compressedSize);

/* Host-buffer convenience: H2D copy, compress, D2H copy (what CEncoder::Code needs). */
int         gc_zstd_compress_host(gc_ctx* ctx, const void* src, size_t n, void* dst, size_t dstCapacity, int level,
                                  size_t* compressedSize);

/* HIP-event timing of the kernels of the last gc_zstd_compress_device call (after gc_zstd_finish):
 * ms[0..4] = lz, huf, seq, plan, emit (huf and seq overlap on two streams); ms[5] = first kernel start -> last kernel end. */
int         gc_zstd_last_timing(gc_ctx* ctx, float ms[6]);

/* Levels >= 3 run the windowed match finder (six kernels: count, scan, scatter, link, verify, parse); ms[0..5] = their
 * HIP-event durations in the last call (any codec).  GC_ERR_PARAM if the last call used the block-local finder. */
int         gc_mf_last_timing(gc_ctx* ctx, float ms[6]);

/* The price-based parse (FLZMA2 level >= 3, zstd level >= 5, brotli quality >= 7: the counterpart of LZMA_optimalParse
 * C/fast-lzma2/lzma2_enc.c:949 and ZSTD_compressBlock_opt_generic C/zstd/zstd_opt.c:1077) runs four kernels inside the "parse"
 * entry above; ms[0..3] = greedy parse + statistics, short candidates, shortest path, second parse pass.  GC_ERR_PARAM if the
 * last call did not run it. */
int         gc_mf_price_timing(gc_ctx* ctx, float ms[4]);

/* The "verify" entry of gc_mf_last_timing covers up to four groups of kernels: ms[0..3] = W5 verify, the far pass (a second
 * W1..W5 with 16- / 12-byte keys: the longer matches that RMF_buildTable radix_engine.h:920 / ZSTD_insertBtAndGetAllMatches
 * zstd_opt.c:590 return), W5b link following, the short pass (a third W1..W5 with 4- / 3-byte keys feeding the price-based
 * parse).  Parts that did not run read 0. */
int         gc_mf_pass_timing(gc_ctx* ctx, float ms[4]);

/* Optional in-kernel phase profile (shader-clock deltas measured by thread 0 of every workgroup, averaged over
 * blocks): cycles[0..6] = K1 {probe, insert, verify, double, chain, walk, emit}, cycles[7..11] = K3 {merge,
 * codes, tables, chains, pack}, cycles[12..15] = parts of K3's chains phase {stage, warm-up, walk, copy-out}.  Off by default (no cost when off). */
int         gc_zstd_set_phase_profile(gc_ctx* ctx, int enable);
int         gc_zstd_phase_profile(gc_ctx* ctx, double cyclesPerBlock[16]);

/* Workspace of a *_compress_device call (HBM, owned by the context, grown on demand and kept): the per-position arrays -- match records, sequences, literals, entropy-stage
 * staging -- take about 35 bytes per input byte (zstd, brotli; + 20 for FLZMA2); the finder's two entry lists take 8 bytes per LISTED position each (levels whose finder frames
 * overlap list a position once per frame that holds it: zstd 16-19 x 1.75, 20-22 x 3.25, FLZMA2 5-6 x 1.5, 7 x 1.9, 8-9 x 3.6) but never more than 24 GiB per list: a call
 * whose lists would be larger runs its match finder in up to eight parts, one after the other over the same lists (same bytes out).  Callers that want less go through
 * gc_multi_compress_host, whose pieces are 64 MiB (FLZMA2: 256 MiB) per context. */

/* ---- FLZMA2 (7-Zip method id 0x21): an LZMA2 chunk stream that the stock decoder NCompress::NLzma2::CDecoder
 * (C/Lzma2Dec.c; registered for FLZMA2 at CPP/7zip/Compress/FastLzma2Register.cpp:15) regenerates bit-exactly.
 *   gc_flzma2_compress_host   <->  the FL2_compressStream loop of NCompress::NLzma2::CFastEncoder::Code
 *                                  (CPP/7zip/Compress/Lzma2Encoder.cpp:260-350; library entry C/fast-lzma2/fl2_compress.c:1020)
 *   gc_flzma2_dict_prop       <->  FL2_getCCtxDictProp: the 1-byte coder property (Lzma2Encoder.cpp:353-364)
 *   gc_flzma2_compress_bound  <->  FL2_compressBound (fl2_compress.c:612)
 * flags: GC_FLZMA2_NO_END_MARK omits the terminating 0x00 so that the streams of several range shards can be concatenated
 * (the last shard writes it); every call starts with a dictionary reset, so shards are independent. */
#define GC_FLZMA2_NO_END_MARK 1u
size_t        gc_flzma2_compress_bound(size_t n);
unsigned char gc_flzma2_dict_prop(int level);
int           gc_flzma2_compress_device(gc_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t dstCapacity, int level, unsigned flags);
int           gc_flzma2_finish(gc_ctx* ctx, size_t* compressedSize);
int           gc_flzma2_compress_host(gc_ctx* ctx, const void* src, size_t n, void* dst, size_t dstCapacity, int level, unsigned flags,
                                      size_t* compressedSize);
/* ms[0..5] = lz (match finder), prep, model (symbols -> probabilities), rc (range coder), plan, emit;
 * ms[6] = first kernel start -> last kernel end */
int           gc_flzma2_last_timing(gc_ctx* ctx, float ms[7]);

/* ---- BROTLI (7-Zip method id 0x4F71102): brotli-mt framed chunks, each a complete brotli stream (RFC 7932), that
 * NCompress::NBROTLI::CDecoder (CPP/7zip/Compress/BrotliDecoder.cpp:124 -> BROTLIMT_decompressDCtx) regenerates bit-exactly.
 *   gc_brotli_compress_host   <->  BROTLIMT_compressCCtx as driven by NCompress::NBROTLI::CEncoder::Code
 *                                  (CPP/7zip/Compress/BrotliEncoder.cpp:118-164; C/zstdmt/brotli-mt_compress.c:209-333)
 * `level` = the reference's quality scale 0..11; it selects the chunk size (1 MiB x level, as brotli-mt does). */
size_t        gc_brotli_compress_bound(size_t n);
int           gc_brotli_compress_device(gc_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t dstCapacity, int level);
int           gc_brotli_finish(gc_ctx* ctx, size_t* compressedSize);
int           gc_brotli_compress_host(gc_ctx* ctx, const void* src, size_t n, void* dst, size_t dstCapacity, int level, size_t* compressedSize);
/* ms[0..3] = lz (match finder), block (histograms, prefix codes, bit stream), plan, emit; ms[4] = first kernel start -> last kernel end */
int           gc_brotli_last_timing(gc_ctx* ctx, float ms[5]);

/* ---- host-side building blocks shared by the three codecs, and the multi-GPU host scheduler
 * The reference's front ends split the input into independent jobs and run them on worker threads: ZSTDMT jobs
 * (C/zstd/zstdmt_compress.c:1184-1247), brotli-mt chunks (C/zstdmt/brotli-mt_compress.c:209-333), FL2 dictionary blocks
 * (C/fast-lzma2/fl2_compress.c:1020).  Here the workers are GPU contexts.
 *   gc_codec_grain          independence grain in bytes: ranges that start at a multiple of it compress independently
 *   gc_host_begin/size/fetch  one gc_*_compress_host call split into its three phases (H2D + enqueue, wait for the size, D2H)
 *   gc_host_alloc/free      pinned host memory (copies from / to it run at link speed and asynchronously)
 *   gc_multi_*              range-split of one host buffer over the contexts of one or more GPUs, two contexts per GPU by default
 *                           so that the PCIe copies of one piece overlap the kernels of another; compressed pieces are
 *                           concatenated in order.  FLZMA2: every piece is coded with GC_FLZMA2_NO_END_MARK and a single end
 *                           marker follows the last one (unless `flags` carries GC_FLZMA2_NO_END_MARK itself). */
#define GC_CODEC_ZSTD   0
#define GC_CODEC_FLZMA2 1
#define GC_CODEC_BROTLI 2
size_t      gc_codec_grain(int codec, int level);
size_t      gc_codec_compress_bound(int codec, size_t n);
int         gc_host_begin(gc_ctx* ctx, int codec, const void* src, size_t n, int level, unsigned flags);
int         gc_host_size(gc_ctx* ctx, size_t* compressedSize);
int         gc_host_fetch(gc_ctx* ctx, void* dst, size_t size);
int         gc_codec_compress_host(gc_ctx* ctx, int codec, const void* src, size_t n, void* dst, size_t dstCapacity, int level, unsigned flags,
                                   size_t* compressedSize);
/* ---- pre-processing on the device INSIDE the compress call ("fused into the same H2D pass"): the CRC-32 of the raw input (C/7zCrc.c) and one
 * of 7-Zip's pre-filters (C/Bra.c, C/Bra86.c, C/Delta.c) applied to the input where it lies in HBM, before the match finder -- ONE transfer for what the 7z
 * folder pipeline does in three host passes (the CRC in the reader CPP/7zip/Common/InOutTempBuffer / CInStreamWithCRC, CFilterCoder CPP/7zip/Common/
 * FilterCoder.cpp:160-262, then the coder; caller: CPP/7zip/Archive/7z/7zEncode.cpp:152-239).  The compressed stream is that of the FILTERED bytes: what a
 * folder "filter -> coder" holds.
 *   filter      0 = none, GC_BRA_* / GC_FILTER_X86 / GC_FILTER_DELTA as for gc_filter_host;  pc, delta, state: as there (state: 4 bytes x86, 256 Delta), in and out
 *   want_crc    != 0: crc receives CrcCalc(src, n) (init / final XOR 0xFFFFFFFF, polynomial 0xEDB88320)
 *   processed   out: bytes the converter has converted (the last few of a stream stay as they are, as at the end of CFilterCoder's stream)
 * ONE CALL = ONE WHOLE STREAM.  All n bytes of the call are compressed, the unconverted tail included, and nothing is carried into a next call: a branch instruction
 * that straddles the end of the call stays as it is, which is right at the end of a stream and WRONG in the middle of one (a decoder's filter converts across the seam).
 * The CRC is not resumable either.  A caller that cuts a stream into pieces must filter / sum on its side (gc_filter_host carries pc and state from call to call) and
 * use the plain entry points for the pieces; pc / state are "in" for streams that do not start at offset 0, "out" for inspection only. */
typedef struct gc_pre {
    int filter; uint32_t pc; unsigned delta; int want_crc;
    unsigned char state[256];
    uint32_t crc; uint64_t processed;
} gc_pre;
int         gc_host_begin_pre(gc_ctx* ctx, int codec, const void* src, size_t n, int level, unsigned flags, gc_pre* pre);
int         gc_codec_compress_host_pre(gc_ctx* ctx, int codec, const void* src, size_t n, void* dst, size_t dstCapacity, int level, unsigned flags,
                                       gc_pre* pre, size_t* compressedSize);
void*       gc_host_alloc(size_t n);
void        gc_host_free(void* p);

typedef struct gc_multi gc_multi;
/* devices == NULL or nDevices <= 0: every visible device.  ctxPerDevice <= 0: 2. */
int         gc_multi_create(gc_multi** out, const int* devices, int nDevices, int ctxPerDevice);
void        gc_multi_destroy(gc_multi* m);
int         gc_multi_workers(const gc_multi* m);
const char* gc_multi_last_error(const gc_multi* m);
/* default piece size for a codec and level: the multiple of the grain closest to 64 MiB (FLZMA2: 256 MiB) from below, at least one grain */
size_t      gc_multi_piece_bytes(int codec, int level);
/* pieceBytes == 0: gc_multi_piece_bytes(); otherwise rounded up to a multiple of the grain */
int         gc_multi_compress_host(gc_multi* m, int codec, const void* src, size_t n, void* dst, size_t dstCapacity, int level, unsigned flags,
                                   size_t pieceBytes, size_t* compressedSize);

/* ---- options for the bare-file handlers (CPP/7zip/Archive/ZstdHandler.cpp:273-280, BrotliHandler.cpp:286-291)
 *   GC_OPT_ZSTD_SEEK_TABLE  append a seek table -- a skippable frame (magic 0x184D2A5E) listing the compressed and decompressed size of every
 *                           zstd frame, footer magic 0x8F92EAB1: the "zstd seekable format" of zstd's contrib/seekable_format, which the
 *                           reference tree does not vendor -- so that a decoder can go frame-parallel without walking the stream.  Every zstd
 *                           decoder skips it (ZstdDecoder.cpp:145-158 accepts skippable frames).
 *   GC_OPT_BROTLI_PLAIN     ONE brotli stream without the brotli-mt frame headers: what BROTLIMT_compressCCtx writes for threads == 0
 *                           (C/zstdmt/brotli-mt_compress.c:462-466) and what a bare .br file is.  Copies still stay inside their chunk.
 * Options hold for the following calls of the context (0 = off, the default). */
#define GC_OPT_ZSTD_SEEK_TABLE 1
#define GC_OPT_BROTLI_PLAIN    2
/* flags of gc_host_begin / gc_codec_compress_host / gc_multi_compress_host for BROTLI: the call is one piece of ONE plain stream */
#define GC_BROTLI_PLAIN     1u
#define GC_BROTLI_NOT_FIRST 2u      /* ... and not its first piece: no stream header */
#define GC_BROTLI_NOT_LAST  4u      /* ... and not its last piece: no closing (ISLAST) meta-block */
int         gc_ctx_set_option(gc_ctx* ctx, int option, int value);
/* 1 in the test build of the library (csrc/libgpucodec_hooks.so: GC_* environment variables select code paths for the tests), 0 in the
 * shipped one, which reads no environment variable at all */
int         gc_test_hooks_enabled(void);

/* ---- CRC-32 of data that lies in device memory (C/7zCrc.c CrcCalc: polynomial 0xEDB88320, init and final XOR 0xFFFFFFFF).
 * Synchronous, on the current device's default stream. */
int         gc_crc32_device(const void* d_src, size_t n, uint32_t* crc);

/* ---- Branch converters on data in device memory (the filters 7-Zip puts in front of a compressor for executables:
 * z7_BranchConv_ARM64_Enc / _Dec ... of C/Bra.c:75-709, called from NCompress::NBranch::CCoder::Filter, CPP/7zip/Compress/BranchMisc.cpp:21-26).
 * pc = virtual address of byte 0 (the filters' kBranchOffset property); encoding != 0 converts relative -> absolute.  d_dst may equal d_src
 * except for GC_BRA_ARMT and GC_BRA_RISCV.  *processed = the byte count the reference's converter reports for one call on the whole buffer (bytes behind it are
 * copied unchanged).  Synchronous on the default stream. */
#define GC_BRA_ARM64 0
#define GC_BRA_ARM   1
#define GC_BRA_ARMT  2
#define GC_BRA_PPC   3
#define GC_BRA_SPARC 4
#define GC_BRA_IA64  5
#define GC_BRA_RISCV 6      /* out of place only, like GC_BRA_ARMT */
int         gc_bra_convert_device(int kind, const void* d_src, void* d_dst, size_t n, uint32_t pc, int encoding, size_t* processed);
/* The X86 converter ("BCJ": z7_BranchConvSt_X86_Enc / _Dec, C/Bra86.c:49-186): *state goes in and out as with the reference (0 at the start of a
 * stream, Z7_BRANCH_CONV_ST_X86_STATE_INIT_VAL); out of place only.  Converted bytes, *processed and *state equal the reference's for one call. */
int         gc_bra_x86_convert_device(const void* d_src, void* d_dst, size_t n, uint32_t pc, int encoding, uint32_t* state, size_t* processed);

/* The Delta filter (Delta_Encode / Delta_Decode, C/Delta.c:16-169; method 3 of the 7z pipeline) on data in device memory: delta 1..256;
 * state[256] in/out as the reference keeps it (zeros at the start of a stream: Delta_Init); encoding is out of place only. */
int         gc_delta_convert_device(const void* d_src, void* d_dst, size_t n, unsigned delta, int encoding, unsigned char state[256]);

/* One Filter() call of a 7-Zip pre-filter on a HOST buffer, in place -- what NCompress::NBranch::CCoder::Filter (CPP/7zip/Compress/BranchMisc.cpp:21-26),
 * NCompress::NBcj::CCoder2::Filter (BcjCoder.cpp:17-22) and NCompress::NDelta::CEncoder / CDecoder::Filter (DeltaFilter.cpp:47-51, :106-110) do with their
 * buffer: host -> device, the converter of gc_bra_* / gc_delta_* above, device -> host.  kind: a GC_BRA_* value, GC_FILTER_X86 or GC_FILTER_DELTA.
 * pc as above; `delta` 1..256 (GC_FILTER_DELTA only); `state`: 4 bytes for GC_FILTER_X86 (the converter's state word, 0 at the start of a stream), 256 for
 * GC_FILTER_DELTA, unused otherwise; *processed = the bytes converted (the caller presents the rest again with more data behind it, as 7-Zip's filter
 * coder does).  This is the path the plugin's filter objects (BCJGPU, ARM64GPU, DELTAGPU, ...) run on. */
#define GC_FILTER_X86   100
#define GC_FILTER_DELTA 101
int         gc_filter_host(gc_ctx* ctx, int kind, void* data, size_t n, uint32_t pc, int encoding, unsigned delta, unsigned char* state, size_t* processed);

/* ---- The mainline LZMA match finders on data in device memory: IMatchFinder2::GetMatches (C/LzFind.h:127-140) for EVERY
 * position of a buffer in one call -- exactly the values Hc4_MatchFinder_GetMatches (C/LzFind.c:1362) / Bt4_MatchFinder_GetMatches (:1219) write
 * there when the reference runs over the same buffer (MatchFinder_Create(historySize, 0, niceLen, ...), cutValue = cut), position by position.
 *   bt           0 = HC4 (hash chains), 1 = BT4 (binary trees)
 *   d_counts[i]  number of uint32 values of position i (two per match: length, distance - 1), stored at d_pairs[i * stride ...]
 * GC_ERR_DST_SMALL if a position has more than `stride` values (its list is cut).  n < 2^31 - 16; synchronous on the default stream. */
int         gc_lzfind_get_matches_device(const void* d_src, size_t n, int bt, uint32_t historySize, uint32_t cut, uint32_t niceLen,
                                         uint32_t* d_counts, uint32_t* d_pairs, uint32_t stride);

/* ---- ZSTD decoding on the device.  Replaces the ZSTD_decompressStream loop of NCompress::NZSTD::CDecoder::CodeSpec
 * (CPP/7zip/Compress/ZstdDecoder.cpp:66-240; C/zstd/zstd_decompress.c:2086) for callers that hold a whole compressed stream.
 * Entropy decoding (Huffman literals, FSE sequences) runs per BLOCK (<= 128 KiB, one workgroup each, whatever the frame structure);
 * the match copies run per FRAME (one workgroup each, blocks in order): streams of this engine's encoder carry one frame per 8 MiB, a
 * stream of the reference's encoder is one frame.  Frames with a dictionary id are refused (GC_ERR_PARAM); a damaged stream, a wrong content
 * checksum (XXH64, checked on the device) or a content size field that does not match give GC_ERR_CORRUPT.
 *   gc_zstd_scan_frames       host: walks frame and block headers (ZSTD_findFrameCompressedSize zstd_decompress.c:809, ZSTD_getFrameContentSize
 *                             :569), skips skippable frames.  frames may be NULL to count.  *contentTotal = sum of the content sizes, or
 *                             UINT64_MAX if a frame does not state its size (then the caller has to guess dstCapacity).
 *   gc_zstd_decompress_device frames as the scan returned them (host memory); d_src / d_dst device memory.  Frames that state their size are
 *                             decoded concurrently; a frame that does not ends a batch (its size is read back before the next batch starts).
 *   gc_zstd_decompress_host   scan + H2D + decode + D2H. */
typedef struct gc_zstd_frame {
    uint64_t src_off, src_size;      /* the frame inside the compressed stream */
    uint64_t dst_off;                /* where its content starts in the output; UINT64_MAX if unknown before decoding */
    uint64_t content_size;           /* valid if flags & 2 */
    uint32_t flags;                  /* 1: content checksum present, 2: content size known */
    uint32_t header_size;
    uint32_t n_blocks;               /* blocks of the frame (the unit of parallelism of the entropy stage) */
    uint32_t reserved;
} gc_zstd_frame;
int         gc_zstd_scan_frames(const void* src, size_t n, gc_zstd_frame* frames, size_t maxFrames, size_t* nFrames, uint64_t* contentTotal);
/* the same for a stream that is still being read: an input that ends inside a frame is not an error, *consumed = the bytes the whole
 * frames (and skippable frames) in front of it take -- what a streaming caller (the plugin's ICompressCoder::Code) can decode now */
int         gc_zstd_scan_prefix(const void* src, size_t n, gc_zstd_frame* frames, size_t maxFrames, size_t* nFrames, uint64_t* contentTotal,
                                size_t* consumed);
int         gc_zstd_decompress_device(gc_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t dstCapacity,
                                      const gc_zstd_frame* frames, size_t nFrames, size_t* decompressedSize);
int         gc_zstd_decompress_host(gc_ctx* ctx, const void* src, size_t n, void* dst, size_t dstCapacity, size_t* decompressedSize);
/* The decoder's run-time self-check: every context decodes one known frame of the reference's encoder through the sequences kernel that takes six blocks
 * per wave before it trusts that kernel (its lane-to-lane moves were once miscompiled into silently wrong match lengths; on a wrong result the context
 * uses the one-block-per-wave kernel).  Runs the check if the context has not done so yet; *state = 1 verified, -1 wrong on this device / build. */
int         gc_zstd_decompress_selfcheck(gc_ctx* ctx, int* state);
/* HIP-event duration of the decode kernels of the last gc_zstd_decompress_* call */
int         gc_zstd_decompress_timing(gc_ctx* ctx, float* ms);
/* ... and of its kernels: ms[0] index, ms[1] literals (second stream, beside the sequences), ms[2] sequences, ms[3] execution */
int         gc_zstd_decompress_kernel_timing(gc_ctx* ctx, float ms[4]);
/* pointer-jumping rounds of the last call when it took the wide execution path (all blocks of all frames at once: batches of few frames),
 * 0 when every batch went through the frame-per-workgroup execution kernel */
int         gc_zstd_decompress_wide_rounds(gc_ctx* ctx, unsigned* rounds);

/* ---- BROTLI decoding on the device.  Replaces BROTLIMT_decompressDCtx (C/zstdmt/brotli-mt_decompress.c:191-288, pt_read / pt_decompress) and the
 * BrotliDecoderDecompressStream loop under it (C/brotli/br_decode.c) as NCompress::NBROTLI::CDecoder::CodeSpec drives them (CPP/7zip/Compress/BrotliDecoder.cpp:124)
 * for callers that hold whole brotli-mt frames.  One wave per brotli-mt chunk (the format's unit of independence: inside a stream every literal's prefix code
 * depends on the two bytes in front of it); a bare RFC 7932 stream is ONE chunk and runs on one wave.
 *   gc_brotli_scan_prefix         host: walks the 16-byte frame headers {0x184D2A50, 8, compressed size, 0x5242, hint}; an input that ends inside a frame is not an
 *                                 error, *consumed = the bytes of the whole frames.  chunks may be NULL to count.  capacity = hint << 16 (the reference's output buffer).
 *   gc_brotli_decompress_device   chunks as the scan returned them (host memory), d_src / d_dst device memory; the content of the chunks in order, packed.
 *   gc_brotli_decompress_host     scan + H2D + decode + D2H; an input that does not start with a brotli-mt header is taken as one bare stream of at most dstCapacity bytes
 *                                 (brotli-mt_decompress.c:573, st_decompress).
 *   gc_brotli_dec_set_dictionary  the static dictionary of RFC 7932 Appendix A (122 784 bytes; checked against the CRC-32 the RFC states, 0x5136cb04), process-wide, copied.
 *                                 This library does not carry it: a host that has a brotli of its own passes BrotliGetDictionary()->data (C/brotli/common/dictionary.h).
 *                                 Without it a stream that refers to the dictionary gives GC_ERR_UNSUPPORTED; streams of this engine's encoder never refer to it.
 * GC_ERR_CORRUPT: damaged stream, or a chunk longer than its hint (brotli-mt_decompress.c:243 sizes the output buffer by it). */
typedef struct gc_brotli_chunk {
    uint64_t src_off;                /* the RFC 7932 stream of the chunk inside the compressed buffer (behind its 16-byte header) */
    uint32_t src_size;
    uint32_t capacity;               /* upper bound of its content in bytes */
} gc_brotli_chunk;
int         gc_brotli_scan_prefix(const void* src, size_t n, gc_brotli_chunk* chunks, size_t maxChunks, size_t* nChunks, uint64_t* capacityTotal, size_t* consumed);
int         gc_brotli_dec_set_dictionary(const void* data, size_t n);
int         gc_brotli_dec_has_dictionary(void);
int         gc_brotli_decompress_device(gc_ctx* ctx, const void* d_src, size_t n, void* d_dst, size_t dstCapacity,
                                        const gc_brotli_chunk* chunks, size_t nChunks, size_t* decompressedSize);
int         gc_brotli_decompress_host(gc_ctx* ctx, const void* src, size_t n, void* dst, size_t dstCapacity, size_t* decompressedSize);
/* HIP-event duration of the decode kernels of the last gc_brotli_decompress_* call */
int         gc_brotli_decompress_timing(gc_ctx* ctx, float* ms);

/* raw stream handle (hipStream_t) so callers can order their own work against the context */
void*       gc_ctx_stream(gc_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif
