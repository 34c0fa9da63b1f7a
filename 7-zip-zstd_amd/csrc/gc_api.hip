// gc_api.hip -- host side of libgpucodec.so: the C ABI declared in include/gpucodec.h.
//
// Mirrors what NCompress::NZSTD::CEncoder does around the library (CPP/7zip/Compress/ZstdEncoder.cpp:250-462):
// own a context, feed bytes, get a zstd stream back.  All work is enqueued on one HIP stream per context:
//   K1 lz -> (K2 huf || K3 seq on a second stream) -> K4 plan -> K5 emit.
// No CPU codec path exists here: if no gfx950 device can be opened every call fails with GC_ERR_NO_DEVICE.
#include "gpucodec.h"
#include "gc_common.h"
#include "gc_lzma2.h"
#include "gc_brotli.h"
#include "gc_mf.h"
#include "gc_zstd_dec.h"
#include "gc_brotli_dec.h"
#ifdef HIPEMU
#include "hip_runtime_stub.h"
#else
#include <hip/hip_runtime.h>
#define GC_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "gc_host_stream.h"
#ifdef GC_TEST_HOOKS
#include <vector>
#include <unistd.h>
#endif

struct GcFramePlan { uint64_t off; uint32_t size; uint32_t compressed; };

// FLZMA2 can process its input in up to GC_MAX_PARTS frame-aligned parts whose stages (match finder, model, range coder) run on
// different HIP streams, stage k of part p overlapping stage k-1 of part p+1.  Off by default: see gc_flzma2_compress_device.
#define GC_MAX_PARTS   8u
#define GC_PART_EVENTS 8u    // 0 finder start, 1 finder end (= prep start), 2 prep end, 3 model start, 4 model end, 5 rc start, 6 rc end

extern "C" __global__ void gc_zstd_lz_kernel(const uint8_t*, uint64_t, GcSeqRaw*, uint8_t*, GcBlockMeta*, unsigned long long*);
extern "C" __global__ void gc_zstd_huf_kernel(const uint8_t*, const GcBlockMeta*, uint8_t*, GcSectionInfo*);
extern "C" __global__ void gc_zstd_seq_codes_kernel(const GcSeqRaw*, const GcBlockMeta*, uint64_t*, uint32_t*, uint8_t*, GcSeqHist*, uint8_t*, GcSectionInfo*, uint32_t, unsigned long long*);
extern "C" __global__ void gc_zstd_seq_tables_kernel(const GcSeqHist*, const GcSectionInfo*, GcSeqTabG*, unsigned long long*);
extern "C" __global__ void gc_zstd_seq_chain_kernel(const uint8_t*, const GcSectionInfo*, GcSeqTabG*, uint16_t*, unsigned long long*);
extern "C" __global__ void gc_zstd_seq_pack_kernel(const uint64_t*, const uint8_t*, const uint16_t*, const GcSeqTabG*, uint8_t*, GcSectionInfo*, uint64_t, unsigned long long*);
extern "C" __global__ void gc_zstd_plan_kernel(const GcSectionInfo*, uint32_t, uint64_t, uint64_t, uint32_t, GcFramePlan*, uint64_t*, uint32_t);
extern "C" __global__ void gc_zstd_emit_kernel(const uint8_t*, uint64_t, const uint8_t*, const uint8_t*, const GcSectionInfo*,
                                               const GcFramePlan*, const uint64_t*, uint32_t, uint32_t, uint8_t*);

extern "C" __global__ void gc_mf_count_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_scan_kernel(uint32_t*, uint32_t);
extern "C" __global__ void gc_mf_scatter_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcMfEntry*);
extern "C" __global__ void gc_mf_link_kernel(const uint32_t*, const GcMfEntry*, GcMfEntry*, uint32_t, uint64_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_verify_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, uint32_t*);
extern "C" __global__ void gc_mf_count_short_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_scatter_short_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcMfEntry*);
extern "C" __global__ void gc_mf_verify_short_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, const uint32_t*, uint32_t*, const uint32_t*);
extern "C" __global__ void gc_mf_verify_shortb_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, const uint32_t*, uint32_t*, const uint32_t*);
extern "C" __global__ void gc_mf_count_far_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_scatter_far_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcMfEntry*);
extern "C" __global__ void gc_mf_verify_far_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, uint32_t*);
extern "C" __global__ void gc_mf_count_far2_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_scatter_far2_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcMfEntry*);
extern "C" __global__ void gc_mf_verify_far2_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, uint32_t*);
extern "C" __global__ void gc_mf_deepen_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t*, uint32_t*, uint32_t*);
extern "C" __global__ void gc_mf_count_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_scan_kernel_p8(uint32_t*, uint32_t);
extern "C" __global__ void gc_mf_scatter_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcMfEntry*);
extern "C" __global__ void gc_mf_link_kernel_p8(const uint32_t*, const GcMfEntry*, GcMfEntry*, uint32_t, uint64_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_verify_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, uint32_t*);
extern "C" __global__ void gc_mf_count_short_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_scatter_short_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcMfEntry*);
extern "C" __global__ void gc_mf_verify_short_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, const uint32_t*, uint32_t*, const uint32_t*);
extern "C" __global__ void gc_mf_verify_shortb_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, const uint32_t*, uint32_t*, const uint32_t*);
extern "C" __global__ void gc_mf_count_far_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_scatter_far_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcMfEntry*);
extern "C" __global__ void gc_mf_verify_far_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, uint32_t*);
extern "C" __global__ void gc_mf_count_far2_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t*);
extern "C" __global__ void gc_mf_scatter_far2_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcMfEntry*);
extern "C" __global__ void gc_mf_verify_far2_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, uint32_t*);
extern "C" __global__ void gc_mf_deepen_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t*, uint32_t*, uint32_t*);
extern "C" __global__ void gc_mf_vparse_tile_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, GcSeqRaw*, uint8_t*, GcBlockMeta*, uint32_t*, uint32_t*, unsigned long long*);
extern "C" __global__ void gc_mf_vparse_tile_kernel_p8(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t*, const GcMfEntry*, GcSeqRaw*, uint8_t*, GcBlockMeta*, uint32_t*, uint32_t*, unsigned long long*);
extern "C" __global__ void gc_mf_ringparse_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcSeqRaw*, uint8_t*, GcBlockMeta*);
extern "C" __global__ void gc_mf_parse_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, const uint32_t*, GcSeqRaw*, uint8_t*, GcBlockMeta*, uint16_t*, uint32_t);
extern "C" __global__ void gc_mf_short_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint16_t*);
extern "C" __global__ void gc_mf_dp2_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t, const uint32_t*, const uint16_t*, const uint16_t*, uint32_t*, uint32_t*);
extern "C" __global__ void gc_mf_dp3_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t, const uint32_t*, const uint16_t*, const uint16_t*, uint32_t*, uint32_t*);
extern "C" __global__ void gc_mf_litprice_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, const uint16_t*, uint32_t, uint8_t*);
extern "C" __global__ void gc_mf_dpl2_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t, const uint32_t*, const uint16_t*, const uint16_t*, uint32_t*, uint32_t*, const uint8_t*);
extern "C" __global__ void gc_mf_dpl2s_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t, const uint32_t*, const uint16_t*, const uint16_t*, uint32_t*, uint32_t*, const uint8_t*);
extern "C" __global__ void gc_mf_dplz_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t, const uint32_t*, const uint16_t*, const uint16_t*, uint32_t*, uint32_t*, const uint8_t*);
extern "C" __global__ void gc_mf_dplzs_kernel(const uint8_t*, uint64_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, uint32_t, const uint32_t*, const uint16_t*, const uint16_t*, uint32_t*, uint32_t*, const uint8_t*);

extern "C" __global__ void gc_lzma2_prep_kernel(const GcSeqRaw*, const GcBlockMeta*, uint64_t, uint64_t*, uint32_t*);
extern "C" __global__ void gc_lzma2_model_kernel(const uint8_t*, uint64_t, const uint64_t*, const uint32_t*, uint32_t, uint32_t, uint16_t*, GcLzmaChunkInfo*, const uint32_t*, unsigned long long*, uint32_t, uint32_t, uint32_t, uint8_t*, uint32_t, uint32_t, uint32_t);
extern "C" __global__ void gc_lzma2_segkind_kernel(const GcLzmaChunkInfo*, const uint8_t*, uint32_t, uint32_t, uint8_t*);
extern "C" __global__ void gc_lzma2_rc_kernel(uint16_t*, uint32_t, uint32_t, uint8_t*, GcLzmaChunkInfo*);
extern "C" __global__ void gc_lzma2_rc_fin_kernel(const uint16_t*, uint32_t, uint32_t, uint8_t*, GcLzmaChunkInfo*);
extern "C" __global__ void gc_lzma2_plan_kernel(const GcLzmaChunkInfo*, uint32_t, uint32_t, uint64_t, uint32_t, GcLzmaPlan*, uint64_t*, const uint8_t*, const uint8_t*);
extern "C" __global__ void gc_lzma2_emit_kernel(const uint8_t*, uint32_t, const uint8_t*, const GcLzmaChunkInfo*, const GcLzmaPlan*, uint32_t,
                                                uint32_t, const uint64_t*, uint8_t*, const uint8_t*);

extern "C" __global__ void gc_brotli_block_kernel(const uint8_t*, uint64_t, const GcSeqRaw*, const uint8_t*, const GcBlockMeta*, uint64_t*, uint32_t*,
                                                  uint32_t, uint32_t, uint32_t, uint32_t, uint32_t*, GcBrotliBlockInfo*);
extern "C" __global__ void gc_brotli_plan_kernel(const GcBrotliBlockInfo*, uint32_t, uint32_t, uint64_t, GcBrotliPlan*, uint64_t*);
extern "C" __global__ void gc_brotli_emit_kernel(const uint8_t*, uint64_t, const uint8_t*, const GcBrotliBlockInfo*, const GcBrotliPlan*, uint32_t,
                                                 uint32_t, uint32_t, const uint64_t*, uint8_t*);

struct gc_ctx {
    int device;
    hipStream_t stream;       // main stream: K1 -> K2 -> (join) -> K4 -> K5
    hipStream_t stream2;      // K3 runs here, concurrently with K2 (both only depend on K1); FLZMA2: model stage
    hipStream_t stream3;      // FLZMA2: range-coder stage
    hipEvent_t evPart[GC_MAX_PARTS][GC_PART_EVENTS];   // per input part: stage boundaries (see gc_flzma2_compress_device)
    uint32_t nParts;
    uint32_t lazyDepth;       // W6: 1 = one-step lazy, 2 = lazy2 (set per call from codec + level)
    uint32_t searchDepth;     // W5b: match links followed per position (0 = W5's two candidates only) -- by the starts of matches in tiles with long matches
    uint32_t searchShallow;   // ... and everywhere else (gc_mf_deepen_kernel)
    uint32_t shortPass;       // third finder pass with 4- / 3-byte keys; its merged records feed the price-based parse only
    uint32_t farPass;         // second finder pass with 16- / 12-byte keys (longer matches), merged into the records by gain
    uint32_t optSeekTable, optBrotliPlain;   // gc_ctx_set_option
    uint32_t mfFast;          // geometry of the windowed finder (gc_mf.h): 1 = 256 partitions / 8 KiB tiles, 0 = 1024 partitions / 16 KiB tiles
    uint32_t priceParse;      // W5s + W7: price-based parse on top of the greedy one (gc_lz_price.hip)
    uint32_t priceMinLen, priceLitCtx;        // its shortest match and literal context bits (LZMA: 2, 7; zstd: 3, 0)
    int lastCodecHint;        // codec of the call being enqueued (0 zstd, 1 flzma2, 2 brotli): which W7L kernels the finder launches
    uint32_t allLengths;      // W7L (zstd): every length of a candidate is an edge
    uint32_t smallWin2k;      // W7L: windows of 2 KiB in calls of <= 1 024 blocks (launch_finder_part)
    uint32_t shortPlain;      // overlapping frames: the pass with 4- / 3-byte keys runs over frames that tile the input (launch_finder_part)
    uint32_t farPass2;        // one more pass of the far kind with keys of 32 / 24 bytes (gc_lz_window.hip MF_FAR2)
    uint32_t laneParse;       // the price-based parse is W7L (a lane per window, repeat distances at every node) rather than W7
    uint32_t ringGeom;        // W6r: threads per block (16 per sub-block)
    uint32_t ringParse;       // W6r (brotli qualities 5-6): the parse walks the records in order with the last four distances as candidates; the value = shortest copy at a ring distance, 0 = W6
    uint32_t dbgFrameBlocks, dbgPartFrames;   // test hooks (env GC_FRAME_BLOCKS / GC_PART_FRAMES): small frames / parts so that
                                              // the multi-frame and multi-part paths can be exercised on small inputs
    hipEvent_t ev[8];         // 0 lz start, 1 lz end, 2 huf end, 3 seq start, 4 seq end, 5 plan start, 6 plan end, 7 emit end
    char err[256];
    // workspace, grown on demand
    uint32_t capBlocks;
    GcSeqRaw* seqRaw; uint8_t* lit; GcBlockMeta* meta;
    uint64_t* seqPacked; uint32_t* seqOff; uint8_t* codes; uint16_t* stOut; GcSeqHist* seqHist; GcSeqTabG* seqTabs;
    uint8_t* litSec; uint8_t* seqSec; GcSectionInfo* info; GcFramePlan* plan; uint64_t* result;
    uint8_t* lzProps;         // FLZMA2: props byte per model segment (lc / lp chosen by L2), 128 KiB >> GC_LZMA_SEG_LOG_MIN of them per block
    uint32_t* lzNM; GcLzmaChunkInfo* lzInfo; GcLzmaPlan* lzPlan; uint16_t* lzStream; size_t lzStreamCap;       // FLZMA2 path
    uint64_t* lzM; size_t lzMCap; uint8_t* lzRcOut; size_t lzRcOutCap;     // item lists, range-coder staging (allocated on the first FLZMA2 call)
    uint8_t* brStage; GcBrotliBlockInfo* brInfo; GcBrotliPlan* brPlan;    // BROTLI path
    // windowed match finder (gc_mf.h): counts/offsets, partition starts, entry lists; grown on demand
    uint32_t* mfTileWord; size_t mfTileWordCap;   // fused verify + parse: one word of counts per tile
    uint32_t nCU; uint32_t* mfTicket;      // compute units of the device; ticket counters of the persistent launches (4 per part)
    uint32_t* mfCnt; size_t mfCntCap; GcMfEntry* mfEnt; size_t mfEntCap; GcMfEntry* mfEnt2; size_t mfEnt2Cap; uint32_t* mfRec; size_t mfRecCap; uint32_t* mfRec2; size_t mfRec2Cap; uint32_t* mfChanged; size_t mfChangedCap;
    uint16_t* mfRec3; size_t mfRec3Cap; uint32_t* mfDp; size_t mfDpCap; uint16_t* mfPrice; size_t mfPriceCap; uint32_t* mfWinCost; size_t mfWinCostCap; uint32_t* mfDpStat; size_t mfDpStatCap; uint8_t* mfLitPrice; size_t mfLitPriceCap;      // (+ W7L: literal price per position) W5s records, W7 records, price tables, W7 phase-A symbol counts
    hipEvent_t evShort[GC_MAX_PARTS];       // W5s (near 2-3 byte candidates) runs beside the finder on stream2: done
    hipEvent_t evMf[GC_MAX_PARTS][13];      // per part: W1 start, W1 end, W2 end, W3 end, W4 end, W5 end, W6 end; price-based parse: greedy W6 end, W5s end, W7 end;
                                            // inside W5: first verify end, far pass end, deepen end
    bool mfPriced;                          // the last call ran the price-based parse (events 7..9 are valid)
    bool mfTimed; uint32_t mfParts;
    int lastCodec;            // 0 zstd, 1 flzma2, 2 brotli: which kernels the events of the last call bracket
    uint64_t* hostResult;     // pinned
    // staging for the host-buffer entry point
    GcBrDecWork brd;          // BROTLI decoder (gc_brotli_dec.hip)
    uint8_t* dIn; size_t dInCap; uint8_t* dOut; size_t dOutCap; uint8_t* dPre; size_t dPreCap;      // (dPre: the pre-filtered input of gc_host_begin_pre)
    bool pending; bool timed;
    // zstd decoder (gc_zstd_dec.hip): per-workgroup literal / sequence workspace, frame table, per-frame results, ticket counter
    uint8_t* zdLit; size_t zdLitCap; void* zdSeq; size_t zdSeqCap; GcZdFrame* zdFrames; size_t zdFramesCap; uint64_t* zdResult; uint64_t* zdTot;
    GcZdBlock* zdBlocks; size_t zdBlocksCap; uint32_t* zdOrder; size_t zdOrderCap; uint32_t* zdReady; size_t zdReadyCap; uint32_t* zdTicket; GcZdPlace* zdPlace; size_t zdPlaceCap; uint32_t* zdPtr; size_t zdPtrCap; uint8_t* zdDone; size_t zdDoneCap; uint32_t* zdFerr; size_t zdFerrCap; uint32_t zdRounds; hipEvent_t zdEv[2]; float zdMs; float zdKms[4]; int zdSeqvState /* 0 not checked yet, 1 verified on this device, -1 wrong: the one-block-per-wave kernel is used */; bool zdSelfTest;   // last call: whole, and index / literals / sequences / execution kernels
    unsigned long long* prof;  // device: GC_LZ_PHASES + GC_SEQ_PHASES cycle sums, only when profiling is on
    bool profOn; uint32_t profBlocks;
};

// Test hooks (environment variables) exist only in the TEST build of this file (-DGC_TEST_HOOKS: csrc/libgpucodec_hooks.so and the emulator
// library of tests/emu); in the shipped library gc_env_u32 is the constant `false`, no getenv is compiled in, and the bytes a call produces
// depend on its arguments alone.  Where a hook is read its value is validated: one outside [lo, hi] is ignored, so that no setting can push
// the kernels outside the geometry their entry formats were sized for (23-bit frame-relative positions, 8 MiB dictionary property, ...).
#ifdef GC_TEST_HOOKS
static bool gc_env_u32(const char* name, uint32_t lo, uint32_t hi, uint32_t* out)
{
    const char* e = getenv(name);
    if (!e || !*e) return false;
    char* end = nullptr;
    const long long v = strtoll(e, &end, 10);
    if (end == e || v < (long long)lo || v > (long long)hi) return false;
    *out = (uint32_t)v;
    return true;
}
extern "C" int gc_test_hooks_enabled(void) { return 1; }
#else
static inline bool gc_env_u32(const char*, uint32_t, uint32_t, uint32_t*) { return false; }
extern "C" int gc_test_hooks_enabled(void) { return 0; }
#endif

#define HIPCHK(ctx, call)                                                                       \
    do { hipError_t e_ = (call);                                                                \
         if (e_ != hipSuccess) { snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s", #call, hipGetErrorString(e_)); return GC_ERR_HIP; } } while (0)

extern "C" int gc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" size_t gc_zstd_compress_bound(size_t n)
{
    size_t nb = n ? (n + GC_ZSTD_BLOCK_MAX - 1) / GC_ZSTD_BLOCK_MAX : 1;
    return n + nb * GC_FRAME_OVERHEAD + 16 + nb * 8 + 17;       // (+ the optional seek table: 8 bytes per frame, 17 of header and footer)
}

static void free_workspace(gc_ctx* c);
void gc_brd_release(GcBrDecWork* w);
int gc_brd_decode(hipStream_t st, GcBrDecWork* w, const uint8_t* d_src, const gc_brotli_chunk* chunks, size_t nChunks, uint8_t* d_dst, size_t dstCap, size_t* produced, char* err, size_t errCap);
static void ctx_release(gc_ctx* c);

extern "C" int gc_ctx_create(gc_ctx** out, int device)
{
    if (!out) return GC_ERR_PARAM;
    *out = nullptr;
    int n = gc_device_count();
    if (n <= 0 || device < 0 || device >= n) return GC_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return GC_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GC_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return GC_ERR_NO_DEVICE;     // kernels are built for gfx950 only
    gc_ctx* c = new (std::nothrow) gc_ctx();
    if (!c) return GC_ERR_NOMEM;
    memset(c, 0, sizeof(*c));
    c->device = device;
    c->nCU = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    // every failure below releases what has been created so far (ctx_release skips what is still null)
    int rc = GC_OK;
    if (hipStreamCreate(&c->stream) != hipSuccess || hipStreamCreate(&c->stream2) != hipSuccess || hipStreamCreate(&c->stream3) != hipSuccess) rc = GC_ERR_HIP;
    for (int i = 0; rc == GC_OK && i < 8; i++) if (hipEventCreate(&c->ev[i]) != hipSuccess) rc = GC_ERR_HIP;
    for (uint32_t p = 0; rc == GC_OK && p < GC_MAX_PARTS; p++) {
        for (int i = 0; rc == GC_OK && i < 13; i++) if (hipEventCreate(&c->evMf[p][i]) != hipSuccess) rc = GC_ERR_HIP;
        if (rc == GC_OK && hipEventCreate(&c->evShort[p]) != hipSuccess) rc = GC_ERR_HIP;
        for (uint32_t i = 0; rc == GC_OK && i < GC_PART_EVENTS; i++) if (hipEventCreate(&c->evPart[p][i]) != hipSuccess) rc = GC_ERR_HIP;
    }
    if (rc == GC_OK && hipMalloc((void**)&c->prof, (GC_LZ_PHASES + GC_SEQ_PHASES) * sizeof(unsigned long long)) != hipSuccess) rc = GC_ERR_NOMEM;
    if (rc == GC_OK && (hipMalloc((void**)&c->mfTicket, GC_MAX_PARTS * 16u * sizeof(uint32_t)) != hipSuccess || hipMemsetAsync(c->mfTicket, 0, GC_MAX_PARTS * 16u * sizeof(uint32_t), c->stream) != hipSuccess)) rc = GC_ERR_NOMEM;
    if (rc == GC_OK && (hipMalloc((void**)&c->result, 16) != hipSuccess || hipHostMalloc((void**)&c->hostResult, 16 + GC_MAX_PARTS * 16u * sizeof(uint32_t)) != hipSuccess)) rc = GC_ERR_NOMEM;      // (+ a copy of the ticket / watchdog words)
    if (rc != GC_OK) { ctx_release(c); return rc; }
    c->dbgFrameBlocks = 0; c->dbgPartFrames = 0;
    gc_env_u32("GC_FRAME_BLOCKS", 1u, GC_MF_MAX_FRAME_BLOCKS, &c->dbgFrameBlocks);      // test hooks: small frames / parts
    gc_env_u32("GC_PART_FRAMES", 1u, 1u << 20, &c->dbgPartFrames);
    *out = c;
    return GC_OK;
}

static void free_workspace(gc_ctx* c)
{
    hipFree(c->seqRaw); hipFree(c->lit); hipFree(c->meta); hipFree(c->seqPacked); hipFree(c->seqOff); hipFree(c->codes);
    hipFree(c->stOut); hipFree(c->seqHist); hipFree(c->seqTabs); hipFree(c->litSec); hipFree(c->seqSec); hipFree(c->info); hipFree(c->plan);
    hipFree(c->brStage); hipFree(c->brInfo); hipFree(c->brPlan); c->brStage = nullptr; c->brInfo = nullptr; c->brPlan = nullptr;
    hipFree(c->lzProps); c->lzProps = nullptr; hipFree(c->lzNM); hipFree(c->lzInfo); hipFree(c->lzPlan); hipFree(c->lzStream); c->lzStream = nullptr; c->lzStreamCap = 0; c->lzNM = nullptr; c->lzInfo = nullptr; c->lzPlan = nullptr;
    hipFree(c->lzM); c->lzM = nullptr; c->lzMCap = 0; hipFree(c->lzRcOut); c->lzRcOut = nullptr; c->lzRcOutCap = 0;
    c->seqRaw = nullptr; c->lit = nullptr; c->meta = nullptr; c->seqPacked = nullptr; c->seqOff = nullptr; c->codes = nullptr;
    c->stOut = nullptr; c->seqHist = nullptr; c->seqTabs = nullptr; c->litSec = nullptr; c->seqSec = nullptr; c->info = nullptr; c->plan = nullptr; c->capBlocks = 0;
}

static void ctx_release(gc_ctx* c)
{
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    free_workspace(c);
    hipFree(c->prof); hipFree(c->mfTicket); hipFree(c->result); if (c->hostResult) hipHostFree(c->hostResult); hipFree(c->dIn); hipFree(c->dOut); hipFree(c->dPre);
    gc_brd_release(&c->brd);
    hipFree(c->zdLit); hipFree(c->zdSeq); hipFree(c->zdFrames); hipFree(c->zdResult); hipFree(c->zdTot); hipFree(c->zdBlocks); hipFree(c->zdOrder); hipFree(c->zdReady); hipFree(c->zdTicket); hipFree(c->zdPlace); hipFree(c->zdPtr); hipFree(c->zdDone); hipFree(c->zdFerr);
    for (int i = 0; i < 2; i++) if (c->zdEv[i]) hipEventDestroy(c->zdEv[i]);
    hipFree(c->mfTileWord); hipFree(c->mfCnt); hipFree(c->mfEnt); hipFree(c->mfEnt2); hipFree(c->mfRec); hipFree(c->mfRec2); hipFree(c->mfChanged); hipFree(c->mfRec3); hipFree(c->mfDp); hipFree(c->mfPrice); hipFree(c->mfWinCost); hipFree(c->mfDpStat); hipFree(c->mfLitPrice);
    for (int i = 0; i < 8; i++) if (c->ev[i]) hipEventDestroy(c->ev[i]);
    for (uint32_t p = 0; p < GC_MAX_PARTS; p++) {
        for (int i = 0; i < 13; i++) if (c->evMf[p][i]) hipEventDestroy(c->evMf[p][i]);
        if (c->evShort[p]) hipEventDestroy(c->evShort[p]);
        for (uint32_t i = 0; i < GC_PART_EVENTS; i++) if (c->evPart[p][i]) hipEventDestroy(c->evPart[p][i]);
    }
    if (c->stream3) hipStreamDestroy(c->stream3);
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void gc_ctx_destroy(gc_ctx* c)
{
    if (!c) return;
    ctx_release(c);
}

extern "C" int gc_ctx_set_option(gc_ctx* c, int option, int value)
{
    if (!c) return GC_ERR_PARAM;
    if (option == GC_OPT_ZSTD_SEEK_TABLE) c->optSeekTable = value != 0;
    else if (option == GC_OPT_BROTLI_PLAIN) c->optBrotliPlain = value ? ((uint32_t)value & 7u) | 1u : 0u;     // bit 0 plain, bits 1 / 2: GC_BROTLI_NOT_FIRST / GC_BROTLI_NOT_LAST
    else return GC_ERR_PARAM;
    return GC_OK;
}

extern "C" const char* gc_last_error_message(const gc_ctx* c) { return c ? c->err : "no context"; }
extern "C" void* gc_ctx_stream(gc_ctx* c) { return c ? (void*)c->stream : nullptr; }

static int ensure_workspace(gc_ctx* c, uint32_t nBlocks)
{
    if (nBlocks <= c->capBlocks) return GC_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_workspace(c);
    const size_t nb = nBlocks, ms = GC_MAX_SEQ_PER_BLOCK;
    if (hipMalloc((void**)&c->seqRaw, nb * ms * sizeof(GcSeqRaw)) != hipSuccess ||
        hipMalloc((void**)&c->lit, nb * GC_ZSTD_BLOCK_MAX) != hipSuccess ||
        hipMalloc((void**)&c->meta, nb * sizeof(GcBlockMeta)) != hipSuccess ||
        hipMalloc((void**)&c->seqPacked, nb * ms * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void**)&c->seqOff, nb * ms * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&c->codes, nb * ms * 3) != hipSuccess ||
        hipMalloc((void**)&c->stOut, nb * GC_SEQ_ST_STRIDE * 3 * sizeof(uint16_t)) != hipSuccess ||
        hipMalloc((void**)&c->seqHist, nb * sizeof(GcSeqHist)) != hipSuccess ||
        hipMalloc((void**)&c->seqTabs, nb * 3 * sizeof(GcSeqTabG)) != hipSuccess ||
        hipMalloc((void**)&c->litSec, nb * GC_LITSEC_STRIDE) != hipSuccess ||
        hipMalloc((void**)&c->seqSec, nb * GC_SEQSEC_STRIDE) != hipSuccess ||
        hipMalloc((void**)&c->info, nb * sizeof(GcSectionInfo)) != hipSuccess ||
        hipMalloc((void**)&c->plan, nb * sizeof(GcFramePlan)) != hipSuccess ||
        hipMalloc((void**)&c->brStage, nb * GC_BR_STAGE_STRIDE) != hipSuccess ||
        hipMalloc((void**)&c->brInfo, nb * sizeof(GcBrotliBlockInfo)) != hipSuccess ||
        hipMalloc((void**)&c->brPlan, nb * sizeof(GcBrotliPlan)) != hipSuccess ||
        hipMalloc((void**)&c->lzProps, 2u * nb * (GC_ZSTD_BLOCK_MAX >> GC_LZMA_SEG_LOG_MIN)) != hipSuccess ||      // (second half: LZMA / stored per model segment, gc_lzma2_segkind_kernel)
        hipMalloc((void**)&c->lzNM, nb * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&c->lzInfo, nb * GC_LZMA_RC_PER_BLOCK * sizeof(GcLzmaChunkInfo)) != hipSuccess ||
        hipMalloc((void**)&c->lzPlan, nb * GC_LZMA_RC_PER_BLOCK * sizeof(GcLzmaPlan)) != hipSuccess) {
        free_workspace(c);
        snprintf(c->err, sizeof(c->err), "workspace allocation for %u blocks failed", nBlocks);
        return GC_ERR_NOMEM;
    }
    c->capBlocks = nBlocks;
    return GC_OK;
}

// ------------------------------------------------------------------------------------------------ match finder dispatch
// frameBlocks == 1: K1, the block-local finder (hash tables in LDS, matches stay inside the 128 KiB block).
// frameBlocks  > 1: W1..W6, the windowed finder (gc_lz_window.hip): matches reach back to the start of the frame.
// Both leave the same interface behind: seqRaw / lit / meta per block.
static int mf_grow(gc_ctx* c, void** p, size_t* cap, size_t needBytes, const char* what)
{
    if (needBytes <= *cap) return GC_OK;
    void* np = nullptr;                                            // the old buffer stays valid if the growth fails
    if (hipMalloc(&np, needBytes) != hipSuccess) {
        hipFree(*p); *p = nullptr; *cap = 0;                       // second try with the old one released first
        if (hipMalloc(&np, needBytes) != hipSuccess) { snprintf(c->err, sizeof(c->err), "workspace (%s) of %zu bytes failed", what, needBytes); return GC_ERR_NOMEM; }
    }
    hipFree(*p); *p = np; *cap = needBytes;
    { uint32_t poison = 0; if (gc_env_u32("GC_POISON_WORKSPACE", 0u, 255u, &poison) && hipMemsetAsync(np, (int)poison, needBytes, c->stream) != hipSuccess) return GC_ERR_HIP; }   // test hook: the
                                                                   // finder's workspace starts filled with this byte -- what a recycled allocation holds is not zeros
    return GC_OK;
}

// workspace of the windowed finder for n input bytes
// The two entry lists (8 bytes per LISTED position each -- with overlapping frames a position is listed once per frame that holds it: FLZMA2 levels 8-9 list 29 frames of 8 MiB per
// 64 MiB group, 29 GiB per GiB of input and list) and the count table are what the finder's passes W1..W5 hand to each other and nothing behind the finder reads, and the parts of
// a call run their finder passes one after the other on the main stream: they are sized for the LARGEST PART and every part uses them from their start.  A call whose lists would
// exceed GC_MF_ENT_BUDGET is taken in as many parts as it needs (mf_auto_parts; parts never change the bytes), so the workspace of a direct call of any size is bounded by the
// per-position arrays (records, sequences, literals: ~35 bytes per input byte, + 20 for FLZMA2) plus this budget.
#define GC_MF_ENT_BUDGET ((size_t)24u << 30)                  // bytes per entry list and part
static uint32_t mf_auto_parts(size_t n, uint32_t frameArg)
{
    if (MF_F(frameArg) <= 1u) return 1u;
    const size_t perGroup = (size_t)MF_FPG(frameArg) * MF_F(frameArg) * GC_ZSTD_BLOCK_MAX * sizeof(GcMfEntry);
    const size_t nGroups = (gc_num_blocks(n) + MF_C(frameArg) - 1u) / MF_C(frameArg);
    size_t perPart = GC_MF_ENT_BUDGET / perGroup; if (perPart < 1u) perPart = 1u;
    size_t parts = (nGroups + perPart - 1u) / perPart;
    return (uint32_t)(parts < 1u ? 1u : (parts > GC_MAX_PARTS ? GC_MAX_PARTS : parts));
}
static int ensure_finder_workspace(gc_ctx* c, size_t n, uint32_t frameBlocks, size_t maxPartBytes = 0 /* bytes of the largest part; 0: one part */)
{
    if (MF_F(frameBlocks) <= 1u) return GC_OK;                  // (frameBlocks: F, or F | S << 8 | C << 16 -- overlapping frames, gc_mf.h)
    const GcMfGeom g = gc_mf_geom(n, frameBlocks, c->mfFast != 0u);
    const GcMfGeom gp = gc_mf_geom(maxPartBytes && maxPartBytes < n ? maxPartBytes : n, frameBlocks, c->mfFast != 0u);
    const size_t needCnt = gp.cntWords * sizeof(uint32_t), needEnt = (size_t)gp.nFrames * gp.frameBytes * sizeof(GcMfEntry);
    const size_t needRec = (size_t)g.nBlocks * GC_ZSTD_BLOCK_MAX * sizeof(uint32_t);
    const size_t needPrice = (size_t)g.nBlocks * GC_PRICE_WORDS * sizeof(uint16_t);
    const size_t needTileWord = ((size_t)g.nTiles + 64u) * sizeof(uint32_t);
    if (needTileWord > c->mfTileWordCap || needCnt > c->mfCntCap || needEnt > c->mfEntCap || needEnt > c->mfEnt2Cap || needRec > c->mfRecCap || ((c->searchDepth || c->shortPass) && (needRec > c->mfRec2Cap || needRec / 32u + 64u > c->mfChangedCap)) ||
        (c->priceParse && (needRec / 2u > c->mfRec3Cap || needRec > c->mfDpCap || needPrice > c->mfPriceCap || (size_t)g.nBlocks * 128u > c->mfWinCostCap || (size_t)g.nBlocks * GC_DPS_WORDS * 4u > c->mfDpStatCap || needRec / 4u > c->mfLitPriceCap))) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int rc;
        if ((rc = mf_grow(c, (void**)&c->mfTileWord, &c->mfTileWordCap, needTileWord, "tile counts")) != GC_OK) return rc;
        if ((rc = mf_grow(c, (void**)&c->mfCnt, &c->mfCntCap, needCnt, "offsets")) != GC_OK) return rc;
        if ((rc = mf_grow(c, (void**)&c->mfEnt, &c->mfEntCap, needEnt, "entries")) != GC_OK) return rc;
        if ((rc = mf_grow(c, (void**)&c->mfEnt2, &c->mfEnt2Cap, needEnt, "linked entries")) != GC_OK) return rc;
        if ((rc = mf_grow(c, (void**)&c->mfRec, &c->mfRecCap, needRec, "records")) != GC_OK) return rc;
        if ((c->searchDepth || c->shortPass) && (rc = mf_grow(c, (void**)&c->mfRec2, &c->mfRec2Cap, needRec, "deepened records")) != GC_OK) return rc;
        if ((c->searchDepth || c->shortPass) && (rc = mf_grow(c, (void**)&c->mfChanged, &c->mfChangedCap, needRec / 32u + 64u, "changed-record bitmap")) != GC_OK) return rc;
        if (c->priceParse) {
            if ((rc = mf_grow(c, (void**)&c->mfRec3, &c->mfRec3Cap, needRec / 2u, "short candidates")) != GC_OK) return rc;
            if ((rc = mf_grow(c, (void**)&c->mfDp, &c->mfDpCap, needRec, "price-parse records")) != GC_OK) return rc;
            if ((rc = mf_grow(c, (void**)&c->mfPrice, &c->mfPriceCap, needPrice, "price tables")) != GC_OK) return rc;
            if ((rc = mf_grow(c, (void**)&c->mfWinCost, &c->mfWinCostCap, (size_t)g.nBlocks * 128u, "window costs")) != GC_OK) return rc;
            if ((rc = mf_grow(c, (void**)&c->mfDpStat, &c->mfDpStatCap, (size_t)g.nBlocks * GC_DPS_WORDS * 4u, "path symbol counts")) != GC_OK) return rc;
            if ((rc = mf_grow(c, (void**)&c->mfLitPrice, &c->mfLitPriceCap, needRec / 4u, "literal prices")) != GC_OK) return rc;
        }
    }
    return GC_OK;
}

// Match finder for one part of the input: `src` / `n` are the part, blk0 its first block (a multiple of frameBlocks: parts are
// whole frames, so everything inside is relative to the part and only the workspace pointers are offset).
static int launch_finder_part(gc_ctx* c, hipStream_t st, uint32_t part, const uint8_t* src, size_t n, uint32_t frameArg /* F, or F | S << 8 | C << 16: overlapping frames (gc_mf.h) */, uint32_t blk0, unsigned long long* prof)
{
    const uint32_t frameBlocks = frameArg;                           // what W1..W5b take (they decode it: mf_tile)
    const uint32_t groupBlocks = MF_C(frameArg);                     // what the stages behind the finder call a frame: matches reach back to the start of the GROUP
    const uint32_t nBlocks = gc_num_blocks(n);
    GcSeqRaw* seqRaw = c->seqRaw + (size_t)blk0 * GC_MAX_SEQ_PER_BLOCK;
    uint8_t* lit = c->lit + (size_t)blk0 * GC_ZSTD_BLOCK_MAX;
    GcBlockMeta* meta = c->meta + blk0;
    if (MF_F(frameArg) <= 1u) {
        GC_LAUNCH(gc_zstd_lz_kernel, nBlocks, 1024, st, src, (uint64_t)n, seqRaw, lit, meta, prof);
        return GC_OK;
    }
    const GcMfGeom g = gc_mf_geom(n, frameBlocks, c->mfFast != 0u);
    const uint32_t frame0 = (blk0 / groupBlocks) * MF_FPG(frameArg);       // (parts are whole groups)
    uint32_t* cnt = c->mfCnt;                                              // count table and entry lists: every part from their start (ensure_finder_workspace)
    GcMfEntry* ent = c->mfEnt;
    GcMfEntry* ent2 = c->mfEnt2;
    uint32_t* rec = c->mfRec + (size_t)blk0 * GC_ZSTD_BLOCK_MAX;
    const uint32_t perT = gc_xcd_per(g.nTiles), perB = gc_xcd_per(nBlocks);
    const bool fast = c->mfFast != 0u;
    const uint32_t nParts = 1u << g.partLog;
#define MFSEL(k) (fast ? k##_p8 : k)          // the kernel of this geometry
    // W4 is a persistent launch: one-wave workgroups, six per CU (24 KiB of LDS each), fed from a ticket counter (one counter per launch and part)
    uint32_t linkLaunch = 0;
    uint32_t linkWpc = 6u; gc_env_u32("GC_LINK_WPC", 1u, 32u, &linkWpc);         // test hook: one-wave workgroups per CU
    const uint32_t nLists = g.nFrames * nParts, linkGrid = nLists * GC_MF_LINK_SEGS < c->nCU * linkWpc ? nLists * GC_MF_LINK_SEGS : c->nCU * linkWpc;
    HIPCHK(c, hipMemsetAsync(c->mfTicket + part * 16u, 0, 16u * sizeof(uint32_t), st));      // words 0..3: W4's launches, 8..15: the fused kernel's XCD classes
#define MF_LINK(cnt_, ent_, ent2_) do { uint32_t* ticket_ = c->mfTicket + part * 16u + linkLaunch++; \
        GC_LAUNCH(MFSEL(gc_mf_link_kernel), linkGrid, 64, st, (const uint32_t*)(cnt_), (const GcMfEntry*)(ent_), ent2_, g.tilesPerFrame, g.frameBytes, nLists, ticket_); } while (0)
    hipEvent_t* ev = c->evMf[part];
    HIPCHK(c, hipEventRecord(ev[0], st));
    // W5s reads nothing but the input: it runs beside the finder's passes on the context's second stream (round 5; it was 1.5 ms of the main stream's chain per 212 MB).  stream2 is
    // also where the stages behind the finder run, later in the same call: stream order keeps them apart.
    const bool shortBeside = c->priceParse && st == c->stream;
    if (shortBeside) {
        HIPCHK(c, hipEventRecord(c->evShort[part], st)); HIPCHK(c, hipStreamWaitEvent(c->stream2, c->evShort[part], 0));        // (the input and the workspace are ready where `st` stands now)
        const uint32_t nChunkWg = (uint32_t)(((n + 2047u) / 2048u + 3u) / 4u), perC = gc_xcd_per(nChunkWg);
        GC_LAUNCH(gc_mf_short_kernel, perC * GC_XCDS, 256, c->stream2, src, (uint64_t)n, groupBlocks, (uint32_t)((n + 2047u) / 2048u), perC, c->mfRec3 + (size_t)blk0 * GC_ZSTD_BLOCK_MAX);
        HIPCHK(c, hipEventRecord(c->evShort[part], c->stream2));
    }
    GC_LAUNCH(MFSEL(gc_mf_count_kernel), perT * GC_XCDS, nParts, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, cnt);
    HIPCHK(c, hipEventRecord(ev[1], st));
    GC_LAUNCH(MFSEL(gc_mf_scan_kernel), g.nFrames, 1024, st, cnt, g.tilesPerFrame);
    HIPCHK(c, hipEventRecord(ev[2], st));
    GC_LAUNCH(MFSEL(gc_mf_scatter_kernel), perT * GC_XCDS, nParts, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, (const uint32_t*)cnt, ent);
    HIPCHK(c, hipEventRecord(ev[3], st));
    MF_LINK(cnt, ent, ent2);
    HIPCHK(c, hipEventRecord(ev[4], st));
    // the levels that parse the first pass's records as they are: verify + parse in one kernel, the records stay in LDS (W5 + W6 fused)
    uint32_t fused = (!c->farPass && !c->farPass2 && !c->searchDepth && !c->priceParse && !c->shortPass) ? 1u : 0u;
    gc_env_u32("GC_FUSED_PARSE", 0u, 1u, &fused);               // test hook: 0 = the two kernels
    if (fused && (c->farPass || c->farPass2 || c->searchDepth || c->priceParse || c->shortPass || MF_C(frameArg) != MF_F(frameArg))) fused = 0u;
    if (fused) {
        {
            const uint32_t tpb = GC_ZSTD_BLOCK_MAX >> g.tileLog;
            const uint32_t perV = ((gc_xcd_per(g.nTiles) + tpb - 1u) / tpb) * tpb;      // whole blocks per XCD class: a tile never waits for a tile of another class
            uint32_t* tw = c->mfTileWord + (size_t)frame0 * g.tilesPerFrame;
            HIPCHK(c, hipMemsetAsync(tw, 0, (size_t)g.nTiles * sizeof(uint32_t), st));
            GC_LAUNCH(MFSEL(gc_mf_vparse_tile_kernel), perV * GC_XCDS, g.verifyT, st, src, (uint64_t)n, frameBlocks, g.nTiles, perV, c->lazyDepth, (const uint32_t*)cnt,
                      (const GcMfEntry*)ent2, seqRaw, lit, meta, c->mfTicket + part * 16u + 8u, tw, prof);
        }
        for (int i = 10; i <= 12; i++) HIPCHK(c, hipEventRecord(ev[i], st));
        HIPCHK(c, hipEventRecord(ev[5], st));
        HIPCHK(c, hipEventRecord(ev[6], st));
        (void)prof;
        return GC_OK;
    }
    GC_LAUNCH(MFSEL(gc_mf_verify_kernel), perT * GC_XCDS, g.verifyT, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, (const uint32_t*)cnt,
                   (const GcMfEntry*)ent2, rec);
    HIPCHK(c, hipEventRecord(ev[10], st));
    if (c->farPass) {                                           // second pass with 16- / 12-byte keys, merged into rec (timed with W5)
        GC_LAUNCH(MFSEL(gc_mf_count_far_kernel), perT * GC_XCDS, nParts, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, cnt);
        GC_LAUNCH(MFSEL(gc_mf_scan_kernel), g.nFrames, 1024, st, cnt, g.tilesPerFrame);
        GC_LAUNCH(MFSEL(gc_mf_scatter_far_kernel), perT * GC_XCDS, nParts, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, (const uint32_t*)cnt, ent);
        MF_LINK(cnt, ent, ent2);
        GC_LAUNCH(MFSEL(gc_mf_verify_far_kernel), perT * GC_XCDS, g.verifyT, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, (const uint32_t*)cnt,
                  (const GcMfEntry*)ent2, rec);
    }
    if (c->farPass2) {                                          // third pass of the far kind: keys of 32 / 24 bytes, capped records ranked by what lies behind the cap (gc_lz_window.hip MF_FAR2; timed with W5)
        GC_LAUNCH(MFSEL(gc_mf_count_far2_kernel), perT * GC_XCDS, nParts, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, cnt);
        GC_LAUNCH(MFSEL(gc_mf_scan_kernel), g.nFrames, 1024, st, cnt, g.tilesPerFrame);
        GC_LAUNCH(MFSEL(gc_mf_scatter_far2_kernel), perT * GC_XCDS, nParts, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, (const uint32_t*)cnt, ent);
        MF_LINK(cnt, ent, ent2);
        GC_LAUNCH(MFSEL(gc_mf_verify_far2_kernel), perT * GC_XCDS, g.verifyT, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, (const uint32_t*)cnt,
                  (const GcMfEntry*)ent2, rec);
    }
    HIPCHK(c, hipEventRecord(ev[11], st));
    uint32_t* const chg = (c->searchDepth && c->shortPass && c->priceParse) ? c->mfChanged + (size_t)blk0 * (GC_ZSTD_BLOCK_MAX / 32u) : (uint32_t*)nullptr;      // which records W5b changes: what the pass with 4- / 3-byte keys looks at again (gc_lz_window.hip "continuation")
    if (c->searchDepth) {                                       // W5b: follow match links (timed with W5)
        uint32_t* rec2 = c->mfRec2 + (size_t)blk0 * GC_ZSTD_BLOCK_MAX;
        GC_LAUNCH(MFSEL(gc_mf_deepen_kernel), perT * GC_XCDS, 256, st, src, (uint64_t)n, frameBlocks, g.nTiles, perT, c->searchDepth | (c->searchShallow << 8), (const uint32_t*)rec, rec2, chg);
        rec = rec2;
    }
    HIPCHK(c, hipEventRecord(ev[12], st));
    const uint32_t litCtxArg = c->priceLitCtx | (blk0 != 0u ? 0x80000000u : 0u);   // bit 31: a later part -- the byte in front of src exists
    const uint32_t* recDp = rec;                                // what W7 reads: the records, or the records + short candidates
    if (c->priceParse && c->shortPass) {                        // third pass with 4- / 3-byte keys (timed with W5)
        uint32_t* recN = rec == c->mfRec + (size_t)blk0 * GC_ZSTD_BLOCK_MAX ? c->mfRec2 + (size_t)blk0 * GC_ZSTD_BLOCK_MAX : c->mfRec + (size_t)blk0 * GC_ZSTD_BLOCK_MAX;
        // Overlapping frames list and link every position once per frame that holds it.  A match of 3-4 bytes MiBs back is never worth its distance, so this pass runs
        // over frames that tile the input (the plain geometry F: a frame then lies inside its group, whose start is as far back as the stages behind the finder let a match reach)
        const uint32_t fbS = c->shortPlain ? MF_F(frameArg) : frameArg;                 // (the levels that are not after speed keep the overlap: the generator of lz-7zip copies 3-4 bytes from anywhere in its window, 32 MiB at FLZMA2 level 7: +0.26 % without)
        const GcMfGeom gs = gc_mf_geom(n, fbS, c->mfFast != 0u);
        const uint32_t perTs = gc_xcd_per(gs.nTiles), nListsS = gs.nFrames * nParts;
        uint32_t* cntS = c->mfCnt;
        GC_LAUNCH(MFSEL(gc_mf_count_short_kernel), perTs * GC_XCDS, nParts, st, src, (uint64_t)n, fbS, gs.nTiles, perTs, cntS);
        GC_LAUNCH(MFSEL(gc_mf_scan_kernel), gs.nFrames, 1024, st, cntS, gs.tilesPerFrame);
        GC_LAUNCH(MFSEL(gc_mf_scatter_short_kernel), perTs * GC_XCDS, nParts, st, src, (uint64_t)n, fbS, gs.nTiles, perTs, (const uint32_t*)cntS, ent);
        {   uint32_t* ticket_ = c->mfTicket + part * 16u + linkLaunch++;
            const uint32_t gridS = nListsS * GC_MF_LINK_SEGS < c->nCU * linkWpc ? nListsS * GC_MF_LINK_SEGS : c->nCU * linkWpc;
            GC_LAUNCH(MFSEL(gc_mf_link_kernel), gridS, 64, st, (const uint32_t*)cntS, (const GcMfEntry*)ent, ent2, gs.tilesPerFrame, gs.frameBytes, nListsS, ticket_); }
        if (c->shortPlain) GC_LAUNCH(MFSEL(gc_mf_verify_short_kernel), perTs * GC_XCDS, gs.verifyT, st, src, (uint64_t)n, fbS, gs.nTiles, perTs, (const uint32_t*)cntS,
                  (const GcMfEntry*)ent2, (const uint32_t*)rec, recN, (const uint32_t*)chg);
        else GC_LAUNCH(MFSEL(gc_mf_verify_shortb_kernel), perTs * GC_XCDS, gs.verifyT, st, src, (uint64_t)n, fbS, gs.nTiles, perTs, (const uint32_t*)cntS,
                  (const GcMfEntry*)ent2, (const uint32_t*)rec, recN, (const uint32_t*)chg);      // (with the catch-up: gc_lz_window.hip)
        recDp = recN;
    }
    HIPCHK(c, hipEventRecord(ev[5], st));
    if (c->priceParse) {
        // greedy parse first (its symbol statistics become the block's price table), then the price-based parse W7 over the same
        // candidates + the short ones of W5s, written as records that W6 follows as they are (lazy 0)
        uint16_t* price = c->mfPrice + (size_t)blk0 * GC_PRICE_WORDS;
        uint16_t* rec3 = c->mfRec3 + (size_t)blk0 * GC_ZSTD_BLOCK_MAX;
        uint32_t* dp = c->mfDp + (size_t)blk0 * GC_ZSTD_BLOCK_MAX;
        uint32_t* wc = c->mfWinCost + (size_t)blk0 * 32u;
        GC_LAUNCH(gc_mf_parse_kernel, perB * GC_XCDS, GC_MF_PARSE_T, st, src, (uint64_t)n, nBlocks, perB, c->lazyDepth, (const uint32_t*)rec, seqRaw, lit, meta, price, litCtxArg);
        HIPCHK(c, hipEventRecord(ev[7], st));
        const uint32_t nChunkWg = (uint32_t)(((n + 2047u) / 2048u + 3u) / 4u), perC = gc_xcd_per(nChunkWg);
        if (shortBeside) HIPCHK(c, hipStreamWaitEvent(st, c->evShort[part], 0));
        else GC_LAUNCH(gc_mf_short_kernel, perC * GC_XCDS, 256, st, src, (uint64_t)n, groupBlocks, (uint32_t)((n + 2047u) / 2048u), perC, rec3);
        HIPCHK(c, hipEventRecord(ev[8], st));
        // W7 in two phases: A = a sample of the windows (one workgroup of 4 windows per block) under optimistic prices, counting the symbols of its
        // paths; B = every window under prices made from those counts (gc_lz_price.hip)
        uint32_t* dps = c->mfDpStat + (size_t)blk0 * GC_DPS_WORDS;
        HIPCHK(c, hipMemsetAsync(dps, 0, (size_t)nBlocks * GC_DPS_WORDS * sizeof(uint32_t), st));
        uint32_t phase0 = 0; { uint32_t one = 0; if (gc_env_u32("GC_DP_PHASES", 1u, 2u, &one) && one == 1u) phase0 = 2u; }    // test hook: 1 = W6's prices only
        // W7L (gc_lz_dpl.hip): one lane per window; LZMA with the four repeat distances at every node.  Test hook GC_DPL: 0 = W7 (a wave per window)
        // 1: W7L everywhere (FLZMA2); 2: phase A in W7L, phase B per block in W7 or W7L by what phase A's paths did (gc_mf.h GC_DPS_RICH) -- zstd, where text repeats
        // an offset in 1 % of its sequences and sources / binaries in 8-46 %: 125 MB of text at level 19 3.78 -> 4.75 GB/s (run r4s), the sizes of W7L where it matters.
        // (FLZMA2 on the Silesia stand-in: half of the blocks sit right at the threshold, and a W7L launch takes as long for a few blocks as for all of them -- it ends with
        // its slowest wave, and all of its waves fit the device at once -- so the two kernels' times add up: 40 -> 49 ms.)  Test hook GC_DPL: 0 = W7 with its own phase A
        uint32_t laneDp = c->laneParse ? (c->lastCodecHint == 1 ? 1u : 2u) : 0u; gc_env_u32("GC_DPL", 0u, 2u, &laneDp);
        if (c->lastCodecHint == 2 && !c->laneParse) laneDp = 0u;                        // (brotli, hook GC_BR_LANE=0: W7 only.  Qualities 8-11 run zstd's W7L kernels since round 6: the parse's last distances at every node, which B1 then codes as ring
                                                                                        //  entries; 32 MiB at quality 9: shared objects 1.023 -> 0.980 x the reference, real sources 1.092 -> 1.038, text / web-text / lz-7zip as before -- phase B goes to W7 there)
        uint8_t* lpr = c->mfLitPrice + (size_t)blk0 * GC_ZSTD_BLOCK_MAX;
        if (laneDp) GC_LAUNCH(gc_mf_litprice_kernel, perB * GC_XCDS, 256, st, src, (uint64_t)n, nBlocks, perB, (const uint16_t*)price, litCtxArg, lpr);
        // (Round 5 built a re-priced SECOND pass over every window -- the first full pass counts its own paths, the second prices from those counts -- behind a hook: text -0.14 .. -0.25 %,
        //  real sources +1.6 %, FLZMA2 on shared objects -0.04 %, profiles/r05_zstd19.md.  Measured, not taken; removed in round 6.)
        for (uint32_t pass = phase0 == 2u ? 1u : 0u; pass < 2u; pass++) {
            const uint32_t phase = phase0 == 2u ? 2u : (pass == 0u ? 0u : 1u);
            const uint32_t nDpWg = nBlocks * (phase == 0u ? 1u : 8u), perD = gc_xcd_per(nDpWg);
            uint32_t* wcp = phase == 0u ? (uint32_t*)nullptr : wc;
            if (laneDp) {
                // Windows of 2 KiB (one block per wave) where the 4 KiB ones leave the device half empty: W7L takes as long as ONE wave needs for its window, whatever the
                // number of waves, as long as they are all resident (1 024 groups of 64 windows); a call of <= 1 024 blocks (128 MiB) has at most 512 groups of 4 KiB
                // windows.  Round 4 measured 32 MiB 23.5 -> 13.6 ms and 128 MiB 25.4 -> 15.7 ms for +0.02 % (the Silesia stand-in) ... +0.36 % (shared objects) and left it
                // off because the size bars sat at the band's edge; round 5 (merged model segments, overlapping frames) moved them: on at FLZMA2 levels 5-6.
                uint32_t win2k = (c->smallWin2k && nBlocks <= 1024u) ? 1u : 0u; gc_env_u32("GC_DPL_WIN2K", 0u, 1u, &win2k);      // test hook
                const bool w2 = win2k != 0u && phase != 0u;
                const uint32_t nItems = w2 ? nBlocks : (nBlocks + 1u) / 2u, perL = gc_xcd_per(nItems);     // a wave = two blocks (2 KiB windows: one)
                const bool select = laneDp == 2u && phase == 1u && !w2;
                const uint32_t phaseK = phase | (w2 ? 16u : 0u) | (select ? GC_DP_SELECT : 0u) | (c->allLengths ? GC_DP_ALLLEN : 0u);
                if (select) {                                      // the blocks without repeats: W7
                    if (c->priceMinLen <= 2u) GC_LAUNCH(gc_mf_dp2_kernel, perD * GC_XCDS, 256, st, src, (uint64_t)n, nBlocks, perD, groupBlocks, phaseK, dps, litCtxArg, recDp, (const uint16_t*)rec3, (const uint16_t*)price, dp, wcp);
                    else GC_LAUNCH(gc_mf_dp3_kernel, perD * GC_XCDS, 256, st, src, (uint64_t)n, nBlocks, perD, groupBlocks, phaseK, dps, litCtxArg, recDp, (const uint16_t*)rec3, (const uint16_t*)price, dp, wcp);
                }
                if (c->priceMinLen <= 2u) {
                    if (phase == 0u) GC_LAUNCH(gc_mf_dpl2s_kernel, perL * GC_XCDS, GC_DPL_THREADS, st, src, (uint64_t)n, nBlocks, perL, groupBlocks, phase | (c->allLengths ? GC_DP_ALLLEN : 0u), dps, litCtxArg, recDp, (const uint16_t*)rec3, (const uint16_t*)price, dp, wcp, (const uint8_t*)lpr);
                    else GC_LAUNCH(gc_mf_dpl2_kernel, perL * GC_XCDS, GC_DPL_THREADS, st, src, (uint64_t)n, nBlocks, perL, groupBlocks, phaseK, dps, litCtxArg, recDp, (const uint16_t*)rec3, (const uint16_t*)price, dp, wcp, (const uint8_t*)lpr);
                } else {                                           // zstd, and brotli with the ring's first entries standing in for zstd's repeat offsets
                    if (phase == 0u) GC_LAUNCH(gc_mf_dplzs_kernel, perL * GC_XCDS, GC_DPL_THREADS, st, src, (uint64_t)n, nBlocks, perL, groupBlocks, phase | (c->allLengths ? GC_DP_ALLLEN : 0u), dps, litCtxArg, recDp, (const uint16_t*)rec3, (const uint16_t*)price, dp, wcp, (const uint8_t*)lpr);
                    else GC_LAUNCH(gc_mf_dplz_kernel, perL * GC_XCDS, GC_DPL_THREADS, st, src, (uint64_t)n, nBlocks, perL, groupBlocks, phaseK, dps, litCtxArg, recDp, (const uint16_t*)rec3, (const uint16_t*)price, dp, wcp, (const uint8_t*)lpr);
                }
            } else
            if (c->priceMinLen <= 2u) GC_LAUNCH(gc_mf_dp2_kernel, perD * GC_XCDS, 256, st, src, (uint64_t)n, nBlocks, perD, groupBlocks, phase, dps, litCtxArg, recDp, (const uint16_t*)rec3, (const uint16_t*)price, dp, wcp);
            else GC_LAUNCH(gc_mf_dp3_kernel, perD * GC_XCDS, 256, st, src, (uint64_t)n, nBlocks, perD, groupBlocks, phase, dps, litCtxArg, recDp, (const uint16_t*)rec3, (const uint16_t*)price, dp, wcp);
        }
        HIPCHK(c, hipEventRecord(ev[9], st));
        GC_LAUNCH(gc_mf_parse_kernel, perB * GC_XCDS, GC_MF_PARSE_T, st, src, (uint64_t)n, nBlocks, perB, 0u, (const uint32_t*)dp, seqRaw, lit, meta, (uint16_t*)nullptr, 0u);
    } else if (c->ringParse) {
        if ((c->ringParse >> 8) & 0xFFu)                               // W6r on the blocks that W6's parse shows to come back to their last distances
            GC_LAUNCH(gc_mf_parse_kernel, perB * GC_XCDS, GC_MF_PARSE_T, st, src, (uint64_t)n, nBlocks, perB, c->lazyDepth, (const uint32_t*)rec, seqRaw, lit, meta, (uint16_t*)nullptr, 0u);
        GC_LAUNCH(gc_mf_ringparse_kernel, perB * GC_XCDS, c->ringGeom, st, src, (uint64_t)n, nBlocks, perB, c->lazyDepth < 1u ? c->lazyDepth : 1u, c->ringParse, (const uint32_t*)rec, seqRaw, lit, meta);
    }
    else
        GC_LAUNCH(gc_mf_parse_kernel, perB * GC_XCDS, GC_MF_PARSE_T, st, src, (uint64_t)n, nBlocks, perB, c->lazyDepth, (const uint32_t*)rec, seqRaw, lit, meta, (uint16_t*)nullptr, 0u);
    HIPCHK(c, hipEventRecord(ev[6], st));
    (void)prof;
    return GC_OK;
}

// whole input as one part on the main stream
static int launch_finder(gc_ctx* c, const uint8_t* src, size_t n, uint32_t frameBlocks, unsigned long long* prof)
{
    c->mfTimed = false;
    int rc = ensure_finder_workspace(c, n, frameBlocks);
    if (rc != GC_OK) return rc;
    rc = launch_finder_part(c, c->stream, 0, src, n, frameBlocks, 0, prof);
    if (rc != GC_OK) return rc;
    c->mfTimed = MF_F(frameBlocks) > 1u; c->mfParts = 1; c->mfPriced = c->mfTimed && c->priceParse != 0u;
    return GC_OK;
}

// ms[0..5] = count, scan, scatter, link, verify, parse of the windowed match finder in the last call (after *_finish), summed
// over the parts of the input
extern "C" int gc_mf_last_timing(gc_ctx* c, float ms[6])
{
    if (!c || !c->timed || c->pending || !c->mfTimed) return GC_ERR_PARAM;
    for (int i = 0; i < 6; i++) ms[i] = 0.f;
    for (uint32_t p = 0; p < c->mfParts; p++)
        for (int i = 0; i < 6; i++) { float t = 0.f; HIPCHK(c, hipEventElapsedTime(&t, c->evMf[p][i], c->evMf[p][i + 1])); ms[i] += t; }
    return GC_OK;
}

// ms[0..3] = greedy parse (W6 + statistics), short candidates (W5s), shortest path (W7), second W6 pass that follows W7's records:
// the parts of gc_mf_last_timing's "parse" entry when the last call ran the price-based parse (GC_ERR_PARAM otherwise)
extern "C" int gc_mf_price_timing(gc_ctx* c, float ms[4])
{
    if (!c || !c->timed || c->pending || !c->mfTimed || !c->mfPriced) return GC_ERR_PARAM;
    static const int a[4] = { 5, 7, 8, 9 }, b[4] = { 7, 8, 9, 6 };
    for (int i = 0; i < 4; i++) ms[i] = 0.f;
    for (uint32_t p = 0; p < c->mfParts; p++)
        for (int i = 0; i < 4; i++) { float t = 0.f; HIPCHK(c, hipEventElapsedTime(&t, c->evMf[p][a[i]], c->evMf[p][b[i]])); ms[i] += t; }
    return GC_OK;
}

// ms[0..3] = the parts of gc_mf_last_timing's "verify" entry: W5 itself, the far pass (W1'..W5' with 16- / 12-byte keys), W5b link
// following, the short pass (W1"..W5" with 4- / 3-byte keys); a part that did not run reads 0
extern "C" int gc_mf_pass_timing(gc_ctx* c, float ms[4])
{
    if (!c || !c->timed || c->pending || !c->mfTimed) return GC_ERR_PARAM;
    static const int a[4] = { 4, 10, 11, 12 }, b[4] = { 10, 11, 12, 5 };
    for (int i = 0; i < 4; i++) ms[i] = 0.f;
    for (uint32_t p = 0; p < c->mfParts; p++)
        for (int i = 0; i < 4; i++) { float t = 0.f; HIPCHK(c, hipEventElapsedTime(&t, c->evMf[p][a[i]], c->evMf[p][b[i]])); ms[i] += t; }
    return GC_OK;
}

// zstd level -> blocks per frame.  Levels 1-2 (the reference's `fast` strategy, clevels.h:29-30) use the block-local finder and
// one frame per block; level 3 and up (dfast and stronger, clevels.h:31-47, windowLog >= 21) use the windowed finder with
// 8 MiB frames.
// Every level runs the windowed finder over 8 MiB frames (round 3).  Levels 1-2 used the block-local finder before -- a 128 KiB window against the
// 512 KiB / 1 MiB windows of the reference's levels 1 / 2 (clevels.h:26-27): 1.026 x its level 1 on text, 1.085 x its level 2 (run r03_levels).
// The block-local kernel K1 still serves inputs of one block.
static uint32_t zstd_frame_blocks(int level) { return level >= 18 ? GC_MF_WIDE_MAX_FRAME_BLOCKS : GC_MF_MAX_FRAME_BLOCKS; }       // (round 6: 16 MiB windows at 18-22 -- the reference: windowLog 23 at 18-19, 25-27 at 20-22, clevels.h:46-50)
// Levels 16-22: the finder's frames overlap (gc_mf.h "Overlapping frames") inside groups that are the zstd frames.  The reference: windowLog 22 at level 16-17, 23 at 18-19,
// 25 / 26 / 27 at 20 / 21 / 22 (clevels.h:44-50), one frame, ZSTDMT jobs of four windows overlapping by one (zstdmt_compress.c:741-747).  Here the window stays 8 MiB
// (23-bit positions); what the levels choose is how much of it a position is sure to have behind it: stride 4 MiB = 4-8 MiB of history at 16-19, stride 2 MiB = 6-8 MiB at
// 20-22 (each halving of the stride lists and links every position once more: W1..W4 of the three passes).
static uint32_t zstd_group_blocks(int level) { return level >= 16 ? 4u * GC_MF_MAX_FRAME_BLOCKS : GC_MF_MAX_FRAME_BLOCKS; }     // 32 MiB zstd frames (= shard grain) / 8 MiB
static uint32_t zstd_stride_blocks(int level) { return (level == 18 || level == 19) ? GC_MF_MAX_FRAME_BLOCKS : GC_MF_MAX_FRAME_BLOCKS / 2u; }      // a position is sure of 4 MiB of history at 16-17 (8 MiB windows every 4 MiB), of 8 MiB -- the reference's whole window -- at 18-19 (16 MiB windows every 8 MiB), of 12 MiB at 20-22 (every 4 MiB; round 5: 8 MiB windows every 2 MiB)
// zstd level -> match links followed per position (the reference's searchLog grows the same way: clevels.h:25-47)
static uint32_t zstd_search_depth(int level) { return level < 6 ? 0u : (level < 10 ? 2u : (level < 16 ? 4u : (level < 18 ? 8u : 16u))); }

extern "C" int gc_zstd_compress_device(gc_ctx* c, const void* d_src, size_t n, void* d_dst, size_t dstCap, int level)
{
    if (!c || (!d_src && n) || !d_dst) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    c->timed = false;
    if (n == 0) {
        // empty input: one frame with FCS=0 and an empty raw last block (ZSTD_compress on 0 bytes does the same)
        static const uint8_t empty[9] = { 0x28, 0xB5, 0x2F, 0xFD, 0x20, 0x00, 0x01, 0x00, 0x00 };
        if (dstCap < sizeof(empty)) return GC_ERR_DST_SMALL;
        HIPCHK(c, hipMemcpyAsync(d_dst, empty, sizeof(empty), hipMemcpyHostToDevice, c->stream));
        c->hostResult[0] = sizeof(empty); c->hostResult[1] = 0;
        HIPCHK(c, hipMemcpyAsync(c->result, c->hostResult, 16, hipMemcpyHostToDevice, c->stream));
        c->pending = true;
        return GC_OK;
    }
    const uint32_t nBlocks = gc_num_blocks(n);
    int rc = ensure_workspace(c, nBlocks);
    if (rc != GC_OK) return rc;
    const uint8_t* src = (const uint8_t*)d_src;
    if (c->profOn) { HIPCHK(c, hipMemsetAsync(c->prof, 0, (GC_LZ_PHASES + GC_SEQ_PHASES) * sizeof(unsigned long long), c->stream)); c->profBlocks = nBlocks; }
    uint32_t frameBlocks = zstd_frame_blocks(level);
    if (frameBlocks > 1u && c->dbgFrameBlocks) frameBlocks = c->dbgFrameBlocks;                              // test hook: small frames
    if (frameBlocks > nBlocks) frameBlocks = nBlocks;                                                       // short input: one frame
    c->lazyDepth = level >= 6 ? 2u : 1u;          // the reference's lazy2 begins at level 8 of its table; deeper look-ahead from 6 here
    c->mfFast = level <= 6 ? 1u : 0u;             // no far pass below level 7: the fast geometry (gc_mf.h)
    gc_env_u32("GC_MF_FAST", 0u, 1u, &c->mfFast);                                              // test hook
    // (Listing half of the positions, chosen by content, was measured at level 3 in rounds 2 and 5 -- 1 GB of text 43.2 -> 39.0 ms for +3.5-4.5 % size, real sources 0.949 -> 1.158 x
    //  the reference -- and again in round 6's lab with the catch-up in place (tools/zstd_parse_lab.c policy 4: shared objects +10 %): not taken, its kernels are gone.)
    c->searchDepth = zstd_search_depth(level); c->searchShallow = c->searchDepth < 2u ? c->searchDepth : 2u;
    if (gc_env_u32("GC_SEARCH_DEPTH", 0u, 64u, &c->searchDepth)) c->searchShallow = c->searchDepth < 2u ? c->searchDepth : 2u;      // test hook
    gc_env_u32("GC_SEARCH_SHALLOW", 0u, 64u, &c->searchShallow);                                // test hook: links followed by every position
    c->ringParse = 0u;
    c->farPass = level >= 5 ? 1u : 0u;            // where the reference searches chains / trees (lazy2 and up).  Measured (run 29, 32 MiB): level 9
                                                  // 1.027 -> 0.984 x the reference on text, level 12 1.040 -> 1.001 x.  Round 6: from level 5 (was 7) -- the reference's greedy / lazy
                                                  // strategies at 5-6 walk hash chains (zstd_lazy.c:667, searchLog 3: clevels.h:33-34), and on real sources the first pass alone was
                                                  // 1.135 x its level 5 and 1.136 x its level 6 (32 / 8 MiB, run s4: nobody had looked); with the two far passes 0.932 / 0.984 (emulator, 8 MiB)
    c->shortPass = level >= 5 ? 1u : 0u;          // the reference's btopt strategies search 3-byte matches (minMatch 3, clevels.h:44-47); from level 10 since round 3, from level 7 since round 6, see priceParse
    gc_env_u32("GC_FAR_PASS", 0u, 1u, &c->farPass); gc_env_u32("GC_SHORT_PASS", 0u, 1u, &c->shortPass);   // test hooks
    c->shortPlain = 0u; c->smallWin2k = 0u;
    c->allLengths = level >= 18 ? 1u : 0u;        // levels 16-17 (the reference: btopt / btultra with searchLog 5) keep the sparse lengths and the two far passes; 18-22 (btultra / btultra2, searchLog 6-9) price every length
    c->farPass2 = (level >= 5 && level != 16 && level != 17) ? 1u : 0u; gc_env_u32("GC_FAR2_PASS", 0u, 1u, &c->farPass2);      // keys of 32 / 24 bytes where the reference searches chains / trees for the LONGEST match: real sources, emulator, 8 MiB:
                                                                                              // level 9 1.118 -> 1.050 x the reference, level 19 1.115 -> 1.097
    c->laneParse = level >= 16 ? 1u : 0u;         // the reference's btopt .. btultra2 (clevels.h:44-50) price its three repeat offsets at every position; real sources / binaries at level 19
                                                  // (emulator, 4 MiB): 1.109 / 1.124 x the reference with W7, 1.081 / 1.075 with W7L.  Levels 10-15 (the reference: lazy2 / btlazy2) keep W7
    c->lastCodecHint = 0; c->priceMinLen = 3u; c->priceLitCtx = 0u;     // zstd: matches of >= 3 bytes, literals without context (one Huffman table per block)
    c->priceParse = level >= 5 ? 1u : 0u;         // the reference's btopt / btultra strategies start at level 16 (clevels.h:44-47), its levels 10-15 are lazy2 / btlazy2 over deep
                                                  // chains and trees; the greedy / lazy2 parse over this finder's 3-6 candidates was 1.03 x them on lz-7zip (levels 10 and 12, run r03_z12),
                                                  // the price-based parse 0.98 -- so it started at level 10 in round 3.  Round 6: from level 7, where the far passes start.  The reference's
                                                  // lazy2 at 7-9 picks the longest of 16-32 tagged row candidates (zstd_lazy.c:1141); the lazy parse over this finder's gain-merged record was
                                                  // 1.058-1.062 x it on real shared objects (level 7 LARGER than level 5: a far match that wins by `4 len - log2 offset` is often dearer than the
                                                  // near one it replaces); with the short pass + the price-based parse 1.026 / 1.028 (emulator, 8 MiB; real sources 1.038 -> 0.978, text 0.942 -> 0.909).
                                                  // And from level 5 once the far passes start there: levels 5 / 6 on shared objects 1.043 / 1.048 with the lazy parse over the merged records, 1.014 / 1.017 with
                                                  // this one (real sources 0.891).  Levels 5-12 now differ in the geometry (5-6: the fast one), the links followed (0 / 2 / 4) and nothing else
    gc_env_u32("GC_PRICE_PARSE", 0u, 1u, &c->priceParse);                                      // test hook
    // Overlapping finder frames (gc_mf.h) from level 16: the zstd frame becomes the GROUP (the reference's own frames are the whole input with a sliding window).
    uint32_t zGroup = zstd_group_blocks(level); const bool grpHook = gc_env_u32("GC_MF_GROUP", 1u, 65535u, &zGroup);      // test hook: blocks per group (with GC_FRAME_BLOCKS and GC_MF_STRIDE: overlapping frames of a few blocks)
    uint32_t zArg = frameBlocks;                                                               // what the finder takes
    if (zGroup > frameBlocks && (frameBlocks == zstd_frame_blocks(level) || grpHook) && nBlocks > frameBlocks) {
        uint32_t stride = zstd_stride_blocks(level); gc_env_u32("GC_MF_STRIDE", 1u, GC_MF_MAX_FRAME_BLOCKS, &stride);      // test hook (blocks; 64 = no overlap)
        if (stride < frameBlocks && (frameBlocks % stride) == 0u && ((zGroup - frameBlocks) % stride) == 0u) zArg = GC_MF_GEOM_ARG(frameBlocks, stride, zGroup);
    }
    const uint32_t zFrameBlocks = MF_C(zArg);                                                  // blocks per zstd frame
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    // The input can be taken in frame-aligned PARTS, the finder of part p + 1 (main stream) beside the entropy stage of part p (sequences on
    // stream2, literals on stream3).  Measured on MI355X (run r3_g, 1 GB of text at level 3): 1 part 30.9 ms, 2 parts 32.2, 4 parts 32.0,
    // 8 parts 35.2; 100 MB: 3.9 / 5.0 / 7.1 ms -- kernels that run beside each other take the CUs' LDS and wave slots from one another and
    // every part pays its own launch tails.  One part it is; the hook keeps the path exercised.
    const uint32_t nFrames = (nBlocks + zFrameBlocks - 1u) / zFrameBlocks;                      // (zstd frames = the finder's groups)
    uint32_t nParts = 1u;
    gc_env_u32("GC_ZSTD_PARTS", 1u, GC_MAX_PARTS, &nParts);                                    // test hook
    if (c->dbgPartFrames) nParts = nFrames / c->dbgPartFrames;
    { const uint32_t autoParts = mf_auto_parts(n, zArg); if (nParts < autoParts) nParts = autoParts; }      // (entry lists beyond the budget: ensure_finder_workspace)
    if (nParts > nFrames) nParts = nFrames;
    if (nParts > GC_MAX_PARTS) nParts = GC_MAX_PARTS;
    if (nParts < 1u) nParts = 1u;
    c->mfTimed = false;
    const uint32_t framesPerPart = (nFrames + nParts - 1u) / nParts;
    rc = ensure_finder_workspace(c, n, zArg, (size_t)framesPerPart * zFrameBlocks * GC_ZSTD_BLOCK_MAX);
    if (rc != GC_OK) return rc;
    uint32_t usedParts = 0;
    for (uint32_t p = 0; p < nParts; p++) {
        const uint32_t blk0 = p * framesPerPart * zFrameBlocks;
        if (blk0 >= nBlocks) break;
        const size_t off = (size_t)blk0 * GC_ZSTD_BLOCK_MAX;
        const size_t len = (size_t)framesPerPart * zFrameBlocks * GC_ZSTD_BLOCK_MAX < n - off ? (size_t)framesPerPart * zFrameBlocks * GC_ZSTD_BLOCK_MAX : n - off;
        const uint32_t pBlocks = gc_num_blocks(len);
        rc = launch_finder_part(c, c->stream, p, src + off, len, zArg, blk0, c->profOn ? c->prof : nullptr);
        if (rc != GC_OK) return rc;
        HIPCHK(c, hipEventRecord(c->evPart[p][0], c->stream));                                 // finder of this part done
        if (p + 1u == nParts || blk0 + pBlocks >= nBlocks) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        // K2 (literals) and K3 (sequences) are independent consumers of the finder: two more streams
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->evPart[p][0], 0));
        HIPCHK(c, hipStreamWaitEvent(c->stream3, c->evPart[p][0], 0));
        if (p == 0u) HIPCHK(c, hipEventRecord(c->ev[3], c->stream2));
        {   // K3: codes -> tables -> state chains -> pack (gc_zstd_seq.hip)
            unsigned long long* sprof = c->profOn ? c->prof + GC_LZ_PHASES : nullptr;
            uint8_t* codes = c->codes + (size_t)blk0 * 3u * GC_MAX_SEQ_PER_BLOCK;
            uint64_t* packed = c->seqPacked + (size_t)blk0 * GC_MAX_SEQ_PER_BLOCK;
            uint16_t* st = c->stOut + (size_t)blk0 * 3u * GC_SEQ_ST_STRIDE;
            GC_LAUNCH(gc_zstd_seq_codes_kernel, pBlocks, GC_SEQ_T, c->stream2, (const GcSeqRaw*)(c->seqRaw + (size_t)blk0 * GC_MAX_SEQ_PER_BLOCK), (const GcBlockMeta*)(c->meta + blk0),
                      packed, c->seqOff + (size_t)blk0 * GC_MAX_SEQ_PER_BLOCK, codes, c->seqHist + blk0, c->seqSec + (size_t)blk0 * GC_SEQSEC_STRIDE, c->info + blk0, zFrameBlocks, sprof);
            GC_LAUNCH(gc_zstd_seq_tables_kernel, pBlocks * 3u, 64, c->stream2, (const GcSeqHist*)(c->seqHist + blk0), (const GcSectionInfo*)(c->info + blk0), c->seqTabs + (size_t)blk0 * 3u, sprof);
            GC_LAUNCH(gc_zstd_seq_chain_kernel, pBlocks * 3u, 64, c->stream2, (const uint8_t*)codes, (const GcSectionInfo*)(c->info + blk0), c->seqTabs + (size_t)blk0 * 3u, st, sprof);
            GC_LAUNCH(gc_zstd_seq_pack_kernel, pBlocks, GC_SEQ_T, c->stream2, (const uint64_t*)packed, (const uint8_t*)codes, (const uint16_t*)st, (const GcSeqTabG*)(c->seqTabs + (size_t)blk0 * 3u),
                      c->seqSec + (size_t)blk0 * GC_SEQSEC_STRIDE, c->info + blk0, (uint64_t)len, sprof);
        }
        GC_LAUNCH(gc_zstd_huf_kernel, pBlocks, 256, c->stream3, (const uint8_t*)(c->lit + (size_t)blk0 * GC_ZSTD_BLOCK_MAX), (const GcBlockMeta*)(c->meta + blk0),
                  c->litSec + (size_t)blk0 * GC_LITSEC_STRIDE, c->info + blk0);
        usedParts = p + 1u;
    }
    c->mfTimed = frameBlocks > 1u; c->mfParts = usedParts; c->mfPriced = c->mfTimed && c->priceParse != 0u;
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream2));
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream3));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev[2], 0));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev[4], 0));
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    GC_LAUNCH(gc_zstd_plan_kernel, 1, 1024, c->stream, (const GcSectionInfo*)c->info, nBlocks, (uint64_t)n, (uint64_t)dstCap, zFrameBlocks, c->plan, c->result, c->optSeekTable);
    HIPCHK(c, hipEventRecord(c->ev[6], c->stream));
    GC_LAUNCH(gc_zstd_emit_kernel, nBlocks + (c->optSeekTable ? 1u : 0u), 256, c->stream, src, (uint64_t)n, (const uint8_t*)c->litSec, (const uint8_t*)c->seqSec,
              (const GcSectionInfo*)c->info, (const GcFramePlan*)c->plan, (const uint64_t*)c->result, nBlocks, zFrameBlocks, (uint8_t*)d_dst);
    HIPCHK(c, hipEventRecord(c->ev[7], c->stream));
    HIPCHK(c, hipGetLastError());
    c->pending = true; c->timed = true; c->lastCodec = 0;
    return GC_OK;
}

extern "C" int gc_zstd_finish(gc_ctx* c, size_t* compressedSize)
{
    if (!c || !c->pending) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(c->hostResult, c->result, 16, hipMemcpyDeviceToHost, c->stream));
    uint32_t* const words = (uint32_t*)(c->hostResult + 2);          // word 7 of a part: a kernel's bounded wait for another workgroup ran out (gc_lz_window.hip, fused verify + parse)
    HIPCHK(c, hipMemcpyAsync(words, c->mfTicket, GC_MAX_PARTS * 16u * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pending = false;
    { uint32_t trip = 0; gc_env_u32("GC_WATCHDOG_TRIP", 0u, 1u, &trip);       // test hook: as if a wait had run out
      for (uint32_t p = 0; p < GC_MAX_PARTS; p++) trip |= words[p * 16u + 7u];
      if (trip) {       // (a part's words are only zeroed when that part is launched again: clear them all, or a later call that uses fewer parts would report this trip for ever)
          hipMemsetAsync(c->mfTicket, 0, GC_MAX_PARTS * 16u * sizeof(uint32_t), c->stream); hipStreamSynchronize(c->stream);
          snprintf(c->err, sizeof(c->err), "a kernel gave up waiting for another workgroup (watchdog): the output of this call is not valid"); return GC_ERR_HIP; } }
#ifdef GC_TEST_HOOKS
    if (const char* dir = getenv("GC_DUMP_STATE")) {               // test hook: the parse's intermediate state of this call, for comparing two runs offline
        static int serial = 0;
        char path[512]; snprintf(path, sizeof(path), "%s/state_%d_%d.bin", dir, (int)getpid(), serial++);
        if (FILE* f = fopen(path, "wb")) {
            const uint32_t nb = c->capBlocks;
            struct { const char* name; const void* p; size_t bytes; } parts[] = {
                { "wincost", c->mfWinCost, c->mfWinCost ? (size_t)nb * 128u : 0u }, { "dpstat", c->mfDpStat, c->mfDpStat ? (size_t)nb * GC_DPS_WORDS * 4u : 0u },
                { "price", c->mfPrice, c->mfPrice ? (size_t)nb * GC_PRICE_WORDS * 2u : 0u }, { "lzinfo", c->lzInfo, c->lzInfo ? (size_t)nb * GC_LZMA_RC_PER_BLOCK * sizeof(GcLzmaChunkInfo) : 0u },
                { "lznm", c->lzNM, c->lzNM ? (size_t)nb * 4u : 0u },
                { "dp", c->mfDp, (c->mfDp && nb <= 16u) ? (size_t)nb * GC_ZSTD_BLOCK_MAX * 4u : 0u } };      // (W7 / W7L records: small inputs only)
            for (auto& q : parts) {
                std::vector<uint8_t> h(q.bytes);
                if (q.bytes && hipMemcpy(h.data(), q.p, q.bytes, hipMemcpyDeviceToHost) != hipSuccess) h.assign(q.bytes, 0xEE);
                char hdr[32] = { 0 }; snprintf(hdr, sizeof(hdr), "%s", q.name); fwrite(hdr, 1, 16, f);
                const uint64_t nbytes = q.bytes; fwrite(&nbytes, 8, 1, f); fwrite(h.data(), 1, q.bytes, f);
            }
            fclose(f);
        }
    }
#endif
    if (c->hostResult[1]) { snprintf(c->err, sizeof(c->err), "destination too small: need %llu bytes", (unsigned long long)c->hostResult[0]); return GC_ERR_DST_SMALL; }
    if (compressedSize) *compressedSize = (size_t)c->hostResult[0];
    return GC_OK;
}

extern "C" int gc_zstd_last_timing(gc_ctx* c, float ms[6])
{
    if (!c || !c->timed || c->pending || c->lastCodec != 0) return GC_ERR_PARAM;
    HIPCHK(c, hipEventElapsedTime(&ms[0], c->ev[0], c->ev[1]));     // lz
    HIPCHK(c, hipEventElapsedTime(&ms[1], c->ev[1], c->ev[2]));     // huf (runs concurrently with seq)
    HIPCHK(c, hipEventElapsedTime(&ms[2], c->ev[3], c->ev[4]));     // seq
    HIPCHK(c, hipEventElapsedTime(&ms[3], c->ev[5], c->ev[6]));     // plan
    HIPCHK(c, hipEventElapsedTime(&ms[4], c->ev[6], c->ev[7]));     // emit
    HIPCHK(c, hipEventElapsedTime(&ms[5], c->ev[0], c->ev[7]));     // first kernel start -> last kernel end
    return GC_OK;
}

extern "C" int gc_zstd_compress_host(gc_ctx* c, const void* src, size_t n, void* dst, size_t dstCap, int level, size_t* outSize)
{
    return gc_codec_compress_host(c, GC_CODEC_ZSTD, src, n, dst, dstCap, level, 0u, outSize);
}

// Optional in-kernel phase profile (s_memtime deltas of thread 0, averaged over blocks): K1 phases
// [probe, insert, verify, double, chain, walk, emit] then K3 phases [merge, codes, tables, chains, pack].
extern "C" int gc_zstd_set_phase_profile(gc_ctx* c, int enable) { if (!c) return GC_ERR_PARAM; c->profOn = enable != 0; return GC_OK; }
extern "C" int gc_zstd_phase_profile(gc_ctx* c, double cyclesPerBlock[GC_LZ_PHASES + GC_SEQ_PHASES])
{
    if (!c || !c->profOn || c->pending || !c->profBlocks) return GC_ERR_PARAM;
    unsigned long long h[GC_LZ_PHASES + GC_SEQ_PHASES];
    HIPCHK(c, hipMemcpy(h, c->prof, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < GC_LZ_PHASES + GC_SEQ_PHASES; i++) cyclesPerBlock[i] = (double)h[i] / c->profBlocks;
    return GC_OK;
}


// ------------------------------------------------------------------------------------------------ FLZMA2 (LZMA2 stream)
// level -> model segment size (gc_lzma2.h): smaller segments = more model waves in flight (faster), more state resets (larger).
static uint32_t flzma2_seg_log(int level)
{
    { uint32_t v = 0; if (gc_env_u32("GC_SEG_LOG", GC_LZMA_SEG_LOG_MIN, GC_LZMA_SEG_LOG_MAX, &v)) return v; }   // test hook
    if (level <= 1) return 14u;
    if (level == 2) return 17u;     // (run r03_fl2ab, silesia-like 32 MiB: 16 KiB segments 1.032 x the reference's level 2, 32 KiB 1.022, 128 KiB 1.012)
    if (level <= 4) return 15u;
    return 17u;                 // 128 KiB: what the reference's slices are (>= 112 KiB, lzma2_enc.h:22).  Measured (run r2_c, 64 MiB per corpus):
                                // 32 KiB -> 128 KiB segments is worth 0.9-1.5 % of the level-5 size
}

extern "C" size_t gc_flzma2_compress_bound(size_t n)
{
    const size_t nChunks = (n + GC_LZMA_RC_SIZE - 1) >> GC_LZMA_RC_LOG;
    return n + nChunks * 6u + 16u;
}

// level -> blocks per match-finder frame.  Every level runs the windowed finder over 8 MiB frames (round 3; levels 1-2 used the block-local
// finder before: a 128 KiB window against the reference's 1-2 MiB dictionaries, fl2_compress.c:52-63, was 12-24 % behind it, run r03_levels).
static uint32_t flzma2_frame_blocks(int level) { return level >= 7 ? GC_MF_WIDE_MAX_FRAME_BLOCKS : GC_MF_MAX_FRAME_BLOCKS; }     // (round 6: 16 MiB windows at 7-9 -- the reference: dictionaries of 32 / 64 / 64 MiB, fl2_compress.c:74-86)
// Levels 7-9 (the reference: dictionaries of 64 / 64 / 128 MiB, fl2_compress.c:59-62): overlapping finder frames (gc_mf.h) in groups of 64 MiB, stride 4 MiB at 7, 2 MiB at 8-9
// (a position is sure of 4 / 6 MiB of history; the window itself stays 8 MiB: 23-bit positions).  Levels 1-6: frames that tile the input.
// Levels 5-6 (round 5; the reference: 16 / 32 MiB dictionaries): groups of 16 MiB, stride 4 MiB -- real shared objects, 32 MiB at level 5 on the emulator: 1.0186 -> 1.0148 x the reference
static uint32_t flzma2_group_blocks(int level) { return level >= 7 ? 8u * GC_MF_MAX_FRAME_BLOCKS : (level >= 5 ? 2u * GC_MF_MAX_FRAME_BLOCKS : GC_MF_MAX_FRAME_BLOCKS); }
static uint32_t flzma2_stride_blocks(int level) { return level >= 8 ? GC_MF_MAX_FRAME_BLOCKS / 2u : (level >= 7 ? GC_MF_MAX_FRAME_BLOCKS : GC_MF_MAX_FRAME_BLOCKS / 2u); }      // 5-6: 4 MiB (8 MiB windows); 7: 8 MiB, 8-9: 4 MiB (16 MiB windows: 8-16 / 12-16 MiB of history)

// dictionary-size property byte of the 7z coder (Lzma2Encoder.cpp:353-364): dict = (2|(p&1)) << (p/2+11).
// Matches never reach back further than the start of their frame: 128 KiB (p = 10) or 8 MiB (p = 22).
extern "C" unsigned char gc_flzma2_dict_prop(int level) { const uint32_t f = flzma2_frame_blocks(level); return f == 1u ? 10 : (f > GC_MF_MAX_FRAME_BLOCKS ? 24 : 22); }      // 128 KiB / 8 MiB / 16 MiB

extern "C" int gc_flzma2_compress_device(gc_ctx* c, const void* d_src, size_t n, void* d_dst, size_t dstCap, int level, unsigned flags)
{
    if (!c || (!d_src && n) || !d_dst) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    c->timed = false;
    if (n == 0) {
        const size_t sz = (flags & GC_FLZMA2_NO_END_MARK) ? 0 : 1;
        if (dstCap < sz) return GC_ERR_DST_SMALL;
        static const uint8_t endMark[1] = { 0x00 };
        if (sz) HIPCHK(c, hipMemcpyAsync(d_dst, endMark, 1, hipMemcpyHostToDevice, c->stream));
        c->hostResult[0] = sz; c->hostResult[1] = 0;
        HIPCHK(c, hipMemcpyAsync(c->result, c->hostResult, 16, hipMemcpyHostToDevice, c->stream));
        c->pending = true;
        return GC_OK;
    }
    const uint32_t nBlocks = gc_num_blocks(n);
    int rc = ensure_workspace(c, nBlocks);
    if (rc != GC_OK) return rc;
    const uint32_t segLog = flzma2_seg_log(level);
    const uint32_t nSegs = nBlocks * (GC_ZSTD_BLOCK_MAX >> segLog), nRc = nBlocks * GC_LZMA_RC_PER_BLOCK;
    const uint8_t* src = (const uint8_t*)d_src;
    {   // (p, bit) stream between the model kernel and the range coder: 2 bytes per coded bit, <= 9 coded bits per input byte;
        // item lists; range-coder staging
        const size_t needStream = (size_t)nSegs * GC_LZMA_STREAM_WORDS(segLog) * sizeof(uint16_t);
        const size_t needM = (size_t)nBlocks * GC_LZMA_MAX_ITEMS * sizeof(uint64_t), needRc = (size_t)nRc * GC_LZMA_RC_STRIDE;
        if (needStream > c->lzStreamCap || needM > c->lzMCap || needRc > c->lzRcOutCap) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if ((rc = mf_grow(c, (void**)&c->lzStream, &c->lzStreamCap, needStream, "LZMA stream")) != GC_OK) return rc;
            if ((rc = mf_grow(c, (void**)&c->lzM, &c->lzMCap, needM, "LZMA items")) != GC_OK) return rc;
            if ((rc = mf_grow(c, (void**)&c->lzRcOut, &c->lzRcOutCap, needRc, "range-coder staging")) != GC_OK) return rc;
        }
    }
    uint32_t frameBlocks = flzma2_frame_blocks(level);
    c->lazyDepth = level >= 5 ? 2u : 1u;
    c->mfFast = 0;
    c->searchShallow = level >= 5 ? 2u : 0u;
    c->searchDepth = level >= 5 ? (level >= 8 ? 16u : 12u) : 0u;    // links followed where a tile has long matches, by the positions that start one (two links elsewhere: gc_mf_deepen_kernel).
                                                                    // Real source text (64 MiB): two links everywhere 1.030 x the reference, six everywhere 1.017 (run r03_depth)
    c->ringParse = 0u;
    c->farPass = level >= 3 ? 1u : 0u;            // the reference's match table resolves to depth 42 at level 5 (fl2_compress.c:37-104);
                                                  // level 3 (run 30x, 32 MiB): 1.071 -> 1.026 x the reference on text
    c->shortPass = level >= 3 ? 1u : 0u;          // ... and holds the nearest match of >= 2 bytes for every position
    gc_env_u32("GC_FAR_PASS", 0u, 1u, &c->farPass); gc_env_u32("GC_SHORT_PASS", 0u, 1u, &c->shortPass);   // test hooks
    if (gc_env_u32("GC_SEARCH_DEPTH", 0u, 64u, &c->searchDepth)) c->searchShallow = c->searchDepth < 2u ? c->searchDepth : 2u;
    c->shortPlain = level < 7 ? 1u : 0u; c->allLengths = 0u;
    c->smallWin2k = (level == 5 || level == 6) ? 1u : 0u;
    c->farPass2 = level >= 7 ? 1u : 0u; gc_env_u32("GC_FAR2_PASS", 0u, 1u, &c->farPass2);      // keys of 32 / 24 bytes (gc_lz_window.hip MF_FAR2) at the ultra levels
    c->laneParse = 1u; c->lastCodecHint = 1; c->priceMinLen = 2u; c->priceLitCtx = 7u;
    c->priceParse = level >= 3 ? 1u : 0u;         // the reference's FL2_opt strategy starts at level 3 of its 7-Zip table (fl2_compress.c:52-63); round 3 (run r03_fl2ab): level 3 with
                                                  // the greedy parse was 1.038 x the reference on silesia-like, with the price-based parse 1.002
    gc_env_u32("GC_PRICE_PARSE", 0u, 1u, &c->priceParse);                                      // test hook: 0 = greedy parse only
    if (frameBlocks > 1u && c->dbgFrameBlocks) frameBlocks = c->dbgFrameBlocks;
    if (frameBlocks > nBlocks) frameBlocks = nBlocks;
    uint32_t fArg = frameBlocks;                                                               // what the finder takes: overlapping frames from level 7 (gc_mf.h)
    { uint32_t grp = flzma2_group_blocks(level); const bool grpHook = gc_env_u32("GC_MF_GROUP", 1u, 65535u, &grp);      // test hook: blocks per group (with GC_FRAME_BLOCKS and GC_MF_STRIDE: overlapping frames of a few blocks)
      if (grp > frameBlocks && (frameBlocks == flzma2_frame_blocks(level) || grpHook) && nBlocks > frameBlocks) {
          uint32_t stride = flzma2_stride_blocks(level); gc_env_u32("GC_MF_STRIDE", 1u, GC_MF_MAX_FRAME_BLOCKS, &stride);      // test hook (blocks; 64 = no overlap)
          if (stride < frameBlocks && (frameBlocks % stride) == 0u && ((grp - frameBlocks) % stride) == 0u) fArg = GC_MF_GEOM_ARG(frameBlocks, stride, grp);
      } }
    const uint32_t groupBlocks = MF_C(fArg);
    // parts: ONE by default.  Overlapping the stages of several parts was measured and lost (212 MB: 33.9 ms with 4 parts against
    // 24.3 ms with one, profiles/r01_run8_flzma2_kernel_stats.md): model and range coder are chains whose duration is set by the
    // length of one segment / chunk, not by how many there are, so every part pays the full chain again, and the model kernel's
    // LDS footprint keeps the finder of the next part waiting.  GC_PART_FRAMES (test hook) still selects parts of that many frames.
    const uint32_t nFrames = (nBlocks + groupBlocks - 1u) / groupBlocks;                       // (parts are whole groups; without overlap a group is a frame)
    uint32_t nParts = c->dbgPartFrames ? nFrames / c->dbgPartFrames : 1u; if (nParts < 1u) nParts = 1u;
    { const uint32_t autoParts = mf_auto_parts(n, fArg); if (nParts < autoParts) nParts = autoParts; }      // (entry lists beyond the budget: ensure_finder_workspace)
    if (nParts > GC_MAX_PARTS) nParts = GC_MAX_PARTS;
    if (nParts > nFrames) nParts = nFrames;
    if (frameBlocks <= 1u) nParts = 1u;
    {   // the largest part (parts are whole groups: the split below)
        uint32_t maxGroups = 0, g0 = 0;
        for (uint32_t p = 0; p < nParts; p++) { const uint32_t g1 = (uint32_t)(((uint64_t)nFrames * (p + 1u)) / nParts); if (g1 - g0 > maxGroups) maxGroups = g1 - g0; g0 = g1; }
        rc = ensure_finder_workspace(c, n, fArg, (size_t)maxGroups * groupBlocks * GC_ZSTD_BLOCK_MAX);
        if (rc != GC_OK) return rc;
    }
    c->mfTimed = false; c->nParts = nParts;
    const uint32_t segPerBlock = GC_ZSTD_BLOCK_MAX >> segLog;
    uint32_t mergeWords = level <= 6 ? 32768u : GC_LZMA_RC_MERGE_WORDS;                        // coded bits of one LZMA2 chunk = the chain of ONE lane of L3.  Levels <= 6 (run s9, 211.9 MB): 49 152 -> 32 768 words takes 1.65 ms off L3
                                                                                               // (5.98 -> 4.33) for +0.06 % size (10 bytes of header and coder flush per chunk)
    gc_env_u32("GC_RC_MERGE_WORDS", 0u, GC_LZMA_RC_MERGE_WORDS, &mergeWords);                  // test hook: 0 = one LZMA2 chunk per rc chunk
    uint32_t rep4 = 1;                                                                         // rep2 / rep3 coding in L2 (gc_lzma2_enc.hip LzLru); test hook: 0 = rep0 / rep1 only
    gc_env_u32("GC_L2_REP4", 0u, 1u, &rep4);
    uint32_t litSel = 1;                                                                       // lc / lp per model segment (gc_lzma2_model_kernel); test hook: 0 = the reference's lc 3 / lp 0 everywhere
    gc_env_u32("GC_L2_LITSEL", 0u, 1u, &litSel);
    uint32_t wordCap = GC_LZMA_STREAM_WORDS(segLog);                                           // words a segment may produce before it is stored instead
    gc_env_u32("GC_SEG_WORD_CAP", 1u, GC_LZMA_STREAM_WORDS(segLog), &wordCap);                  // test hook: a low cap sends ordinary segments down that path
    // model segments over 2 / 4 / 8 blocks where the parse prices them within what one block of poorly compressible data costs (gc_lzma2_model_kernel): level >= 5.
    // Parts (test hook) are whole frames, frames are multiples of eight blocks: a group never straddles a part.
    uint32_t segMerge = (segLog == 17u && level >= 5 && rep4) ? 8u : 0u;
    gc_env_u32("GC_SEG_MERGE", 0u, 8u, &segMerge);                                              // test hook: 0 = every block a segment of its own (round 4)
    if (segMerge & (segMerge - 1u)) segMerge = 4u;
    while (segMerge > 1u && nBlocks > groupBlocks && (groupBlocks % segMerge) != 0u) segMerge >>= 1;      // (groups are aligned to the input: they must not straddle frames, or a frame-aligned shard would differ from the whole input's bytes)
    uint32_t mergeBudget = 16u * 655360u;                                                      // 640 Ki coded bits (1/16 bit units) = 5 bits per byte of ONE 128 KiB block.  Measured on 211.9 MB (run s9, model kernel / size):
                                                                                               // 896 Ki -- the longest chain the launch has anyway, PCM-like data -- 17.0 ms on the Silesia stand-in and 21.5 ms on shared objects
                                                                                               // (the estimate is the PARSE's: the model's chain comes out longer); 640 Ki 14.4 / 14.5 ms, as without merging, for +0.01 / +0.05 % size
    { uint32_t kbits = 0; if (gc_env_u32("GC_SEG_MERGE_KBITS", 1u, 4096u, &kbits)) mergeBudget = kbits * 16384u; }
    uint8_t* const segKind = c->lzProps + (size_t)c->capBlocks * (GC_ZSTD_BLOCK_MAX >> GC_LZMA_SEG_LOG_MIN);
    if (c->profOn) { HIPCHK(c, hipMemsetAsync(c->prof, 0, (GC_LZ_PHASES + GC_SEQ_PHASES) * sizeof(unsigned long long), c->stream)); c->profBlocks = 1u; }   // L2 phase sums (raw)
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    uint32_t f0 = 0;
    for (uint32_t p = 0; p < nParts; p++) {
        const uint32_t f1 = (uint32_t)(((uint64_t)nFrames * (p + 1u)) / nParts);
        const uint32_t blk0 = f0 * groupBlocks, blk1 = f1 * groupBlocks < nBlocks ? f1 * groupBlocks : nBlocks;
        const uint64_t off = (uint64_t)blk0 * GC_ZSTD_BLOCK_MAX;
        const size_t pn = (size_t)(((uint64_t)blk1 * GC_ZSTD_BLOCK_MAX < n ? (uint64_t)blk1 * GC_ZSTD_BLOCK_MAX : (uint64_t)n) - off);
        const uint32_t pBlocks = blk1 - blk0, pSegs = pBlocks * segPerBlock, pRc = pBlocks * GC_LZMA_RC_PER_BLOCK;
        hipEvent_t* ev = c->evPart[p];
        // stage 1 (main stream): match finder + item lists
        HIPCHK(c, hipEventRecord(ev[0], c->stream));
        rc = launch_finder_part(c, c->stream, p, src + off, pn, fArg, blk0, nullptr);
        if (rc != GC_OK) return rc;
        HIPCHK(c, hipEventRecord(ev[1], c->stream));
        GC_LAUNCH(gc_lzma2_prep_kernel, pBlocks, 256, c->stream, (const GcSeqRaw*)(c->seqRaw + (size_t)blk0 * GC_MAX_SEQ_PER_BLOCK),
                  (const GcBlockMeta*)(c->meta + blk0), (uint64_t)pn, c->lzM + (size_t)blk0 * GC_LZMA_MAX_ITEMS, c->lzNM + blk0);
        HIPCHK(c, hipEventRecord(ev[2], c->stream));
        // stage 2 (stream2): model
        HIPCHK(c, hipStreamWaitEvent(c->stream2, ev[2], 0));
        HIPCHK(c, hipEventRecord(ev[3], c->stream2));
        GC_LAUNCH(gc_lzma2_model_kernel, pSegs, 64, c->stream2, src + off, (uint64_t)pn, (const uint64_t*)(c->lzM + (size_t)blk0 * GC_LZMA_MAX_ITEMS),
                  (const uint32_t*)(c->lzNM + blk0), segLog, (uint32_t)(off != 0u ? 1u : 0u),
                  c->lzStream + (size_t)blk0 * segPerBlock * GC_LZMA_STREAM_WORDS(segLog), c->lzInfo + (size_t)blk0 * GC_LZMA_RC_PER_BLOCK,
                  (const uint32_t*)((c->priceParse && frameBlocks > 1u) ? c->mfWinCost + (size_t)blk0 * 32u : nullptr),
                  c->profOn ? c->prof : nullptr, mergeWords, wordCap, rep4, c->lzProps + (size_t)blk0 * segPerBlock, litSel, segMerge, mergeBudget);
        HIPCHK(c, hipEventRecord(ev[4], c->stream2));
        // stage 3 (stream3): range coder
        HIPCHK(c, hipStreamWaitEvent(c->stream3, ev[4], 0));
        HIPCHK(c, hipEventRecord(ev[5], c->stream3));
        GC_LAUNCH(gc_lzma2_rc_kernel, (pRc + 63u) / 64u, 64, c->stream3, c->lzStream + (size_t)blk0 * segPerBlock * GC_LZMA_STREAM_WORDS(segLog),
                  segLog, pRc, c->lzRcOut + (size_t)blk0 * GC_LZMA_RC_PER_BLOCK * GC_LZMA_RC_STRIDE, c->lzInfo + (size_t)blk0 * GC_LZMA_RC_PER_BLOCK);
        GC_LAUNCH(gc_lzma2_rc_fin_kernel, pRc, 64, c->stream3, (const uint16_t*)(c->lzStream + (size_t)blk0 * segPerBlock * GC_LZMA_STREAM_WORDS(segLog)),
                  segLog, pRc, c->lzRcOut + (size_t)blk0 * GC_LZMA_RC_PER_BLOCK * GC_LZMA_RC_STRIDE, c->lzInfo + (size_t)blk0 * GC_LZMA_RC_PER_BLOCK);
        HIPCHK(c, hipEventRecord(ev[6], c->stream3));
        f0 = f1;
    }
    c->mfTimed = frameBlocks > 1u; c->mfParts = nParts; c->mfPriced = c->mfTimed && c->priceParse != 0u;
    // all parts coded -> headers and assembly on the main stream
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->evPart[nParts - 1u][6], 0));
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    GC_LAUNCH(gc_lzma2_segkind_kernel, (nSegs + 255u) / 256u, 256, c->stream, (const GcLzmaChunkInfo*)c->lzInfo, (const uint8_t*)c->lzProps, nSegs, segLog, segKind);
    GC_LAUNCH(gc_lzma2_plan_kernel, 1, 1024, c->stream, (const GcLzmaChunkInfo*)c->lzInfo, nRc, segLog, (uint64_t)dstCap, (uint32_t)flags, c->lzPlan, c->result,
              (const uint8_t*)c->lzProps, (const uint8_t*)segKind);
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    GC_LAUNCH(gc_lzma2_emit_kernel, nRc + 1u, 256, c->stream, src, segLog, (const uint8_t*)c->lzRcOut, (const GcLzmaChunkInfo*)c->lzInfo,
              (const GcLzmaPlan*)c->lzPlan, nRc, (uint32_t)flags, (const uint64_t*)c->result, (uint8_t*)d_dst, (const uint8_t*)c->lzProps);
    HIPCHK(c, hipEventRecord(c->ev[5], c->stream));
    HIPCHK(c, hipGetLastError());
    c->pending = true; c->timed = true; c->lastCodec = 1;
    return GC_OK;
}

extern "C" int gc_flzma2_finish(gc_ctx* c, size_t* compressedSize) { return gc_zstd_finish(c, compressedSize); }

// Stage times are summed over the parts of the input; stages of different parts overlap (three streams), so their sum exceeds
// ms[6], the time from the first kernel's start to the last kernel's end.
extern "C" int gc_flzma2_last_timing(gc_ctx* c, float ms[7])
{
    if (!c || !c->timed || c->pending || c->lastCodec != 1) return GC_ERR_PARAM;
    for (int i = 0; i < 4; i++) ms[i] = 0.f;
    for (uint32_t p = 0; p < c->nParts; p++) {
        static const int a[4] = { 0, 1, 3, 5 }, b[4] = { 1, 2, 4, 6 };      // lz, prep, model, rc
        for (int i = 0; i < 4; i++) { float t = 0.f; HIPCHK(c, hipEventElapsedTime(&t, c->evPart[p][a[i]], c->evPart[p][b[i]])); ms[i] += t; }
    }
    HIPCHK(c, hipEventElapsedTime(&ms[4], c->ev[3], c->ev[4]));     // plan
    HIPCHK(c, hipEventElapsedTime(&ms[5], c->ev[4], c->ev[5]));     // emit
    HIPCHK(c, hipEventElapsedTime(&ms[6], c->ev[0], c->ev[5]));     // first kernel start -> last kernel end
    return GC_OK;
}

extern "C" int gc_flzma2_compress_host(gc_ctx* c, const void* src, size_t n, void* dst, size_t dstCap, int level, unsigned flags, size_t* outSize)
{
    return gc_codec_compress_host(c, GC_CODEC_FLZMA2, src, n, dst, dstCap, level, flags, outSize);
}


// ------------------------------------------------------------------------------------------------ BROTLI (brotli-mt framed)
// chunk = 1 MiB x level as in brotli-mt (C/zstdmt/brotli-mt_compress.c:115-118), in 128 KiB blocks
static uint32_t brotli_blocks_per_chunk(int level) { if (level < 1) level = 1; if (level > 11) level = 11; return (uint32_t)level * 8u; }

// quality -> blocks per match-finder frame.  Quality 0: block-local finder.  Above (from quality 1 since round 3: 1.04 x the reference before): the windowed finder over frames that
// tile the chunk exactly (a copy must not reach into the previous chunk: every chunk is a brotli stream of its own), the whole
// chunk when it is <= 8 MiB (qualities 3-8), half of it above (72/80/88 blocks -> 36/40/44).
static uint32_t brotli_frame_blocks(int level, uint32_t bpc)
{
    if (level <= 0) return 1u;
    const uint32_t cap = level >= 7 ? GC_MF_WIDE_MAX_FRAME_BLOCKS : GC_MF_MAX_FRAME_BLOCKS;       // (qualities >= 7 run the wide geometry: 16 MiB of positions, so the 9 / 10 / 11 MiB chunks of qualities 9-11 are ONE frame since round 6)
    return bpc <= cap ? bpc : bpc / 2u;
}

extern "C" size_t gc_brotli_compress_bound(size_t n)
{
    const size_t nb = n ? (n + GC_ZSTD_BLOCK_MAX - 1) / GC_ZSTD_BLOCK_MAX : 1;
    return n + nb * 8u + (nb / 8u + 1u) * 17u + 32u;
}

extern "C" int gc_brotli_compress_device(gc_ctx* c, const void* d_src, size_t n, void* d_dst, size_t dstCap, int level)
{
    if (!c || (!d_src && n) || !d_dst) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    c->timed = false;
    if (n == 0) {
        // what brotli-mt writes for an empty input: one frame holding the 1-byte empty brotli stream (WBITS=16, ISLAST, ISLASTEMPTY)
        static const uint8_t empty[17] = { 0x50, 0x2A, 0x4D, 0x18, 8, 0, 0, 0, 1, 0, 0, 0, 0x42, 0x52, 1, 0, 0x06 };
        const size_t skip = c->optBrotliPlain ? 16u : 0u, sz = (c->optBrotliPlain & 6u) ? 0u : sizeof(empty) - skip;       // plain stream: just the empty brotli stream (nothing for an inner piece)
        if (dstCap < sz) return GC_ERR_DST_SMALL;
        HIPCHK(c, hipMemcpyAsync(d_dst, empty + skip, sz, hipMemcpyHostToDevice, c->stream));
        c->hostResult[0] = sz; c->hostResult[1] = 0;
        HIPCHK(c, hipMemcpyAsync(c->result, c->hostResult, 16, hipMemcpyHostToDevice, c->stream));
        c->pending = true;
        return GC_OK;
    }
    const uint32_t nBlocks = gc_num_blocks(n);
    int rc = ensure_workspace(c, nBlocks);
    if (rc != GC_OK) return rc;
    const uint32_t bpcFinder = brotli_blocks_per_chunk(level);
    const uint32_t bpc = c->optBrotliPlain ? 0xFFFFFFFFu : bpcFinder;        // plain: one stream (B1 writes the stream header once, B2 / B3 no frame headers)
    const uint8_t* src = (const uint8_t*)d_src;
    HIPCHK(c, hipMemsetAsync(c->brStage, 0, (size_t)nBlocks * GC_BR_STAGE_STRIDE, c->stream));
    uint32_t frameBlocks = brotli_frame_blocks(level, bpcFinder);
    c->lazyDepth = level >= 5 ? 2u : 1u; gc_env_u32("GC_BR_LAZY", 0u, 2u, &c->lazyDepth);        // W6 looks two positions ahead from quality 5 (round 6, emulator, 4 MiB at quality 6: web-text 0.960 -> 0.950 x the reference;
                                                                                                  // it was 7).  W6r keeps one position (two: shared objects 1.035 -> 1.039).  Test hook
    c->mfFast = level <= 6 ? 1u : 0u;        // (qualities 5-6 run the far pass on the fast geometry: 0.97-0.99 x the reference at 15 % less time than on the wide one)
    // W5b from quality 7: four links (eight from quality 10) for the starts of matches in tiles with long matches, two elsewhere.  Quality 5 stays without it (round 3 measured
    // quality 6 with (8, 0) (run r03_q3): sources 1.084 -> 1.068 x the reference and the Python library 1.024 -> 1.015, but web-text -- config C5's data, whose boilerplate
    // makes most tiles "long" -- 16.6 -> 9.4 GB/s for 0.3 % of its size
    // Quality 6 follows ONE link everywhere since round 4 (run r4brd / r4brd3, 64 MiB per corpus, web-text 500 MB): real Python library 1.024 -> 1.015 x the reference (inside the
    // band), real sources 1.084 -> 1.069, shared objects 1.106 -> 1.101, web-text 17.0 -> 14.5 GB/s (W5b 5.0 ms per 500 MB).  One link for the starts of matches only:
    // 1.020 / 1.076 / 1.104 at 15.0 GB/s; two links: 1.010 / 1.062 / 1.099 at 12.9; four (starts only beyond two): 1.009 / 1.059 / 1.098 at 11.4.  Quality 5 stays without.
    c->searchDepth = level >= 7 ? (level >= 10 ? 8u : 4u) : (level == 6 ? 1u : 0u); c->searchShallow = c->searchDepth < 2u ? c->searchDepth : 2u;
    if (gc_env_u32("GC_SEARCH_DEPTH", 0u, 64u, &c->searchDepth)) c->searchShallow = c->searchDepth < 2u ? c->searchDepth : 2u;      // test hook
    gc_env_u32("GC_SEARCH_SHALLOW", 0u, 64u, &c->searchShallow);                                // test hook: links followed by the positions inside a match and in tiles without long matches
    c->farPass = level >= 5 ? 1u : 0u; c->shortPass = 0;      // longer matches stand in for the context modelling / block splitting B1 lacks
    gc_env_u32("GC_FAR_PASS", 0u, 1u, &c->farPass);                                            // test hook
    c->shortPlain = 0u; c->smallWin2k = 0u; c->allLengths = 0u;
    c->farPass2 = 0u; gc_env_u32("GC_FAR2_PASS", 0u, 1u, &c->farPass2);
    c->ringParse = level >= 5 ? (2u | (8u << 8) | (4u << 16) | (16u << 24)) : 0u; gc_env_u32("GC_BR_RING", 0u, 0xFFFFFFFFu, &c->ringParse);
    c->ringGeom = 256u; { uint32_t g = 0; if (gc_env_u32("GC_BR_RING_GEOM", 64u, 256u, &g) && (g & 63u) == 0u) c->ringGeom = g; }     // test hook: 64 / 128 / 256 threads = 4 / 8 / 16 sub-blocks
    c->laneParse = level >= 7 ? 1u : 0u; gc_env_u32("GC_BR_LANE", 0u, 1u, &c->laneParse);      // qualities 7-11: W7L with the ring's first entries as its repeat distances, per block where phase A's paths repeat (launch_finder_part)
    c->lastCodecHint = 2; c->priceMinLen = 3u; c->priceLitCtx = 0u;     // copies of >= 3 bytes (a 2-byte copy at a fresh distance never pays in brotli), one literal code per meta-block
    c->priceParse = level >= 7 ? 1u : 0u;         // (quality 7 since round 6 -- emulator, 2 MiB, with W7L: real sources 1.018 -> 0.980 x the reference, shared objects 1.040 -> 1.026, text 0.963 -> 0.939.)
                                                  // The reference parses greedily up to quality 9 (zopfli from 10).  Measured at quality 6
                                                  // (run 28, 64 MiB per corpus): greedy + far pass 0.979-1.002 x the reference at 16.6 GB/s,
                                                  // price-based parse without far pass 0.983-1.012 x at 11.1 GB/s, both 0.93-0.98 x at 9.4 GB/s
    gc_env_u32("GC_PRICE_PARSE", 0u, 1u, &c->priceParse);                                      // test hook
    if (frameBlocks > nBlocks) frameBlocks = nBlocks;                       // short input: one chunk, one frame
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    rc = launch_finder(c, src, n, frameBlocks, nullptr);
    if (rc != GC_OK) return rc;
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    uint32_t brRepSub = 2u;                                    // B1's last-distance substitution (gc_brotli.hip): two passes.  Emulator, quality 6, x the reference, passes 0 / 1 / 2 / 3 / 4: shared objects
                                                               // 8 MiB 1.1262 / 1.1095 / 1.0969 / 1.0919 / 1.0888 (2 MiB: 1.0925 / 1.0788 / 1.0771), sources 8 MiB 1.1594 / 1.1439 / 1.1392; a pass carries a
                                                               // distance three commands further along a run of records and costs 0.86 ms per 500 MB on the device (B1 5.4 ms without)
    gc_env_u32("GC_BR_REPSUB", 0u, 64u, &brRepSub);                                             // test hook
    uint32_t brCtx = level >= 5 ? 1u : 0u;                     // literal context modelling from quality 5 (the reference: MIN_QUALITY_FOR_CONTEXT_MODELING, C/brotli/enc/quality.h): B1 chooses one tree or thirteen per meta-block
    gc_env_u32("GC_BR_CTX", 0u, 1u, &brCtx);                                                    // test hook
    GC_LAUNCH(gc_brotli_block_kernel, nBlocks, 256, c->stream, src, (uint64_t)n, (const GcSeqRaw*)c->seqRaw, (const uint8_t*)c->lit,
              (const GcBlockMeta*)c->meta, c->seqPacked, c->seqOff, bpc, c->optBrotliPlain, brRepSub, brCtx, (uint32_t*)c->brStage, c->brInfo);
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
    GC_LAUNCH(gc_brotli_plan_kernel, 1, 1024, c->stream, (const GcBrotliBlockInfo*)c->brInfo, nBlocks, bpc, (uint64_t)dstCap, c->brPlan, c->result);
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    GC_LAUNCH(gc_brotli_emit_kernel, nBlocks, 256, c->stream, src, (uint64_t)n, (const uint8_t*)c->brStage, (const GcBrotliBlockInfo*)c->brInfo,
              (const GcBrotliPlan*)c->brPlan, nBlocks, bpc, c->optBrotliPlain, (const uint64_t*)c->result, (uint8_t*)d_dst);
    HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    HIPCHK(c, hipGetLastError());
    c->pending = true; c->timed = true; c->lastCodec = 2;
    return GC_OK;
}

extern "C" int gc_brotli_finish(gc_ctx* c, size_t* compressedSize) { return gc_zstd_finish(c, compressedSize); }

extern "C" int gc_brotli_last_timing(gc_ctx* c, float ms[5])
{
    if (!c || !c->timed || c->pending || c->lastCodec != 2) return GC_ERR_PARAM;
    for (int i = 0; i < 4; i++) HIPCHK(c, hipEventElapsedTime(&ms[i], c->ev[i], c->ev[i + 1]));   // lz, block, plan, emit
    HIPCHK(c, hipEventElapsedTime(&ms[4], c->ev[0], c->ev[4]));
    return GC_OK;
}

extern "C" int gc_brotli_compress_host(gc_ctx* c, const void* src, size_t n, void* dst, size_t dstCap, int level, size_t* outSize)
{
    return gc_codec_compress_host(c, GC_CODEC_BROTLI, src, n, dst, dstCap, level, 0u, outSize);
}

// ------------------------------------------------------------------------------------------------ host-buffer building blocks
// What the three CEncoder::Code loops need (ZstdEncoder.cpp:398-461, Lzma2Encoder.cpp:260-350, BrotliEncoder.cpp:118-164), split so
// that a host scheduler (gc_multi.hip) can keep the H2D copy of one piece, the kernels of another and the D2H copy of a third in
// flight: begin = stage the bytes into the context's device buffer + enqueue the codec (asynchronous for pinned `src`),
// size = wait for the compressed size, fetch = copy the compressed bytes to their final place.
extern "C" size_t gc_codec_compress_bound(int codec, size_t n)
{
    return codec == GC_CODEC_ZSTD ? gc_zstd_compress_bound(n) : (codec == GC_CODEC_FLZMA2 ? gc_flzma2_compress_bound(n) : gc_brotli_compress_bound(n));
}

struct GcStreamScope { hipStream_t prev; explicit GcStreamScope(hipStream_t st) : prev(gc_tls_stream) { gc_tls_stream = st; } ~GcStreamScope() { gc_tls_stream = prev; } };      // (gc_host_stream.h: the stand-alone entry points run on this context's stream while in scope)
extern "C" int gc_crc32_device(const void* d_src, size_t n, uint32_t* crc);
extern "C" int gc_bra_convert_device(int kind, const void* d_src, void* d_dst, size_t n, uint32_t pc, int encoding, size_t* processed);
extern "C" int gc_bra_x86_convert_device(const void* d_src, void* d_dst, size_t n, uint32_t pc, int encoding, uint32_t* state, size_t* processed);
extern "C" int gc_delta_convert_device(const void* d_src, void* d_dst, size_t n, unsigned delta, int encoding, unsigned char state[256]);

// H2D + [CRC-32 of the raw bytes + pre-filter, both on the device: gpucodec.h gc_pre] + enqueue of the codec's kernels
extern "C" int gc_host_begin_pre(gc_ctx* c, int codec, const void* src, size_t n, int level, unsigned flags, gc_pre* pre)
{
    if (!c || (!src && n) || codec < GC_CODEC_ZSTD || codec > GC_CODEC_BROTLI) return GC_ERR_PARAM;
    const int flt = pre ? pre->filter : 0;
    const bool x86 = flt == GC_FILTER_X86, dl = flt == GC_FILTER_DELTA;
    if (flt != 0 && !x86 && !dl && (flt < GC_BRA_ARM64 || flt > GC_BRA_RISCV)) return GC_ERR_PARAM;
    if (dl && (pre->delta < 1u || pre->delta > 256u)) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t bound = gc_codec_compress_bound(codec, n);
    if (n > c->dInCap) { hipFree(c->dIn); c->dIn = nullptr; c->dInCap = 0; if (hipMalloc((void**)&c->dIn, n + 64) != hipSuccess) return GC_ERR_NOMEM; c->dInCap = n; }
    if (bound > c->dOutCap) { hipFree(c->dOut); c->dOut = nullptr; c->dOutCap = 0; if (hipMalloc((void**)&c->dOut, bound) != hipSuccess) return GC_ERR_NOMEM; c->dOutCap = bound; }
    if (n) HIPCHK(c, hipMemcpyAsync(c->dIn, src, n, hipMemcpyHostToDevice, c->stream));
    const uint8_t* d_in = c->dIn;
    if (pre) { pre->crc = 0; pre->processed = 0; }
    if (pre && n && (pre->want_crc || flt)) {
        const GcStreamScope onMine(c->stream);                       // the CRC and the converters run on this context's stream (round 6: they used the null stream and waited for the whole device), in order behind the copy above
        if (pre->want_crc) { const int rc = gc_crc32_device(c->dIn, n, &pre->crc); if (rc != GC_OK) { snprintf(c->err, sizeof(c->err), "CRC of the input failed on the device"); return rc; } }
        if (flt) {
            if (n > c->dPreCap) { hipFree(c->dPre); c->dPre = nullptr; c->dPreCap = 0; if (hipMalloc((void**)&c->dPre, n + 64) != hipSuccess) return GC_ERR_NOMEM; c->dPreCap = n; }
            size_t done = 0; int rc;
            if (x86) { uint32_t st; memcpy(&st, pre->state, 4); rc = gc_bra_x86_convert_device(c->dIn, c->dPre, n, pre->pc, 1, &st, &done); memcpy(pre->state, &st, 4); }
            else if (dl) { rc = gc_delta_convert_device(c->dIn, c->dPre, n, pre->delta, 1, pre->state); done = n; }
            else rc = gc_bra_convert_device(flt, c->dIn, c->dPre, n, pre->pc, 1, &done);
            if (rc != GC_OK) { snprintf(c->err, sizeof(c->err), "pre-filter %d failed on the device", flt); return rc; }
            pre->processed = done;
            d_in = c->dPre;
        }
    } else if (pre && pre->want_crc) pre->crc = 0u;                  // (CrcCalc of nothing)
    if (codec == GC_CODEC_BROTLI && (flags & GC_BROTLI_PLAIN)) {      // per-call form of GC_OPT_BROTLI_PLAIN (pieces of one stream); the context's own option comes back afterwards
        const uint32_t keep = c->optBrotliPlain;
        c->optBrotliPlain = (flags & 7u) | 1u;
        const int rc = gc_brotli_compress_device(c, d_in, n, c->dOut, c->dOutCap, level);
        c->optBrotliPlain = keep;
        return rc;
    }
    return codec == GC_CODEC_ZSTD ? gc_zstd_compress_device(c, d_in, n, c->dOut, c->dOutCap, level)
         : codec == GC_CODEC_FLZMA2 ? gc_flzma2_compress_device(c, d_in, n, c->dOut, c->dOutCap, level, flags)
                                    : gc_brotli_compress_device(c, d_in, n, c->dOut, c->dOutCap, level);
}
extern "C" int gc_host_begin(gc_ctx* c, int codec, const void* src, size_t n, int level, unsigned flags) { return gc_host_begin_pre(c, codec, src, n, level, flags, nullptr); }

extern "C" int gc_host_size(gc_ctx* c, size_t* compressedSize) { return gc_zstd_finish(c, compressedSize); }

extern "C" int gc_host_fetch(gc_ctx* c, void* dst, size_t size)
{
    if (!c || (!dst && size) || size > c->dOutCap) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    if (size) { HIPCHK(c, hipMemcpyAsync(dst, c->dOut, size, hipMemcpyDeviceToHost, c->stream)); HIPCHK(c, hipStreamSynchronize(c->stream)); }
    return GC_OK;
}

extern "C" int gc_codec_compress_host_pre(gc_ctx* c, int codec, const void* src, size_t n, void* dst, size_t dstCap, int level, unsigned flags, gc_pre* pre, size_t* outSize)
{
    if (!c || (!src && n) || !dst) return GC_ERR_PARAM;
    int rc = gc_host_begin_pre(c, codec, src, n, level, flags, pre);
    if (rc != GC_OK) return rc;
    size_t sz = 0;
    rc = gc_host_size(c, &sz);
    if (rc != GC_OK) return rc;
    if (sz > dstCap) { snprintf(c->err, sizeof(c->err), "destination too small: need %zu bytes", sz); return GC_ERR_DST_SMALL; }
    rc = gc_host_fetch(c, dst, sz);
    if (rc != GC_OK) return rc;
    if (outSize) *outSize = sz;
    return GC_OK;
}
extern "C" int gc_codec_compress_host(gc_ctx* c, int codec, const void* src, size_t n, void* dst, size_t dstCap, int level, unsigned flags, size_t* outSize)
{
    return gc_codec_compress_host_pre(c, codec, src, n, dst, dstCap, level, flags, nullptr, outSize);
}

// pinned host memory for callers that want the copies of gc_host_begin / gc_host_fetch to run at link speed and asynchronously
extern "C" void* gc_host_alloc(size_t n) { void* p = nullptr; return hipHostMalloc(&p, n ? n : 1) == hipSuccess ? p : nullptr; }
extern "C" void gc_host_free(void* p) { if (p) hipHostFree(p); }

// Independence grain of a codec at a level: a range of the input that starts at a multiple of it is compressed to exactly the
// bytes it has inside a whole-buffer call (zstd: frames; brotli: brotli-mt chunks) or to a run of LZMA2 chunks that starts with a
// dictionary reset (FLZMA2: match-finder frames).  Host schedulers split the input at multiples of it.
extern "C" size_t gc_codec_grain(int codec, int level)
{
    uint32_t fb = codec == GC_CODEC_ZSTD ? zstd_frame_blocks(level) : flzma2_frame_blocks(level);
    if (codec != GC_CODEC_BROTLI && fb > 1u) gc_env_u32("GC_FRAME_BLOCKS", 1u, GC_MF_MAX_FRAME_BLOCKS, &fb);     // test hook: small frames (as in gc_ctx_create)
    if (codec != GC_CODEC_BROTLI && fb > 1u) {                                                                   // overlapping frames: the unit is the group (zstd 16-22: 32 MiB, FLZMA2 5-6: 16 MiB, 7-9: 64 MiB)
        uint32_t grp = codec == GC_CODEC_ZSTD ? zstd_group_blocks(level) : flzma2_group_blocks(level);
        const bool grpHook = gc_env_u32("GC_MF_GROUP", 1u, 65535u, &grp);                                        // test hooks, read as the compress paths read them (small overlapping frames: shards must not cut a group)
        uint32_t stride = codec == GC_CODEC_ZSTD ? zstd_stride_blocks(level) : flzma2_stride_blocks(level); gc_env_u32("GC_MF_STRIDE", 1u, GC_MF_MAX_FRAME_BLOCKS, &stride);
        const uint32_t fbDefault = codec == GC_CODEC_ZSTD ? zstd_frame_blocks(level) : flzma2_frame_blocks(level);
        if (grp > fb && (fb == fbDefault || grpHook) && stride < fb && (fb % stride) == 0u && ((grp - fb) % stride) == 0u) fb = grp;
    }
    if (codec != GC_CODEC_BROTLI) return (size_t)fb * GC_ZSTD_BLOCK_MAX;
    return (size_t)brotli_blocks_per_chunk(level) * GC_ZSTD_BLOCK_MAX;
}

// ---------------------------------------------------------------- ZSTD decoding ----------------------------------------------------------------
extern "C" void gc_zstd_dec_launch_index(hipStream_t st, const uint8_t* src, const GcZdFrame* frames, uint32_t nFrames, GcZdBlock* blocks, uint64_t* frameTot);
extern "C" void gc_zstd_dec_launch_literals(hipStream_t st, const uint8_t* src, uint64_t srcSize, const GcZdFrame* frames, GcZdBlock* blocks, uint32_t nBlocks, uint8_t* litWork,
                                            unsigned long long* prof, const uint32_t* order, uint32_t* ready);
extern "C" void gc_zstd_dec_launch_sequences(hipStream_t st, const uint8_t* src, uint64_t srcSize, const GcZdFrame* frames, GcZdBlock* blocks, uint32_t nBlocks, void* seqWork,
                                             unsigned long long* prof, const uint32_t* order, uint32_t* ready, int several);
extern "C" void gc_zstd_dec_launch_exec(hipStream_t st, const uint8_t* src, uint64_t srcSize, uint8_t* dst, uint64_t dstCap, const GcZdFrame* frames, uint32_t nFrames,
                                        GcZdBlock* blocks, uint32_t* ticket, uint8_t* litWork, uint64_t litWorkSize, void* seqWork, uint64_t* result, unsigned long long* prof,
                                        const uint32_t* ready);

extern "C" void gc_zstd_dec_launch_place(hipStream_t st, const GcZdFrame* frames, uint32_t nFrames, const GcZdBlock* blocks, uint64_t dstCap, GcZdPlace* place, uint64_t* result, uint32_t* ferr);
extern "C" void gc_zstd_dec_launch_spread(hipStream_t st, const uint8_t* src, uint64_t srcSize, uint8_t* dst, const GcZdFrame* frames, const GcZdBlock* blocks, uint32_t nBlocks,
                                          const GcZdPlace* place, const uint8_t* litWork, uint64_t litWorkSize, const void* seqWork, uint32_t* ptr, uint64_t batchBase, uint32_t* ferr);
extern "C" void gc_zstd_dec_launch_chase(hipStream_t st, uint8_t* dstBatch, uint32_t* ptr, uint32_t n, uint32_t hops, uint8_t* pieceDone, uint32_t* counter);
extern "C" void gc_zstd_dec_launch_finish(hipStream_t st, const uint8_t* src, const uint8_t* dst, const GcZdFrame* frames, uint32_t nFrames, uint64_t* result, const uint32_t* ferr);

static int zd_grow(gc_ctx* c, void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return GC_OK;
    hipFree(*p); *p = nullptr; *cap = 0;
    const size_t want = need + need / 8u + 4096u;
    if (hipMalloc(p, want) != hipSuccess) { *p = nullptr; snprintf(c->err, sizeof(c->err), "decoder workspace: out of device memory (%zu bytes)", want); return GC_ERR_NOMEM; }
    *cap = want;
    return GC_OK;
}

// Run-time check of the sequences kernel that takes six blocks per wave (gc_zstd_dec_seqv_kernel).  Its lanes exchange values through quad-permute DPP moves
// which one compiler version folded into their users in a way the MI355X executed wrongly (every match length with extra bits off; right under the emulator
// and with -amdgpu-dpp-combine=false: profiles/r02_dpp_combine.md).  The source keeps the moves apart with an empty asm and tests/test_abi.py reads the
// compiled kernel, but the failure mode is SILENT wrong lengths, so every context decodes one known frame through that kernel before it trusts it: a frame
// of the reference's encoder (level 3; 2475 bytes of content made by the generator below: literals, then matches of 3 .. 1000 bytes, i.e. match-length codes
// with 0 .. 9 extra bits) -- if the content comes back wrong, the context uses the one-block-per-wave kernel from then on.  ROCm 7.2.0 / hipcc of this image: passes.
static const uint8_t kZdSelfTestStream[377] = {
    40,181,47,253,96,171,8,125,11,0,52,19,5,4,139,162,232,28,126,140,152,200,10,190,247,18,179,117,101,245,103,243,254,169,108,127,73,108,172,39,
    22,218,79,1,118,74,146,246,3,199,77,82,244,139,170,168,159,41,237,196,117,185,84,182,147,194,113,52,116,171,133,223,100,236,190,36,175,89,91,203,
    203,45,30,103,223,185,137,63,100,222,208,95,213,177,135,229,161,235,98,200,211,24,158,156,28,148,227,23,70,199,187,96,46,72,225,100,198,183,23,188,
    215,143,54,231,102,52,173,210,255,0,168,5,90,27,130,233,252,77,86,238,231,47,185,154,107,88,250,197,167,115,122,125,95,90,241,98,29,160,171,114,
    217,36,33,154,147,163,115,117,147,92,162,31,79,47,148,0,25,223,63,234,23,90,141,1,54,37,197,56,3,84,151,97,173,185,30,24,155,32,100,19,
    69,197,68,134,81,105,187,190,32,165,253,23,61,98,222,238,106,151,106,56,52,182,187,218,79,107,100,247,38,136,252,185,75,83,175,99,213,133,213,164,
    190,182,242,118,213,1,173,12,221,28,140,146,225,46,152,193,230,37,46,144,119,85,153,131,177,147,73,88,212,44,185,37,133,94,83,17,47,52,242,73,
    69,228,205,221,78,41,223,155,237,26,39,34,105,132,72,174,86,216,109,110,94,141,1,54,14,117,90,88,51,198,218,45,120,86,180,45,227,192,16,15,
    0,0,22,48,29,93,1,201,161,21,144,12,101,64,25,232,128,193,49,8,160,52,42,0,40,141,242,60,228,106,24,67,54,71,202,23,161,14,55,146,
    140,191,227,82,3,208,178,212,152,132,69,135,1,6,207,154,2
};
static void zd_selftest_content(uint8_t* buf /* 2475 */)
{
    uint32_t s = 12345u, n = 0;
    auto rnd = [&]() -> uint8_t { s = s * 1664525u + 1013904223u; return (uint8_t)(s >> 24); };
    for (int i = 0; i < 256; i++) buf[n++] = rnd();
    static const uint32_t lens[16] = { 37, 70, 131, 258, 19, 300, 45, 1000, 64, 65, 3, 4, 5, 35, 36, 99 };
    for (uint32_t k = 0; k < 16u; k++) {
        for (int j = 0; j < 3; j++) buf[n++] = rnd();
        const uint32_t pos = (k * 37u) % 200u, n0 = n;
        for (uint32_t i = 0; i < lens[k]; i++) buf[n++] = buf[pos + (i % (n0 - pos))];
    }
}
static void zd_seqv_selftest(gc_ctx* c)
{
    c->zdSeqvState = -1;                                   // until proven right
    c->zdSelfTest = true;
    uint8_t want[2475], got[2475];
    zd_selftest_content(want);
    gc_zstd_frame fr; size_t nf = 0; uint64_t total = 0;
    uint8_t* dIn = nullptr; uint8_t* dOut = nullptr; size_t produced = 0;
    if (gc_zstd_scan_frames(kZdSelfTestStream, sizeof(kZdSelfTestStream), &fr, 1, &nf, &total) == GC_OK && nf == 1 && total == sizeof(want) &&
        hipMalloc((void**)&dIn, sizeof(kZdSelfTestStream) + 64) == hipSuccess && hipMalloc((void**)&dOut, sizeof(want) + 64) == hipSuccess &&
        hipMemcpy(dIn, kZdSelfTestStream, sizeof(kZdSelfTestStream), hipMemcpyHostToDevice) == hipSuccess &&
        gc_zstd_decompress_device(c, dIn, sizeof(kZdSelfTestStream), dOut, sizeof(want), &fr, 1, &produced) == GC_OK && produced == sizeof(want) &&
        hipMemcpy(got, dOut, sizeof(got), hipMemcpyDeviceToHost) == hipSuccess && memcmp(got, want, sizeof(want)) == 0)
        c->zdSeqvState = 1;
    hipFree(dIn); hipFree(dOut);
    c->zdSelfTest = false;
    c->err[0] = 0;                                          // (a failed check is not an error of the caller's stream)
}

extern "C" int gc_zstd_decompress_device(gc_ctx* c, const void* d_src, size_t n, void* d_dst, size_t dstCap, const gc_zstd_frame* frames, size_t nFrames, size_t* outSize)
{
    if (!c || (!d_src && n) || (!d_dst && dstCap) || (!frames && nFrames)) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->zdSeqvState == 0 && !c->zdSelfTest) zd_seqv_selftest(c);          // once per context: the six-blocks-per-wave kernel against a known frame
    c->zdMs = 0.f; c->zdKms[0] = c->zdKms[1] = c->zdKms[2] = c->zdKms[3] = 0.f; c->zdRounds = 0;
    if (outSize) *outSize = 0;
    if (!nFrames) return GC_OK;
    GcZdFrame* h = (GcZdFrame*)calloc(nFrames, sizeof(GcZdFrame));
    uint64_t* res = (uint64_t*)malloc(nFrames * 16u);           // per frame results; also the (literal bytes, sequence records) totals of the index pass
    if (!h || !res) { free(h); free(res); return GC_ERR_NOMEM; }
    int rc = GC_OK;
    if (!c->zdTicket && hipMalloc((void**)&c->zdTicket, 8) != hipSuccess) rc = GC_ERR_NOMEM;      // [0] the execution kernel's frame ticket, [1] the chase kernel's count
    for (int i = 0; i < 2 && rc == GC_OK; i++) if (!c->zdEv[i] && hipEventCreate(&c->zdEv[i]) != hipSuccess) rc = GC_ERR_HIP;
    // test hook GC_ZD_PROF=1: shader cycles of the execution kernel's phases (thread 0's view, summed over blocks) on stderr
    unsigned long long* zdProf = nullptr;
    { uint32_t v = 0; if (gc_env_u32("GC_ZD_PROF", 1, 1, &v) && hipMalloc((void**)&zdProf, 128) == hipSuccess) hipMemsetAsync(zdProf, 0, 128, c->stream); }
    uint64_t dstOff = 0;
    size_t i = 0;
    uint64_t batchCap = GC_ZD_BATCH_BYTES;
    { uint32_t v = 0; if (gc_env_u32("GC_ZD_BATCH_KIB", 1u, 4u << 20, &v)) batchCap = (uint64_t)v << 10; }       // test hook: small batches
    bool serialRetry = false;                                    // the batch at hand is being decoded a second time, the execution kernel behind the entropy kernels
    while (i < nFrames && rc == GC_OK) {
        // a batch: frames that state their content size (up to batchCap bytes of content, at least one frame), closed by at most one that does not
        const uint64_t dstOffBatch = dstOff;
        size_t j = i; uint64_t off = dstOff, nBlocks = 0;
        for (; j < nFrames; j++) {
            const gc_zstd_frame& f = frames[j];
            if (j > i && (f.flags & GC_ZD_F_SIZE_KNOWN) && off - dstOff + f.content_size > batchCap) break;       // the workspaces grow with the batch (≈ 7 bytes per content byte)
            if (f.src_off > n || f.src_size > n - f.src_off || f.header_size < 6u || f.header_size + 3ull + ((f.flags & GC_ZD_F_CHECKSUM) ? 4u : 0u) > f.src_size || !f.n_blocks || (uint64_t)f.n_blocks * 3u > f.src_size || nBlocks + f.n_blocks > 0x7FFFFFFFull) { rc = GC_ERR_PARAM; break; }
            GcZdFrame& g = h[j];
            g.srcOff = f.src_off; g.srcSize = f.src_size; g.dstOff = off; g.contentSize = f.content_size; g.flags = f.flags; g.hdrSize = f.header_size;
            g.nBlocks = f.n_blocks; g.blockBase = (uint32_t)nBlocks; g.litBase = 0; g.seqBase = 0;
            nBlocks += f.n_blocks;
            if (!(f.flags & GC_ZD_F_SIZE_KNOWN)) { j++; break; }
            if (f.content_size > dstCap - off) { snprintf(c->err, sizeof(c->err), "destination too small"); rc = GC_ERR_DST_SMALL; break; }
            off += f.content_size;
        }
        if (rc != GC_OK) break;
        const size_t cnt = j - i;
        if (cnt > c->zdFramesCap) {
            hipFree(c->zdFrames); hipFree(c->zdResult); hipFree(c->zdTot); c->zdFrames = nullptr; c->zdResult = nullptr; c->zdTot = nullptr; c->zdFramesCap = 0;
            const size_t cap = cnt + cnt / 2u + 64u;
            if (hipMalloc((void**)&c->zdFrames, cap * sizeof(GcZdFrame)) != hipSuccess || hipMalloc((void**)&c->zdResult, cap * 8u) != hipSuccess ||
                hipMalloc((void**)&c->zdTot, cap * 16u) != hipSuccess) { rc = GC_ERR_NOMEM; break; }
            c->zdFramesCap = cap;
        }
        if ((rc = zd_grow(c, (void**)&c->zdBlocks, &c->zdBlocksCap, (size_t)nBlocks * sizeof(GcZdBlock))) != GC_OK) break;      // capacity in bytes
        // index pass: block table, per-frame workspace needs
        if (hipMemcpyAsync(c->zdFrames, h + i, cnt * sizeof(GcZdFrame), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = GC_ERR_HIP; break; }
        hipEventRecord(c->zdEv[0], c->stream);
        gc_zstd_dec_launch_index(c->stream, (const uint8_t*)d_src, c->zdFrames, (uint32_t)cnt, c->zdBlocks, c->zdTot);
        hipEventRecord(c->evPart[1][0], c->stream);
        if (hipMemcpyAsync(res, c->zdTot, cnt * 16u, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            snprintf(c->err, sizeof(c->err), "index kernel failed: %s", hipGetErrorString(hipGetLastError())); rc = GC_ERR_HIP; break;
        }
        uint64_t litTot = 0, seqTot = 0;
        for (size_t k = 0; k < cnt; k++) { h[i + k].litBase = litTot; h[i + k].seqBase = seqTot; litTot += res[2u * k]; seqTot += res[2u * k + 1u]; }
        if ((rc = zd_grow(c, (void**)&c->zdLit, &c->zdLitCap, (size_t)litTot + 64u)) != GC_OK) break;
        if ((rc = zd_grow(c, &c->zdSeq, &c->zdSeqCap, (size_t)seqTot * 16u + 64u)) != GC_OK) break;
        if (hipMemcpyAsync(c->zdFrames, h + i, cnt * sizeof(GcZdFrame), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemsetAsync(c->zdTicket, 0, 4, c->stream) != hipSuccess) { rc = GC_ERR_HIP; break; }
        // literals on stream2, sequences on the main stream (both only need the block table), the execution kernel on stream3.  With few frames
        // (fewer than 5/8 of the CUs would host an execution workgroup) the execution kernel starts FIRST and follows the entropy kernels block by
        // block (a counter per block says when its one or two entropy workgroups are through), so that the two stages overlap; the entropy
        // kernels then take the blocks round by round over the frames (block 0 of every frame, block 1, ...), so that every frame's first blocks
        // come first.  With many frames the execution kernel would crowd the entropy kernels out: it runs behind them, blocks in plain order.
        // (The emulator runs one launch after the other: there the entropy kernels always come first.)
        bool overlap = cnt <= 160u;
#ifdef HIPEMU
        overlap = false;
#endif
        { uint32_t v = 0; if (gc_env_u32("GC_ZD_OVERLAP", 0, 1, &v)) overlap = overlap && v != 0u; }
        if (serialRetry) overlap = false;
        // The wide execution (all blocks of all frames at once through byte pointers and pointer jumping, see gc_zstd_dec.hip) instead of one workgroup
        // per frame that copies its blocks in order (~0.27 GB/s per frame): 1 GB in 120 frames 13.7 ms against 49 ms, and it does not care how few
        // the frames are (one frame of 128 MiB: 4.7 against 460 ms).  It needs 4 bytes of workspace per content byte; frames of one block each stay
        // with the frame kernel (nothing to gain there).  Hook GC_ZD_WIDE = 0 / 1 forces the choice.
        bool wide = nBlocks >= 2u * cnt;
        // the sequences kernel that takes several blocks per wave: less work per block, but a longer chain per sequence -- it wins once there are
        // enough blocks to fill the machine (1 GB: 25.0 -> 16.0 ms; 256 blocks: 9.7 -> 12.0 ms).  Hook GC_ZD_SEQV = 0 / 1 forces the choice.
        int seqSeveral = nBlocks >= 1024u ? 1 : 0;
        { uint32_t v = 0; if (gc_env_u32("GC_ZD_SEQV", 0, 1, &v)) seqSeveral = (int)v; }
        if (c->zdSelfTest) seqSeveral = 1;                       // the self-check is about that kernel
        else if (c->zdSeqvState < 0) seqSeveral = 0;             // it decoded the known frame wrongly on this device / build: not used
        { uint32_t v = 0; if (gc_env_u32("GC_ZD_WIDE", 0, 1, &v)) wide = v != 0u; }
        if (wide) overlap = false;
        if ((rc = zd_grow(c, (void**)&c->zdOrder, &c->zdOrderCap, (size_t)nBlocks * 4u)) != GC_OK) break;
        if ((rc = zd_grow(c, (void**)&c->zdReady, &c->zdReadyCap, (size_t)nBlocks * 4u)) != GC_OK) break;
        {
            uint32_t* ord = (uint32_t*)malloc((size_t)nBlocks * 4u);
            if (!ord) { rc = GC_ERR_NOMEM; break; }
            uint32_t maxB = 0; size_t w = 0;
            for (size_t k = 0; k < cnt; k++) if (h[i + k].nBlocks > maxB) maxB = h[i + k].nBlocks;
            if (overlap && (uint64_t)maxB * cnt <= 4u * nBlocks + 1024u) {
                for (uint32_t r = 0; r < maxB; r++) for (size_t k = 0; k < cnt; k++) if (r < h[i + k].nBlocks) ord[w++] = h[i + k].blockBase + r;
            } else for (w = 0; w < nBlocks; w++) ord[w] = (uint32_t)w;          // (also for very uneven frames, where the loop above would be quadratic)
            const bool okc = hipMemcpyAsync(c->zdOrder, ord, (size_t)nBlocks * 4u, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
                             hipMemsetAsync(c->zdReady, 0, (size_t)nBlocks * 4u, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
            free(ord);
            if (!okc) { rc = GC_ERR_HIP; break; }
        }
        hipEventRecord(c->evPart[0][0], c->stream);
        hipStreamWaitEvent(c->stream2, c->evPart[0][0], 0);
        hipStreamWaitEvent(c->stream3, c->evPart[0][0], 0);
        hipEventRecord(c->evPart[1][3], c->stream3);
        if (overlap) gc_zstd_dec_launch_exec(c->stream3, (const uint8_t*)d_src, n, (uint8_t*)d_dst, dstCap, c->zdFrames, (uint32_t)cnt, c->zdBlocks, c->zdTicket,
                                             c->zdLit, litTot + 64u, c->zdSeq, c->zdResult, zdProf, c->zdReady);
        hipEventRecord(c->evPart[1][2], c->stream2);
        gc_zstd_dec_launch_literals(c->stream2, (const uint8_t*)d_src, n, c->zdFrames, c->zdBlocks, (uint32_t)nBlocks, c->zdLit, zdProf, c->zdOrder, c->zdReady);
        hipEventRecord(c->evPart[0][1], c->stream2);
        gc_zstd_dec_launch_sequences(c->stream, (const uint8_t*)d_src, n, c->zdFrames, c->zdBlocks, (uint32_t)nBlocks, c->zdSeq, zdProf, c->zdOrder, c->zdReady, seqSeveral);
        hipEventRecord(c->evPart[1][1], c->stream);
        hipStreamWaitEvent(c->stream, c->evPart[0][1], 0);
        if (wide) {
            hipEventRecord(c->evPart[1][3], c->stream);
            if ((rc = zd_grow(c, (void**)&c->zdPlace, &c->zdPlaceCap, (size_t)nBlocks * sizeof(GcZdPlace))) != GC_OK) break;
            if ((rc = zd_grow(c, (void**)&c->zdFerr, &c->zdFerrCap, cnt * 4u)) != GC_OK) break;
            gc_zstd_dec_launch_place(c->stream, c->zdFrames, (uint32_t)cnt, c->zdBlocks, dstCap, c->zdPlace, c->zdResult, c->zdFerr);
            if (hipMemcpyAsync(res, c->zdResult, cnt * 8u, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
                snprintf(c->err, sizeof(c->err), "decode kernels failed: %s", hipGetErrorString(hipGetLastError())); rc = GC_ERR_HIP; break;
            }
            uint64_t extent = 0;                                   // content bytes of the batch, from its first byte
            for (size_t k = 0; k < cnt; k++) { const uint64_t e = h[i + k].dstOff - h[i].dstOff + (res[k] & 0x00FFFFFFFFFFFFFFull); if (e > extent) extent = e; }
            const uint64_t padded = (extent + 4095u) & ~4095ull;
            uint32_t noRoom = 0; gc_env_u32("GC_ZD_WIDE_NOMEM", 1u, 1u, &noRoom);               // test hook: as if the workspace could not be had
            if (extent > GC_ZD_WIDE_MAX) noRoom = 1;               // (one frame beyond the reach of the 32-bit positions)
            if (noRoom || zd_grow(c, (void**)&c->zdPtr, &c->zdPtrCap, (size_t)padded * 4u + 16u) != GC_OK || zd_grow(c, (void**)&c->zdDone, &c->zdDoneCap, (size_t)(padded / 1024u) + 16u) != GC_OK) {
                wide = false; c->err[0] = 0;                       // no room for the pointers: the frame kernel does it
            }
        }
        if (wide) {
            uint64_t extent = 0;
            for (size_t k = 0; k < cnt; k++) { const uint64_t e = h[i + k].dstOff - h[i].dstOff + (res[k] & 0x00FFFFFFFFFFFFFFull); if (e > extent) extent = e; }
            const uint64_t padded = (extent + 4095u) & ~4095ull;
            if (hipMemsetAsync(c->zdPtr, 0xFF, (size_t)padded * 4u, c->stream) != hipSuccess || hipMemsetAsync(c->zdDone, 0, (size_t)(padded / 1024u) + 16u, c->stream) != hipSuccess) { rc = GC_ERR_HIP; break; }
            gc_zstd_dec_launch_spread(c->stream, (const uint8_t*)d_src, n, (uint8_t*)d_dst, c->zdFrames, c->zdBlocks, (uint32_t)nBlocks, c->zdPlace, c->zdLit, litTot + 64u, c->zdSeq,
                                      c->zdPtr, h[i].dstOff, c->zdFerr);
            uint32_t round = 0, hops = 0;                          // (0: the kernel's own number of links per round; test hook GC_ZD_HOPS)
            gc_env_u32("GC_ZD_HOPS", 1u, 64u, &hops);
            for (; extent && round < 64u; round++) {
                uint32_t left = 0;
                hipMemsetAsync(c->zdTicket + 1, 0, 4, c->stream);
                gc_zstd_dec_launch_chase(c->stream, (uint8_t*)d_dst + h[i].dstOff, c->zdPtr, (uint32_t)extent, hops, c->zdDone, c->zdTicket + 1);
                if (hipMemcpyAsync(&left, c->zdTicket + 1, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
                    snprintf(c->err, sizeof(c->err), "decode kernels failed: %s", hipGetErrorString(hipGetLastError())); rc = GC_ERR_HIP; break;
                }
                if (!left) break;
            }
            if (rc != GC_OK) break;
            if (round >= 64u) { snprintf(c->err, sizeof(c->err), "wide execution did not settle"); rc = GC_ERR_HIP; break; }      // (a chain of 2^32 bytes takes 33 rounds)
            c->zdRounds += round + 1u;
            gc_zstd_dec_launch_finish(c->stream, (const uint8_t*)d_src, (const uint8_t*)d_dst, c->zdFrames, (uint32_t)cnt, c->zdResult, c->zdFerr);
            hipEventRecord(c->evPart[1][4], c->stream);
        } else if (!overlap) {
            hipEventRecord(c->evPart[1][3], c->stream);
            gc_zstd_dec_launch_exec(c->stream, (const uint8_t*)d_src, n, (uint8_t*)d_dst, dstCap, c->zdFrames, (uint32_t)cnt, c->zdBlocks, c->zdTicket,
                                    c->zdLit, litTot + 64u, c->zdSeq, c->zdResult, zdProf, c->zdReady);
            hipEventRecord(c->evPart[1][4], c->stream);
        } else {
            hipEventRecord(c->evPart[1][4], c->stream3);
            hipStreamWaitEvent(c->stream, c->evPart[1][4], 0);
        }
        hipEventRecord(c->zdEv[1], c->stream);
        if (hipMemcpyAsync(res, c->zdResult, cnt * 8u, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            snprintf(c->err, sizeof(c->err), "decode kernels failed: %s", hipGetErrorString(hipGetLastError())); rc = GC_ERR_HIP; break;
        }
        float ms = 0.f; if (hipEventElapsedTime(&ms, c->zdEv[0], c->zdEv[1]) == hipSuccess) c->zdMs += ms;
        if (hipEventElapsedTime(&ms, c->zdEv[0], c->evPart[1][0]) == hipSuccess) c->zdKms[0] += ms;
        if (hipEventElapsedTime(&ms, c->evPart[1][2], c->evPart[0][1]) == hipSuccess) c->zdKms[1] += ms;
        if (hipEventElapsedTime(&ms, c->evPart[0][0], c->evPart[1][1]) == hipSuccess) c->zdKms[2] += ms;
        if (hipEventElapsedTime(&ms, c->evPart[1][3], c->evPart[1][4]) == hipSuccess) c->zdKms[3] += ms;
        bool retryNow = false;
        uint32_t failFirst = 0; gc_env_u32("GC_ZD_FAIL_FIRST", 1u, 1u, &failFirst);            // test hook: the first pass over a batch reports its first frame as damaged
        for (size_t k = 0; k < cnt; k++) {
            uint32_t st = (uint32_t)(res[k] >> 56);
            if (failFirst && !serialRetry && k == 0u) st = GC_ZD_CORRUPT;
            const uint64_t produced = res[k] & 0x00FFFFFFFFFFFFFFull;
            if (st == GC_ZD_OK) { dstOff = h[i + k].dstOff + produced; continue; }
            // The execution kernel that starts before the entropy kernels gives up (and reports its frame as damaged) when their blocks do not arrive --
            // which is what happens to a sound stream where launches are serialised (rocprofv3 --pmc: run r4pmcdec, the self-check frame failed this
            // way) or another process holds the device.  Such a batch is decoded once more with the kernels one after the other before it is
            // called damaged; a damaged stream fails again, one pass later.
            // (only GC_ZD_CORRUPT can come from a wait that ran out: a small destination, an unsupported frame, a checksum or a size mismatch would fail the same way again)
            if ((overlap || failFirst) && st == GC_ZD_CORRUPT && !serialRetry) { retryNow = true; break; }
            snprintf(c->err, sizeof(c->err), "frame %zu: %s", i + k, st == GC_ZD_DST_SMALL ? "destination too small" : st == GC_ZD_CHECKSUM ? "content checksum mismatch" :
                     st == GC_ZD_SIZE ? "content size field does not match" : st == GC_ZD_UNSUPPORTED ? "unsupported frame" : "corrupted data");
            rc = st == GC_ZD_DST_SMALL ? GC_ERR_DST_SMALL : (st == GC_ZD_UNSUPPORTED ? GC_ERR_PARAM : GC_ERR_CORRUPT);
            break;
        }
        if (retryNow) { serialRetry = true; dstOff = dstOffBatch; continue; }               // the same frames again
        serialRetry = false;
        i = j;
    }
    free(h); free(res);
    if (zdProf) {
        unsigned long long pv[16] = { 0 };
        uint32_t dbg_ = 0;
        if (hipMemcpy(pv, zdProf, 128, hipMemcpyDeviceToHost) == hipSuccess && gc_env_u32("GC_ZD_SEQV_DBG", 1u, 1u, &dbg_)) { for (int q = 0; q < 16; q++) fprintf(stderr, "[dbg %2d] %016llx\n", q, pv[q]); }
        else if (pv[3])
            fprintf(stderr, "[GC_ZD_PROF] compressed blocks %llu: cycles per block pass1 %.0f pass2 %.0f flush %.0f; pass 2: %.1f groups of 64 sequences per block, %.2f rounds per group of which %.2f for one special match\n", pv[3],
                    (double)pv[0] / pv[3], (double)pv[1] / pv[3], (double)pv[2] / pv[3], (double)pv[5] / pv[3], pv[5] ? (double)pv[4] / pv[5] : 0.0, pv[5] ? (double)pv[6] / pv[5] : 0.0);
        if (pv[12]) fprintf(stderr, "[GC_ZD_PROF] entropy kernel, cycles per block: sequences wave tables %.0f decode %.0f; literals wave tree %.0f streams %.0f\n",
                            (double)pv[8] / pv[12], (double)pv[9] / pv[12], (double)pv[10] / pv[12], (double)pv[11] / pv[12]);
        hipFree(zdProf);
    }
    if (rc == GC_OK && outSize) *outSize = (size_t)dstOff;
    return rc;
}

// One Filter() call of a 7-Zip pre-filter on a host buffer (include/gpucodec.h): the context's device buffers of the host-buffer decoder serve as staging.
extern "C" int gc_filter_host(gc_ctx* c, int kind, void* data, size_t n, uint32_t pc, int encoding, unsigned delta, unsigned char* state, size_t* processed)
{
    if (processed) *processed = 0;
    if (!c || (!data && n)) return GC_ERR_PARAM;
    const bool x86 = kind == GC_FILTER_X86, dl = kind == GC_FILTER_DELTA;
    if (!x86 && !dl && (kind < GC_BRA_ARM64 || kind > GC_BRA_RISCV)) return GC_ERR_PARAM;
    if ((x86 || dl) && !state) return GC_ERR_PARAM;
    if (dl && (delta < 1u || delta > 256u)) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    if (!n) return GC_OK;
    if (n > c->dInCap) { hipFree(c->dIn); c->dIn = nullptr; c->dInCap = 0; if (hipMalloc((void**)&c->dIn, n + 64) != hipSuccess) return GC_ERR_NOMEM; c->dInCap = n; }
    if (n > c->dOutCap) { hipFree(c->dOut); c->dOut = nullptr; c->dOutCap = 0; if (hipMalloc((void**)&c->dOut, n + 64) != hipSuccess) return GC_ERR_NOMEM; c->dOutCap = n; }
    const GcStreamScope onMine(c->stream);                           // (the converters run on this context's stream, in order with the copies)
    HIPCHK(c, hipMemcpyAsync(c->dIn, data, n, hipMemcpyHostToDevice, c->stream));
    size_t done = 0; int rc;
    if (x86) { uint32_t st; memcpy(&st, state, 4); rc = gc_bra_x86_convert_device(c->dIn, c->dOut, n, pc, encoding, &st, &done); memcpy(state, &st, 4); }
    else if (dl) { rc = gc_delta_convert_device(c->dIn, c->dOut, n, delta, encoding, state); done = n; }
    else rc = gc_bra_convert_device(kind, c->dIn, c->dOut, n, pc, encoding, &done);
    if (rc != GC_OK) { snprintf(c->err, sizeof(c->err), "filter %d failed on the device", kind); return rc; }
    HIPCHK(c, hipMemcpyAsync(data, c->dOut, n, hipMemcpyDeviceToHost, c->stream));   // (bytes behind `done` come back unchanged)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (processed) *processed = done;
    return GC_OK;
}

extern "C" int gc_zstd_decompress_selfcheck(gc_ctx* c, int* state)
{
    if (!c || !state) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->zdSeqvState == 0) zd_seqv_selftest(c);
    *state = c->zdSeqvState;
    return GC_OK;
}
extern "C" int gc_zstd_decompress_timing(gc_ctx* c, float* ms) { if (!c || !ms) return GC_ERR_PARAM; *ms = c->zdMs; return GC_OK; }
extern "C" int gc_zstd_decompress_kernel_timing(gc_ctx* c, float ms[4]) { if (!c || !ms) return GC_ERR_PARAM; for (int i = 0; i < 4; i++) ms[i] = c->zdKms[i]; return GC_OK; }

extern "C" int gc_zstd_decompress_wide_rounds(gc_ctx* c, unsigned* rounds) { if (!c || !rounds) return GC_ERR_PARAM; *rounds = c->zdRounds; return GC_OK; }

extern "C" int gc_zstd_decompress_host(gc_ctx* c, const void* src, size_t n, void* dst, size_t dstCap, size_t* outSize)
{
    if (!c || (!src && n) || (!dst && dstCap)) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    if (outSize) *outSize = 0;
    size_t nFrames = 0;
    int rc = gc_zstd_scan_frames(src, n, nullptr, 0, &nFrames, nullptr);
    if (rc != GC_OK) { snprintf(c->err, sizeof(c->err), "not a zstd stream this decoder handles (frame scan failed)"); return rc; }
    if (!nFrames) return GC_OK;
    gc_zstd_frame* fr = (gc_zstd_frame*)malloc(nFrames * sizeof(gc_zstd_frame));
    if (!fr) return GC_ERR_NOMEM;
    uint64_t total = 0;
    rc = gc_zstd_scan_frames(src, n, fr, nFrames, &nFrames, &total);
    if (rc == GC_OK && total != ~0ull && total > dstCap) { snprintf(c->err, sizeof(c->err), "destination too small: need %llu bytes", (unsigned long long)total); rc = GC_ERR_DST_SMALL; }
    if (rc == GC_OK && n > c->dInCap) { hipFree(c->dIn); c->dIn = nullptr; c->dInCap = 0; if (hipMalloc((void**)&c->dIn, n + 64) != hipSuccess) rc = GC_ERR_NOMEM; else c->dInCap = n; }
    const size_t need = total != ~0ull ? (size_t)total : dstCap;
    if (rc == GC_OK && need > c->dOutCap) { hipFree(c->dOut); c->dOut = nullptr; c->dOutCap = 0; if (hipMalloc((void**)&c->dOut, need + 64) != hipSuccess) rc = GC_ERR_NOMEM; else c->dOutCap = need; }
    size_t produced = 0;
    if (rc == GC_OK && hipMemcpyAsync(c->dIn, src, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = GC_ERR_HIP;
    if (rc == GC_OK) rc = gc_zstd_decompress_device(c, c->dIn, n, c->dOut, need, fr, nFrames, &produced);
    if (rc == GC_OK && produced && (hipMemcpyAsync(dst, c->dOut, produced, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) rc = GC_ERR_HIP;
    free(fr);
    if (rc == GC_OK && outSize) *outSize = produced;
    return rc;
}

// ---- BROTLI decoder (gc_brotli_dec.hip): the context's stream and buffers around gc_brd_decode
extern "C" int gc_brotli_decompress_device(gc_ctx* c, const void* d_src, size_t n, void* d_dst, size_t dstCap, const gc_brotli_chunk* chunks, size_t nChunks, size_t* outSize)
{
    if (!c || (!d_src && n) || (!chunks && nChunks) || !outSize) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    *outSize = 0;
#ifdef GC_TEST_HOOKS
    { const char* e = getenv("GC_BRD_LDS"); c->brd.ldsCap = e ? (uint32_t)atoi(e) : 0u; e = getenv("GC_BRD_INSTANCE"); c->brd.instance = e ? (uint32_t)atoi(e) : 0u; }
#endif
    for (size_t i = 0; i < nChunks; i++) if (chunks[i].src_off > n || chunks[i].src_size > n - chunks[i].src_off) { snprintf(c->err, sizeof(c->err), "brotli chunk %zu lies outside the %zu input bytes", i, n); return GC_ERR_PARAM; }
    return gc_brd_decode(c->stream, &c->brd, (const uint8_t*)d_src, chunks, nChunks, (uint8_t*)d_dst, dstCap, outSize, c->err, sizeof(c->err));
}
extern "C" int gc_brotli_decompress_host(gc_ctx* c, const void* src, size_t n, void* dst, size_t dstCap, size_t* outSize)
{
    if (!c || (!src && n) || (!dst && dstCap)) return GC_ERR_PARAM;
    HIPCHK(c, hipSetDevice(c->device));
    if (outSize) *outSize = 0;
    if (n == 0) return GC_OK;
    size_t nChunks = 0, consumed = 0;
    gc_brotli_chunk one; gc_brotli_chunk* ch = nullptr;
    uint32_t magic = 0; if (n >= 4) memcpy(&magic, src, 4);
    int rc = GC_OK;
    if (magic != 0x184D2A50u) {                                   // a bare RFC 7932 stream: one chunk
        one.src_off = 0; one.src_size = (uint32_t)n; one.capacity = (uint32_t)(dstCap < 0xFFFF0000u ? dstCap : 0xFFFF0000u);
        if (n > 0xFFFFFFFFu) return GC_ERR_PARAM;
        ch = &one; nChunks = 1;
    } else {
        rc = gc_brotli_scan_prefix(src, n, nullptr, 0, &nChunks, nullptr, &consumed);
        if (rc == GC_OK && consumed != n) rc = GC_ERR_CORRUPT;   // the input ends inside a frame
        if (rc != GC_OK) { snprintf(c->err, sizeof(c->err), "not a whole brotli-mt stream (frame scan failed)"); return rc; }
        ch = (gc_brotli_chunk*)malloc((nChunks ? nChunks : 1) * sizeof(gc_brotli_chunk));
        if (!ch) return GC_ERR_NOMEM;
        rc = gc_brotli_scan_prefix(src, n, ch, nChunks, &nChunks, nullptr, &consumed);
    }
    if (rc == GC_OK && n > c->dInCap) { hipFree(c->dIn); c->dIn = nullptr; c->dInCap = 0; if (hipMalloc((void**)&c->dIn, n + 64) != hipSuccess) rc = GC_ERR_NOMEM; else c->dInCap = n; }
    if (rc == GC_OK && dstCap > c->dOutCap) { hipFree(c->dOut); c->dOut = nullptr; c->dOutCap = 0; if (hipMalloc((void**)&c->dOut, dstCap + 64) != hipSuccess) rc = GC_ERR_NOMEM; else c->dOutCap = dstCap; }
    size_t produced = 0;
    if (rc == GC_OK && hipMemcpyAsync(c->dIn, src, n, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = GC_ERR_HIP;
#ifdef GC_TEST_HOOKS
    { const char* e = getenv("GC_BRD_LDS"); c->brd.ldsCap = e ? (uint32_t)atoi(e) : 0u; e = getenv("GC_BRD_INSTANCE"); c->brd.instance = e ? (uint32_t)atoi(e) : 0u; }
#endif
    if (rc == GC_OK) rc = gc_brd_decode(c->stream, &c->brd, c->dIn, ch, nChunks, c->dOut, dstCap, &produced, c->err, sizeof(c->err));
    if (rc == GC_OK && produced && (hipMemcpyAsync(dst, c->dOut, produced, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)) rc = GC_ERR_HIP;
    if (ch != &one) free(ch);
    if (rc == GC_OK && outSize) *outSize = produced;
    return rc;
}
extern "C" int gc_brotli_decompress_timing(gc_ctx* c, float* ms) { if (!c || !ms) return GC_ERR_PARAM; *ms = c->brd.ms; return GC_OK; }
