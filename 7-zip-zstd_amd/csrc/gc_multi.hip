// gc_multi.hip -- host scheduler of libgpucodec.so: one host buffer range-split over the GPU contexts of a node.
//
// The counterpart of the job front ends of the reference, with GPU contexts in place of worker threads: ZSTDMT_compressStream_generic
// cuts the input into jobs, hands them to pool threads and flushes their output in job order (C/zstd/zstdmt_compress.c:1184-1247,
// :1450-1530); BROTLIMT_compressCCtx does the same with chunks and a write list ordered by frame number
// (C/zstdmt/brotli-mt_compress.c:209-333); FL2 codes dictionary blocks one after another (C/fast-lzma2/fl2_compress.c:1020).
// Here: pieces (multiples of the codec's independence grain, gc_codec_grain) are dealt to workers in order; a worker owns one
// gc_ctx (its own HIP streams and workspace) on one device; two workers per device by default, so that while one piece's kernels
// run, the other worker's H2D / D2H copies use the link.  A worker commits its piece in piece order -- it takes the running output
// offset as soon as all earlier pieces have announced their sizes -- and then copies its bytes device -> final place, so the D2H
// copies of neighbouring pieces overlap as well.  No data-path collective: compressed pieces only meet in the caller's buffer.
//
// Host code only (no kernels); plain C++ threads.  Under tests/emu the same file runs against the emulated runtime.
#include "gpucodec.h"
#include "gc_common.h"
#ifdef HIPEMU
#include "hip_runtime_stub.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

struct gc_multi {
    std::vector<int> workerDevice;        // device of worker w (workers are dealt to devices round-robin: few pieces spread over all GPUs)
    std::vector<gc_ctx*> ctx;             // created on first use, on the worker's own thread
    char err[256];
};

extern "C" int gc_multi_create(gc_multi** out, const int* devices, int nDevices, int ctxPerDevice)
{
    if (!out) return GC_ERR_PARAM;
    *out = nullptr;
    const int have = gc_device_count();
    if (have <= 0) return GC_ERR_NO_DEVICE;
    std::vector<int> dev;
    if (!devices || nDevices <= 0) for (int d = 0; d < have; d++) dev.push_back(d);
    else for (int i = 0; i < nDevices; i++) { if (devices[i] < 0 || devices[i] >= have) return GC_ERR_NO_DEVICE; dev.push_back(devices[i]); }
    if (ctxPerDevice <= 0) ctxPerDevice = 2;
    if (ctxPerDevice > 8) ctxPerDevice = 8;
    gc_multi* m = new (std::nothrow) gc_multi();
    if (!m) return GC_ERR_NOMEM;
    m->err[0] = 0;
    for (int k = 0; k < ctxPerDevice; k++) for (size_t i = 0; i < dev.size(); i++) m->workerDevice.push_back(dev[i]);
    m->ctx.assign(m->workerDevice.size(), nullptr);
    // the first context is created here so that a machine without a usable device fails at creation, like gc_ctx_create
    const int rc = gc_ctx_create(&m->ctx[0], m->workerDevice[0]);
    if (rc != GC_OK) { delete m; return rc; }
    *out = m;
    return GC_OK;
}

extern "C" void gc_multi_destroy(gc_multi* m)
{
    if (!m) return;
    for (gc_ctx* c : m->ctx) if (c) gc_ctx_destroy(c);
    delete m;
}

extern "C" int gc_multi_workers(const gc_multi* m) { return m ? (int)m->workerDevice.size() : 0; }
extern "C" const char* gc_multi_last_error(const gc_multi* m) { return m ? m->err : "no scheduler"; }

extern "C" size_t gc_multi_piece_bytes(int codec, int level)
{
    // FLZMA2's model and range-coder kernels take as long as their longest chain however few segments there are, so its pieces are larger
    // (measured on 211.9 MB: 64 MiB pieces 1.7 GB/s, one piece 3.0 GB/s)
    const size_t grain = gc_codec_grain(codec, level), target = (size_t)(codec == GC_CODEC_FLZMA2 ? 256u : 64u) << 20;
    const size_t k = target / grain;
    return (k ? k : 1u) * grain;
}

namespace {
struct Job {
    gc_multi* m; int codec, level; unsigned flags;
    const uint8_t* src; size_t n; uint8_t* dst; size_t dstCap; size_t piece, nPieces;
    std::atomic<size_t> next{0};
    std::mutex mu; std::condition_variable cv;
    size_t commitIdx = 0, commitOff = 0;  // pieces [0, commitIdx) have announced their sizes; their bytes end at commitOff
    int rc = GC_OK;                       // first failure (under mu)
};

void job_fail(Job& j, size_t k, int rc, const char* what, gc_ctx* c)
{
    std::unique_lock<std::mutex> lk(j.mu);
    if (j.rc == GC_OK) { j.rc = rc; snprintf(j.m->err, sizeof(j.m->err), "piece %zu: %s: %s", k, what, c ? gc_last_error_message(c) : ""); }
    // later pieces must not wait for this one
    j.cv.wait(lk, [&] { return j.commitIdx >= k; });
    if (j.commitIdx == k) j.commitIdx = k + 1;
    j.cv.notify_all();
}

void worker(Job& j, size_t w)
{
    gc_multi* m = j.m;
    for (;;) {
        const size_t k = j.next.fetch_add(1);
        if (k >= j.nPieces) return;
        {   // an earlier piece failed: pass the turn on without working
            std::unique_lock<std::mutex> lk(j.mu);
            if (j.rc != GC_OK) { j.cv.wait(lk, [&] { return j.commitIdx >= k; }); if (j.commitIdx == k) j.commitIdx = k + 1; j.cv.notify_all(); continue; }
        }
        if (!m->ctx[w]) { const int rc = gc_ctx_create(&m->ctx[w], m->workerDevice[w]); if (rc != GC_OK) { m->ctx[w] = nullptr; job_fail(j, k, rc, "gc_ctx_create", nullptr); continue; } }
        gc_ctx* c = m->ctx[w];
        const size_t off = k * j.piece, len = (j.n - off) < j.piece ? (j.n - off) : j.piece;
        unsigned flags = j.codec == GC_CODEC_FLZMA2 ? (j.flags | GC_FLZMA2_NO_END_MARK) : j.flags;    // one end marker for the whole stream, written by the caller below
        if (j.codec == GC_CODEC_BROTLI && (j.flags & GC_BROTLI_PLAIN))                                  // pieces of ONE brotli stream: header in the first, closing meta-block in the last
            flags |= (k > 0 ? GC_BROTLI_NOT_FIRST : 0u) | (k + 1 < j.nPieces ? GC_BROTLI_NOT_LAST : 0u);
        int rc = gc_host_begin(c, j.codec, j.src + off, len, j.level, flags);
        if (rc != GC_OK) { job_fail(j, k, rc, "gc_host_begin", c); continue; }
        size_t sz = 0;
        rc = gc_host_size(c, &sz);
        if (rc != GC_OK) { job_fail(j, k, rc, "gc_host_size", c); continue; }
        size_t at = 0; bool fits = true;
        {
            std::unique_lock<std::mutex> lk(j.mu);
            j.cv.wait(lk, [&] { return j.commitIdx == k; });
            at = j.commitOff;
            fits = j.rc == GC_OK && at + sz <= j.dstCap;
            if (j.rc == GC_OK && !fits) { j.rc = GC_ERR_DST_SMALL; snprintf(m->err, sizeof(m->err), "destination too small at piece %zu", k); }
            if (fits) j.commitOff = at + sz;
            j.commitIdx = k + 1;
            j.cv.notify_all();
        }
        if (!fits) continue;
        rc = gc_host_fetch(c, j.dst + at, sz);
        if (rc != GC_OK) { std::unique_lock<std::mutex> lk(j.mu); if (j.rc == GC_OK) { j.rc = rc; snprintf(m->err, sizeof(m->err), "piece %zu: gc_host_fetch: %s", k, gc_last_error_message(c)); } }
    }
}
}  // namespace

extern "C" int gc_multi_compress_host(gc_multi* m, int codec, const void* src, size_t n, void* dst, size_t dstCap, int level, unsigned flags,
                                      size_t pieceBytes, size_t* outSize)
{
    if (!m || (!src && n) || !dst || codec < GC_CODEC_ZSTD || codec > GC_CODEC_BROTLI) return GC_ERR_PARAM;
    m->err[0] = 0;
    const size_t grain = gc_codec_grain(codec, level);
    size_t piece = pieceBytes ? (pieceBytes + grain - 1u) / grain * grain : gc_multi_piece_bytes(codec, level);
    if (n <= piece) {                     // one piece (or the empty input): the plain single-context call
        const int rc = gc_codec_compress_host(m->ctx[0], codec, src, n, dst, dstCap, level, flags, outSize);
        if (rc != GC_OK) snprintf(m->err, sizeof(m->err), "%s", gc_last_error_message(m->ctx[0]));
        return rc;
    }
    Job j; j.m = m; j.codec = codec; j.level = level; j.flags = flags; j.src = (const uint8_t*)src; j.n = n; j.dst = (uint8_t*)dst; j.dstCap = dstCap;
    j.piece = piece; j.nPieces = (n + piece - 1u) / piece;
    const size_t nWorkers = j.nPieces < m->workerDevice.size() ? j.nPieces : m->workerDevice.size();
    // Pieces are dealt from an atomic counter, so the job completes with however many helpers could be started: a std::thread that cannot be
    // created (std::system_error) or a vector that cannot grow (std::bad_alloc) only costs parallelism -- no exception leaves this extern "C" call
    std::vector<std::thread> th;
    try {
        th.reserve(nWorkers);
        for (size_t w = 1; w < nWorkers; w++) th.emplace_back(worker, std::ref(j), w);
    } catch (...) {}
    worker(j, 0);                         // the calling thread is worker 0
    for (std::thread& t : th) t.join();
    if (j.rc != GC_OK) return j.rc;
    size_t total = j.commitOff;
    if (codec == GC_CODEC_FLZMA2 && !(flags & GC_FLZMA2_NO_END_MARK)) {
        if (total + 1u > dstCap) { snprintf(m->err, sizeof(m->err), "destination too small for the end marker"); return GC_ERR_DST_SMALL; }
        ((uint8_t*)dst)[total++] = 0x00;  // LZMA2 end of stream (C/Lzma2Dec.c:97)
    }
    if (outSize) *outSize = total;
    return GC_OK;
}
