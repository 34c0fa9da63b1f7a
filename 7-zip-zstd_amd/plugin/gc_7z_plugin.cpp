// gc_7z_plugin.cpp -- lib7zgpucodec.so: the 7-Zip codec-plugin surface over libgpucodec's C ABI.
//
// What a 7-Zip host (`7z` built with Z7_EXTERNAL_CODECS; CPP/7zip/UI/Common/LoadCodecs.cpp:531-650) does with a module
// found in its Codecs/ directory, and what therefore has to exist here:
//   dlsym GetModuleProp        -> kInterfaceType must be 0 (IUnknown without virtual destructor)   CodecExports.cpp:360-378
//   dlsym GetNumberOfMethods / GetMethodProperty -> id, name, encoder/decoder class ids            CodecExports.cpp:198-265
//   dlsym CreateEncoder / CreateDecoder / CreateObject -> COM-style coder object                   CodecExports.cpp:127-195
// The coder object mirrors NCompress::NZSTD::CEncoder (CPP/7zip/Compress/ZstdEncoder.h:35-78) and
// NCompress::NLzma2::CFastEncoder (CPP/7zip/Compress/Lzma2Encoder.h:60-100): ICompressCoder,
// ICompressSetCoderMt, ICompressSetCoderProperties, ICompressSetCoderPropertiesOpt, ICompressWriteCoderProperties.
// ENCODERS for the three methods (the hot path of SURVEY.md section 8) and a DECODER for ZSTD (section 8f1; mirrors
// NCompress::NZSTD::CDecoder, CPP/7zip/Compress/ZstdDecoder.h:43-100).  Decoders are looked up by method id, built-in codecs first
// (CreateCoder.cpp:206-232): a host that has a ZSTD decoder of its own keeps using it, a host without one (mainline 7-Zip) gets this one.
//
// No C++ exception crosses the boundary (the reference wraps with COM_TRY, CodecExports.cpp:97-124): nothing below
// throws; allocations are checked.
#include "gc_7z_abi.h"
#include "gpucodec.h"
#include <atomic>
#include <chrono>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <link.h>

#define GC_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct MethodInfo { uint64_t id; const char* name; int kind; int filter; /* -1: a codec; else the gc_filter_host kind of a pre-filter (round 3) */ unsigned align; };
enum { KIND_ZSTD = 0, KIND_FLZMA2 = 1, KIND_BROTLI = 2, KIND_FILTER = 3 };
// Names and ids as registered by the reference (CPP/7zip/Compress/ZstdRegister.cpp:13-17, FastLzma2Register.cpp:13-18,
// BrotliRegister.cpp:13-17).  A host that has these codecs built in (the reference's own 7z.so) resolves a method NAME to its
// built-in encoder first (FindMethod_Index, CPP/7zip/Common/CreateCoder.cpp:160-204), so every method is registered a second
// time under an alias with the SAME id -- the way FastLzma2Register.cpp:13-18 registers FLZMA2 beside LZMA2 under id 0x21:
// `7z a -m0=ZSTDGPU` then runs this module's encoder, the archive records only the id, and any 7-Zip-zstd decodes it.
const MethodInfo kMethods[] = {
    { 0x4F71101, "ZSTD", KIND_ZSTD, -1, 0 },
    { 0x21, "FLZMA2", KIND_FLZMA2, -1, 0 },
    { 0x4F71102, "BROTLI", KIND_BROTLI, -1, 0 },
    { 0x4F71101, "ZSTDGPU", KIND_ZSTD, -1, 0 },
    { 0x21, "FLZMA2GPU", KIND_FLZMA2, -1, 0 },
    { 0x4F71102, "BROTLIGPU", KIND_BROTLI, -1, 0 },
    // 7-Zip's pre-filters on the device, under names of their own with the reference's ids (BcjRegister.cpp:12-15, BranchRegister.cpp:33-57,
    // DeltaFilter.cpp:121-124): `7z a -m0=BCJGPU -m1=ZSTDGPU` runs both on the GPU and any 7-Zip extracts the archive with its built-in filters;
    // `align`: the alignment mask of the branch offset property (BranchRegister.cpp:56-57)
    { 0x3030103, "BCJGPU", KIND_FILTER, GC_FILTER_X86, 0 },
    { 0x3030205, "PPCGPU", KIND_FILTER, GC_BRA_PPC, 0 },
    { 0x3030401, "IA64GPU", KIND_FILTER, GC_BRA_IA64, 0 },
    { 0x3030501, "ARMGPU", KIND_FILTER, GC_BRA_ARM, 0 },
    { 0x3030701, "ARMTGPU", KIND_FILTER, GC_BRA_ARMT, 0 },
    { 0x3030805, "SPARCGPU", KIND_FILTER, GC_BRA_SPARC, 0 },
    { 0xA, "ARM64GPU", KIND_FILTER, GC_BRA_ARM64, 3 },
    { 0xB, "RISCVGPU", KIND_FILTER, GC_BRA_RISCV, 1 },
    { 3, "DELTAGPU", KIND_FILTER, GC_FILTER_DELTA, 0 },
};
const uint32_t kNumMethods = sizeof(kMethods) / sizeof(kMethods[0]);

HRESULT write_all(ISequentialOutStream* out, const void* data, size_t size)     // StreamUtils.cpp:54-100
{
    const uint8_t* p = (const uint8_t*)data;
    while (size) {
        uint32_t cur = size > 0x80000000u ? 0x80000000u : (uint32_t)size, done = 0;
        HRESULT r = out->Write(p, cur, &done);
        p += done; size -= done;
        if (r != S_OK) return r;
        if (done == 0) return E_FAIL;
    }
    return S_OK;
}

HRESULT read_full(ISequentialInStream* in, void* data, size_t* size)            // ReadStream, StreamUtils.cpp:20-40
{
    uint8_t* p = (uint8_t*)data; size_t want = *size; *size = 0;
    while (want) {
        uint32_t cur = want > 0x80000000u ? 0x80000000u : (uint32_t)want, done = 0;
        HRESULT r = in->Read(p, cur, &done);
        p += done; want -= done; *size += done;
        if (r != S_OK) return r;
        if (done == 0) return S_OK;
    }
    return S_OK;
}

HRESULT hresult_of(int gcErr)
{
    switch (gcErr) {
        case GC_OK: return S_OK;
        case GC_ERR_NOMEM: return E_OUTOFMEMORY;
        case GC_ERR_PARAM: return E_INVALIDARG;
        default: return E_FAIL;           // no device / HIP failure: there is no CPU codec to fall back to
    }
}

// Process-wide state (round 3).  A 7z run creates one coder object per folder / archive member, and what is slow about the first one is not its
// compression: opening the HIP runtime and the devices (about 0.2 s) and page-locking the staging buffers.  So the host scheduler, the decoder's
// context and the pinned buffers belong to the PROCESS: created on first use, lent to the coder objects, never torn down (the HIP runtime's own
// exit handlers run in an unspecified order against ours).  `gpu` is held for one batch on the devices at a time -- never across the host's
// Read / Write calls, which may wait for another coder of the same folder (CoderMixerMT runs them as threads).
struct BufSet { uint8_t* in[2] = { nullptr, nullptr }; size_t inCap[2] = { 0, 0 }; uint8_t* out = nullptr; size_t outCap = 0; };
struct Shared {
    std::mutex mu;                                    // guards multi / dec creation and the pool
    std::mutex gpu;                                   // one batch on the devices at a time
    gc_multi* multi = nullptr;                        // host scheduler over every visible GPU, two contexts each (csrc/gc_multi.hip)
    std::atomic<int> multiState{0};                   // 0 not there yet, 1 ready, -1 could not be created
    std::atomic<int> warm{0};                         // the warm-up thread has been started
    std::mutex warmMu; std::thread warmThread;        // ... and is joined before any coder object dies / the module is unloaded (warm_up_join)
    gc_ctx* dec = nullptr;                            // the ZSTD decoder's context (device 0)
    BufSet pool[4]; int nPool = 0;                    // buffer sets handed back by finished Code() calls
};
Shared* shared()
{
    static Shared* s = new (std::nothrow) Shared();   // leaked on purpose, see above
    return s;
}
gc_multi* shared_multi(int* rcOut)
{
    Shared* s = shared();
    if (!s) { *rcOut = GC_ERR_NOMEM; return nullptr; }
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->multi) { const int rc = gc_multi_create(&s->multi, nullptr, 0, 2); if (rc != GC_OK) { s->multi = nullptr; s->multiState.store(-1); *rcOut = rc; return nullptr; } }
    s->multiState.store(1);
    *rcOut = GC_OK;
    return s->multi;
}
// Opening the HIP runtime and the devices takes about 0.2 s -- a third of what `7z a` needs for 1 GB.  It is started in the background as soon as the host asks for an
// encoder object, and Code() spends the wait reading input (the host's reader computes its CRC there): a started `7z a` then pays max(start-up, reading), not their sum.
unsigned test_env_u32(const char* name)                // test hooks of the plugin: compiled in only for the emulator module (tests/emu/Makefile)
{
#ifdef GC_PLUGIN_TEST_HOOKS
    const char* v = getenv(name);
    return v ? (unsigned)strtoul(v, nullptr, 10) : 0u;
#else
    (void)name; return 0u;
#endif
}
void warm_up_async()
{
    Shared* s = shared();
    int zero = 0;
    if (!s || !s->warm.compare_exchange_strong(zero, 1)) return;
    try {
        std::lock_guard<std::mutex> lk(s->warmMu);
        s->warmThread = std::thread([]() {
            const unsigned delay = test_env_u32("GC_PLUGIN_WARM_DELAY_MS");        // (test hook: a slow start-up on a machine that has none)
            if (delay) std::this_thread::sleep_for(std::chrono::milliseconds(delay));
            int rc; shared_multi(&rc);
        });
    } catch (...) {}                                   // no thread to be had: Code() creates the scheduler itself
}
// The warm-up thread is inside the HIP runtime's initialisation for ~0.2 s.  A host that creates an encoder and never reaches Code() (a rejected property, an
// abort) must not get to exit() / static destruction while it is: every encoder object joins it when it dies, and so does the module when it is unloaded.
void warm_up_join()
{
    Shared* s = shared();
    if (!s) return;
    std::lock_guard<std::mutex> lk(s->warmMu);
    if (s->warmThread.joinable() && s->warmThread.get_id() != std::this_thread::get_id()) s->warmThread.join();
}
__attribute__((destructor)) static void plugin_unload() { warm_up_join(); }
gc_ctx* shared_dec_ctx()
{
    Shared* s = shared();
    if (!s) return nullptr;
    std::lock_guard<std::mutex> lk(s->mu);
    if (!s->dec && gc_ctx_create(&s->dec, 0) != GC_OK) s->dec = nullptr;
    return s->dec;
}
void buf_acquire(BufSet* b)
{
    Shared* s = shared();
    *b = BufSet();
    if (!s) return;
    std::lock_guard<std::mutex> lk(s->mu);
    if (s->nPool) *b = s->pool[--s->nPool];
}
void buf_release(BufSet* b)
{
    Shared* s = shared();
    bool kept = false;
    const size_t kKeep = (size_t)256 << 20;             // a long-lived host must not sit on gigabytes of page-locked memory after one large archive
    const bool small = b->inCap[0] <= kKeep && b->inCap[1] <= kKeep && b->outCap <= kKeep;
    if (s && small) { std::lock_guard<std::mutex> lk(s->mu); if (s->nPool < 4) { s->pool[s->nPool++] = *b; kept = true; } }
    if (!kept) { gc_host_free(b->in[0]); gc_host_free(b->in[1]); gc_host_free(b->out); }
    *b = BufSet();
}
// pinned buffer of at least `need` bytes; `keep` bytes of the old one survive a growth
bool buf_grow(uint8_t** buf, size_t* cap, size_t need, size_t keep)
{
    if (need <= *cap) return true;
    uint8_t* nb = (uint8_t*)gc_host_alloc(need);
    if (!nb) return false;
    if (keep) memcpy(nb, *buf, keep);
    gc_host_free(*buf); *buf = nb; *cap = need;
    return true;
}
struct BufLease {                                     // a Code() call's buffers go back to the pool however the call ends
    BufSet b;
    BufLease() { buf_acquire(&b); }
    ~BufLease() { buf_release(&b); }
};

class CGpuEncoder final : public ICompressCoder, public ICompressSetCoderMt, public ICompressSetCoderProperties,
                              public ICompressSetCoderPropertiesOpt, public ICompressWriteCoderProperties {
    ULONG refs_ = 1;
    const int kind_;
    int level_;                                       // ZSTD_CLEVEL_DEFAULT 3 / FL2 default 5 (Lzma2Encoder.cpp:178-239 maps -mx to it)
    uint8_t props_[5] = { 1, 5, 3, 0, 0 };            // ZSTD: CProps{major, minor, level, reserved[2]}  ZstdEncoder.h:17-32
    uint64_t expected_ = 0;
    bool plainBrotli_ = false;
    int codec() const { return kind_ == KIND_ZSTD ? GC_CODEC_ZSTD : (kind_ == KIND_FLZMA2 ? GC_CODEC_FLZMA2 : GC_CODEC_BROTLI); }

public:
    explicit CGpuEncoder(int kind) : kind_(kind), level_(default_level(kind)) {}
    static int default_level(int kind) { return kind == KIND_ZSTD ? 3 : (kind == KIND_FLZMA2 ? 5 : 3); }   // BrotliEncoder.h: _props._level = 3

    HRESULT QueryInterface(const GUID& iid, void** out) override
    {
        if (!out) return E_INVALIDARG;
        *out = nullptr;
        if (iid == IID_IUnknown || iid == IID_ICompressCoder) *out = static_cast<ICompressCoder*>(this);
        else if (iid == IID_ICompressSetCoderMt) *out = static_cast<ICompressSetCoderMt*>(this);
        else if (iid == IID_ICompressSetCoderProperties) *out = static_cast<ICompressSetCoderProperties*>(this);
        else if (iid == IID_ICompressSetCoderPropertiesOpt) *out = static_cast<ICompressSetCoderPropertiesOpt*>(this);
        else if (iid == IID_ICompressWriteCoderProperties) *out = static_cast<ICompressWriteCoderProperties*>(this);
        else return E_NOINTERFACE;
        ++refs_;
        return S_OK;
    }
    ULONG AddRef() override { return ++refs_; }
    ULONG Release() override { if (--refs_ != 0) return refs_; warm_up_join(); delete this; return 0; }      // non-atomic like MyCom.h:380-390

    // the workers are GPU contexts, their number follows the devices; BROTLI with "0 threads" = a plain .br stream without brotli-mt
    // framing, which is how the reference's bare-file handler asks for it (BrotliHandler.cpp:286-291, BrotliEncoder.cpp:166-177)
    HRESULT SetNumberOfThreads(uint32_t n) override { plainBrotli_ = kind_ == KIND_BROTLI && (int)n < 1; return S_OK; }

    HRESULT SetCoderProperties(const PROPID* ids, const PROPVARIANT* props, uint32_t n) override
    {
        // same clamping as the reference for the properties that have a meaning here (ZstdEncoder.cpp:51-230, Lzma2Encoder.cpp:178-239,
        // BrotliEncoder.cpp:30-80); the remaining tuning properties (dictionary, strategy, fast bytes, ...) are accepted and ignored,
        // as the reference's default branches do: the GPU path has one configuration per level
        level_ = default_level(kind_); props_[2] = 3;
        for (uint32_t i = 0; i < n; i++) {
            if (ids[i] != NCoderPropID::kLevel) continue;
            if (kind_ == KIND_FLZMA2 && props[i].vt != VT_UI4) return E_INVALIDARG;           // Lzma2Encoder.cpp SetLzma2Prop: level must be VT_UI4
            const uint32_t v = props[i].ulVal;
            int lv = (int)v;
            if (v < 1) lv = 1;
            if (kind_ == KIND_ZSTD) { if (v > 22) lv = 22; props_[2] = (uint8_t)lv; }   // ZSTD_maxCLevel()
            else if (kind_ == KIND_FLZMA2) { if (v > 9) lv = 9; }                        // FL2_MAX_7Z_CLEVEL
            else { lv = (int)v; if (v > 11) lv = 11; }                                  // BROTLIMT_LEVEL_MIN..MAX = 0..11
            level_ = lv;
        }
        return S_OK;
    }
    HRESULT SetCoderPropertiesOpt(const PROPID* ids, const PROPVARIANT* props, uint32_t n) override
    {
        for (uint32_t i = 0; i < n; i++)
            if (ids[i] == NCoderPropID::kExpectedDataSize && props[i].vt == VT_UI8 && props[i].uhVal) expected_ = props[i].uhVal;
        return S_OK;
    }
    HRESULT WriteCoderProperties(ISequentialOutStream* out) override
    {
        if (kind_ == KIND_FLZMA2) { const uint8_t p = gc_flzma2_dict_prop(level_); return write_all(out, &p, 1); }    // Lzma2Encoder.cpp:353-364
        if (kind_ == KIND_BROTLI) { const uint8_t p[3] = { 1, 2, (uint8_t)level_ }; return write_all(out, p, 3); }    // {BROTLI_VERSION_MAJOR, _MINOR, level}: BrotliEncoder.h:18-32, C/brotli/common/version.h:20-21
        return write_all(out, props_, sizeof(props_));
    }

    // The input is taken in batches of whole pieces (gc_multi_piece_bytes: a multiple of the codec's independence grain -- 8 MiB
    // frames, or the brotli-mt chunk of `level` MiB), one piece per GPU context.  zstd and brotli: the stream equals what one
    // whole-buffer call would produce; FLZMA2: every piece starts with a dictionary reset (as every dictionary block of the
    // reference does after its overlap, fl2_compress.c:1020) and the single end marker follows the last one.
    HRESULT Code(ISequentialInStream* in, ISequentialOutStream* out, const uint64_t*, const uint64_t*, ICompressProgressInfo* progress) override
    {
        if (!in || !out) return E_INVALIDARG;
        size_t piece = gc_multi_piece_bytes(codec(), level_);
        { const unsigned kib = test_env_u32("GC_PLUGIN_PIECE_KIB"); if (kib) piece = (size_t)kib << 10; }       // (test hook: pieces an emulator can chew)
        // While the scheduler is still being created (warm_up_async), read ahead: whole pieces back to back into ONE ordinary buffer, at most 16 of them /
        // 1 GiB (reserved at once, touched as it fills).  They are compressed first by the helper thread -- a batch of pieces per call, so that every GPU
        // context takes its piece as in the loop below -- while this thread goes on reading into the pinned buffers.
        warm_up_async();
        struct EarlyBuf { uint8_t* p = nullptr; ~EarlyBuf() { free(p); } } early;
        size_t earlyBytes = 0; bool earlyEof = false;
        size_t earlyMost = ((size_t)1 << 30) / piece; if (earlyMost > 16u) earlyMost = 16u; earlyMost *= piece;
        while (shared() && shared()->multiState.load() == 0 && earlyBytes + piece <= earlyMost && !earlyEof) {
            if (!early.p && !(early.p = (uint8_t*)malloc(earlyMost))) break;
            size_t got = piece;
            const HRESULT r = read_full(in, early.p + earlyBytes, &got);
            if (r != S_OK) return r;
            if (got == 0) { earlyEof = earlyBytes != 0u; break; }             // (an empty input goes through the loop below, which codes the empty stream)
            earlyBytes += got;
            if (got < piece) earlyEof = true;
        }
        int mrc = GC_OK;
        gc_multi* const multi = shared_multi(&mrc);
        if (!multi) return hresult_of(mrc);
        // a batch = whole pieces, one per worker, but at most 1 GiB (and at least one piece): three pinned buffers of that size are all the
        // host memory a coder takes, whatever the number of GPUs
        size_t perBatch = (size_t)gc_multi_workers(multi), most = ((size_t)1 << 30) / piece;
        if (perBatch > most) perBatch = most;
        if (perBatch < 1) perBatch = 1;
        size_t batch = piece * perBatch;
        if (expected_ && expected_ < batch) batch = (size_t)((expected_ + 131071u) & ~(uint64_t)131071u);
        // two input buffers: while a helper thread compresses batch k on the GPUs and writes it out, this thread reads batch k + 1 (the host's
        // reader also computes its CRC there), so that the slower of the two sides sets the pace, not their sum
        BufLease lease; BufSet& B = lease.b;
        if (!buf_grow(&B.in[0], &B.inCap[0], batch, 0) || !buf_grow(&B.in[1], &B.inCap[1], batch, 0)) return E_OUTOFMEMORY;
        const size_t inCap = batch;
        const size_t earlyCall = earlyBytes < piece * perBatch ? earlyBytes : piece * perBatch;      // bytes of one call over the pieces read ahead
        if (!buf_grow(&B.out, &B.outCap, gc_codec_compress_bound(codec(), inCap > earlyCall ? inCap : earlyCall) + 1u, 0)) return E_OUTOFMEMORY;
        uint8_t* const outBuf = B.out; const size_t outCap = B.outCap;
        uint64_t totalIn = 0, totalOut = 0;
        struct Job { std::thread th; HRESULT hr = S_OK; size_t got = 0, produced = 0; bool active = false; } job;
        auto finish = [&]() -> HRESULT {                      // wait for the batch in flight, account for it
            if (!job.active) return S_OK;
            job.th.join(); job.active = false;
            if (job.hr != S_OK) return job.hr;
            totalIn += job.got; totalOut += job.produced;
            return progress ? progress->SetRatioInfo(&totalIn, &totalOut) : S_OK;
        };
        HRESULT res = S_OK;
        unsigned segIdx = 0;                                   // gc_multi calls handed out so far: the first one of a plain brotli stream carries the stream header
        auto flags_of = [&](unsigned idx) -> unsigned {       // FLZMA2: one end marker behind the last call; plain brotli: the closing meta-block behind the last
            return kind_ == KIND_FLZMA2 ? GC_FLZMA2_NO_END_MARK : plainBrotli_ ? (GC_BROTLI_PLAIN | GC_BROTLI_NOT_LAST | (idx != 0 ? GC_BROTLI_NOT_FIRST : 0u)) : 0u;
        };
        if (earlyBytes) {                                      // the pieces read ahead: calls of one batch each, in order, on the helper thread
            job.got = earlyBytes; job.produced = 0; job.hr = S_OK; job.active = true;
            const unsigned first = segIdx; segIdx += (unsigned)((earlyBytes + earlyCall - 1u) / earlyCall);
            const uint8_t* const ep = early.p;
            auto work = [this, multi, ep, earlyBytes, earlyCall, first, &flags_of, out, outBuf, outCap, &job]() {
                unsigned i = 0;
                for (size_t off = 0; off < earlyBytes && job.hr == S_OK; off += earlyCall, i++) {
                    const size_t n = earlyBytes - off < earlyCall ? earlyBytes - off : earlyCall;
                    size_t produced = 0; int rc;
                    { std::lock_guard<std::mutex> g(shared()->gpu); rc = gc_multi_compress_host(multi, codec(), ep + off, n, outBuf, outCap, level_, flags_of(first + i), 0, &produced); }
                    job.hr = rc != GC_OK ? hresult_of(rc) : write_all(out, outBuf, produced);
                    job.produced += produced;
                }
            };
            try { job.th = std::thread(work); }
            catch (...) { job.active = false; work(); if (job.hr != S_OK) return job.hr; totalIn += earlyBytes; totalOut += job.produced; }
        }
        for (; !earlyEof;) {
            uint8_t* const buf = B.in[segIdx & 1u];
            size_t got = inCap;
            HRESULT r = read_full(in, buf, &got);
            HRESULT f = finish();
            if (r != S_OK) { res = r; break; }
            if (f != S_OK) { res = f; break; }
            if (got == 0 && (segIdx != 0 || kind_ == KIND_FLZMA2)) break;
            const unsigned fl = flags_of(segIdx);
            if (plainBrotli_ && got == 0) break;
            segIdx++;
            job.got = got; job.produced = 0; job.hr = S_OK; job.active = true;
            auto work = [this, multi, buf, got, fl, out, outBuf, outCap, &job]() {
                int rc;
                { std::lock_guard<std::mutex> g(shared()->gpu); rc = gc_multi_compress_host(multi, codec(), buf, got, outBuf, outCap, level_, fl, 0, &job.produced); }
                job.hr = rc != GC_OK ? hresult_of(rc) : write_all(out, outBuf, job.produced);
            };
            try { job.th = std::thread(work); }
            catch (...) { job.active = false; work(); if (job.hr != S_OK) { res = job.hr; break; } totalIn += got; totalOut += job.produced; }   // no helper thread to be had: this one does the batch
            if (got < inCap) break;       // short read = end of stream
        }
        { const HRESULT f = finish(); if (res == S_OK) res = f; }
        if (res != S_OK) return res;
        if (plainBrotli_) {
            const uint8_t closeStream = totalIn != 0 ? 0x03 : 0x06;      // ISLAST + ISLASTEMPTY (an empty input: WBITS 16 + the same)
            HRESULT r = write_all(out, &closeStream, 1);
            if (r != S_OK) return r;
            totalOut += 1;
            if (progress) { r = progress->SetRatioInfo(&totalIn, &totalOut); if (r != S_OK) return r; }
        }
        if (kind_ == KIND_FLZMA2) {
            const uint8_t endMark = 0x00;
            HRESULT r = write_all(out, &endMark, 1);
            if (r != S_OK) return r;
            totalOut += 1;
            if (progress) { r = progress->SetRatioInfo(&totalIn, &totalOut); if (r != S_OK) return r; }
        }
        return S_OK;
    }
};

// ZSTD decoder: the input stream is read in large pieces; every piece is cut at the end of its last whole frame (gc_zstd_scan_prefix), those
// frames are decoded on the GPU (entropy stage per block, match copies per frame), the content is written out, the cut-off tail moves to the
// front of the buffer.  A frame larger than the buffer grows the buffer.  Errors follow ZstdDecoder.cpp:113-131: damaged data E_FAIL,
// unsupported frames (dictionary) E_NOTIMPL.
class CGpuZstdDecoder final : public ICompressCoder, public ICompressSetDecoderProperties2, public ICompressSetCoderMt {
    ULONG refs_ = 1;
    gc_zstd_frame* frames_ = nullptr; size_t framesCap_ = 0;

public:
    ~CGpuZstdDecoder() { free(frames_); }

    HRESULT QueryInterface(const GUID& iid, void** out) override
    {
        if (!out) return E_INVALIDARG;
        *out = nullptr;
        if (iid == IID_IUnknown || iid == IID_ICompressCoder) *out = static_cast<ICompressCoder*>(this);
        else if (iid == IID_ICompressSetDecoderProperties2) *out = static_cast<ICompressSetDecoderProperties2*>(this);
        else if (iid == IID_ICompressSetCoderMt) *out = static_cast<ICompressSetCoderMt*>(this);
        else return E_NOINTERFACE;
        ++refs_;
        return S_OK;
    }
    ULONG AddRef() override { return ++refs_; }
    ULONG Release() override { if (--refs_ != 0) return refs_; delete this; return 0; }

    HRESULT SetNumberOfThreads(uint32_t) override { return S_OK; }
    // 1 byte of flags; 3 or 5 bytes from older encoders (and from this module's encoder) are accepted as well -- ZstdDecoder.cpp:32-49
    HRESULT SetDecoderProperties2(const uint8_t*, uint32_t size) override { return (size == 1 || size == 3 || size == 5) ? S_OK : E_NOTIMPL; }

    HRESULT Code(ISequentialInStream* in, ISequentialOutStream* out, const uint64_t*, const uint64_t* outSize, ICompressProgressInfo* progress) override
    {
        if (!in || !out) return E_INVALIDARG;
        gc_ctx* const ctx = shared_dec_ctx();
        if (!ctx) return E_FAIL;                                                               // no gfx950 device: there is no CPU decoder behind this object
        BufLease lease;
        uint8_t*& inBuf_ = lease.b.in[0]; size_t& inCap_ = lease.b.inCap[0]; uint8_t*& outBuf_ = lease.b.out; size_t& outCap_ = lease.b.outCap;
        const size_t kPiece = (size_t)64 << 20;
        const size_t kMaxFrameIn = (size_t)1 << 30;          // largest compressed frame buffered whole (pinned host memory)
        const size_t kMaxContent = (size_t)4 << 30;          // largest content of one piece (pinned host + HBM, + 4 B per byte on the wide path)
        if (!buf_grow(&inBuf_, &inCap_, kPiece, 0)) return E_OUTOFMEMORY;
        uint64_t totalIn = 0, totalOut = 0;
        size_t have = 0;
        bool eof = false;
        for (;;) {
            while (!eof && have < inCap_) {
                const size_t want = inCap_ - have;
                size_t got = want;
                HRESULT r = read_full(in, inBuf_ + have, &got);
                if (r != S_OK) return r;
                have += got; totalIn += got;
                if (got < want) eof = true;
            }
            if (have == 0) break;
            size_t nFrames = 0, consumed = 0; uint64_t content = 0;
            int rc = gc_zstd_scan_prefix(inBuf_, have, nullptr, 0, &nFrames, nullptr, &consumed);
            if (rc != GC_OK) return rc == GC_ERR_PARAM ? E_NOTIMPL : E_FAIL;
            if (consumed == 0) {                                   // not even one whole frame in the buffer
                if (eof) return E_FAIL;                            // the stream ends inside a frame
                // One frame is the unit this decoder works on, and it holds the frame (pinned) and its content (pinned + HBM): the buffer grows
                // up to kMaxFrameIn and no further -- beyond it the object answers E_NOTIMPL ("unsupported") so that a host with another ZSTD
                // decoder takes the stream there (the reference's decoder needs only a window, ZstdDecoder.cpp:66-240); INTEGRATION.md states the bound.
                if (inCap_ >= kMaxFrameIn) return E_NOTIMPL;
                if (!buf_grow(&inBuf_, &inCap_, inCap_ * 2u, have)) return E_OUTOFMEMORY;
                continue;
            }
            if (nFrames > framesCap_) {
                free(frames_); framesCap_ = 0;
                frames_ = (gc_zstd_frame*)malloc((nFrames + 64u) * sizeof(gc_zstd_frame));
                if (!frames_) return E_OUTOFMEMORY;
                framesCap_ = nFrames + 64u;
            }
            if (nFrames) {
                rc = gc_zstd_scan_prefix(inBuf_, have, frames_, framesCap_, &nFrames, &content, &consumed);
                if (rc != GC_OK) return E_FAIL;
                // capacity: the content sizes the frames state; a frame that does not state one regenerates at most 128 KiB per block
                // (zstd_decompress_block.c: blockSizeMax), and the scan has counted its blocks -- no guessing, no second decode
                // (a piece whose frames regenerate more than kMaxContent is decoded a run of frames at a time; one frame beyond it is refused)
                size_t cap = 0;
                for (size_t i = 0; i < nFrames; i++) {
                    const uint64_t most = (uint64_t)frames_[i].n_blocks * (128u << 10);
                    if ((frames_[i].flags & 2u) && frames_[i].content_size > most) return E_FAIL;      // a stated size the frame's blocks cannot regenerate: damaged (or forged to make this object pin gigabytes)
                    const uint64_t fc = (frames_[i].flags & 2u) ? frames_[i].content_size : most;
                    if (fc > kMaxContent) return E_NOTIMPL;
                    if (cap + fc > kMaxContent) { consumed = (size_t)frames_[i].src_off; break; }
                    cap += (size_t)fc;
                }
                size_t produced = 0;
                if (!buf_grow(&outBuf_, &outCap_, cap ? cap : 1u, 0)) return E_OUTOFMEMORY;
                { std::lock_guard<std::mutex> g(shared()->gpu); rc = gc_zstd_decompress_host(ctx, inBuf_, consumed, outBuf_, cap, &produced); }
                if (rc != GC_OK) return rc == GC_ERR_PARAM ? E_NOTIMPL : hresult_of(rc);
                HRESULT r = write_all(out, outBuf_, produced);
                if (r != S_OK) return r;
                totalOut += produced;
            }
            have -= consumed;
            if (have) memmove(inBuf_, inBuf_ + consumed, have);
            if (progress) {
                const uint64_t pin = totalIn - have;
                HRESULT r = progress->SetRatioInfo(&pin, &totalOut);
                if (r != S_OK) return r;
            }
            if (eof && have == 0) break;
        }
        return S_OK;
    }
};

// BROTLI decoder (NCompress::NBROTLI::CDecoder, BrotliDecoder.cpp:124; brotli-mt_decompress.c:191-288): the input is read in large pieces, every piece is cut at the end of its
// last whole brotli-mt frame (gc_brotli_scan_prefix), the chunks are decoded on the GPU (one wave per chunk), the content is written out, the tail moves to the front.
// An input that does not start with a brotli-mt header is a bare RFC 7932 stream (what the reference writes with "0 threads", brotli-mt_decompress.c:573): one chunk, read whole;
// the host must have told its size (SetOutStreamSize / Code's outSize), else E_NOTIMPL.
// The static dictionary of RFC 7932 is not in this module: the first decoder takes it from the host process -- a host built from the reference tree carries the brotli library,
// whose BrotliGetDictionary (C/brotli/common/dictionary.h) is looked up among the loaded objects -- or from the file named by GPUCODEC_BROTLI_DICTIONARY (the 122 784 bytes).
// Without it a stream that refers to the dictionary ends with E_NOTIMPL ("unsupported"), as the ZSTD decoder does with dictionary frames.
void brotli_dictionary_once()
{
    static std::once_flag once;
    std::call_once(once, [] {
        if (gc_brotli_dec_has_dictionary()) return;
        struct Found { const void* fn; } found = { nullptr };
        dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* arg) -> int {
            void* h = dlopen(info->dlpi_name && info->dlpi_name[0] ? info->dlpi_name : nullptr, RTLD_NOLOAD | RTLD_LAZY);
            if (!h) return 0;
            void* f = dlsym(h, "BrotliGetDictionary");
            dlclose(h);
            if (f) { ((Found*)arg)->fn = f; return 1; }
            return 0;
        }, &found);
        if (found.fn) {
            // struct BrotliDictionary { uint8_t size_bits_by_length[32]; uint32_t offsets_by_length[32]; size_t data_size; const uint8_t* data; ... } (common/dictionary.h:18-62);
            // gc_brotli_dec_set_dictionary checks size and the CRC-32 the RFC states, so another layout cannot get wrong bytes in
            struct Head { uint8_t bits[32]; uint32_t offs[32]; size_t dataSize; const uint8_t* data; };
            const Head* d = ((const Head* (*)(void))found.fn)();
            if (d && d->data && d->dataSize == 122784u && gc_brotli_dec_set_dictionary(d->data, d->dataSize) == GC_OK) return;
        }
        if (const char* path = getenv("GPUCODEC_BROTLI_DICTIONARY")) {
            if (FILE* f = fopen(path, "rb")) {
                std::vector<uint8_t> b(122784u + 1u);
                const size_t n = fread(b.data(), 1, b.size(), f);
                fclose(f);
                gc_brotli_dec_set_dictionary(b.data(), n);
            }
        }
    });
}

class CGpuBrotliDecoder final : public ICompressCoder, public ICompressSetDecoderProperties2, public ICompressSetCoderMt {
    ULONG refs_ = 1;
    gc_brotli_chunk* chunks_ = nullptr; size_t chunksCap_ = 0;

public:
    ~CGpuBrotliDecoder() { free(chunks_); }

    HRESULT QueryInterface(const GUID& iid, void** out) override
    {
        if (!out) return E_INVALIDARG;
        *out = nullptr;
        if (iid == IID_IUnknown || iid == IID_ICompressCoder) *out = static_cast<ICompressCoder*>(this);
        else if (iid == IID_ICompressSetDecoderProperties2) *out = static_cast<ICompressSetDecoderProperties2*>(this);
        else if (iid == IID_ICompressSetCoderMt) *out = static_cast<ICompressSetCoderMt*>(this);
        else return E_NOINTERFACE;
        ++refs_;
        return S_OK;
    }
    ULONG AddRef() override { return ++refs_; }
    ULONG Release() override { if (--refs_ != 0) return refs_; delete this; return 0; }

    HRESULT SetNumberOfThreads(uint32_t) override { return S_OK; }
    HRESULT SetDecoderProperties2(const uint8_t*, uint32_t size) override { return size == 3 ? S_OK : E_NOTIMPL; }     // {major, minor, level}: BrotliDecoder.cpp:86-96

    HRESULT Code(ISequentialInStream* in, ISequentialOutStream* out, const uint64_t*, const uint64_t* outSize, ICompressProgressInfo* progress) override
    {
        if (!in || !out) return E_INVALIDARG;
        gc_ctx* const ctx = shared_dec_ctx();
        if (!ctx) return E_FAIL;                                                               // no gfx950 device: there is no CPU decoder behind this object
        brotli_dictionary_once();
        BufLease lease;
        uint8_t*& inBuf_ = lease.b.in[0]; size_t& inCap_ = lease.b.inCap[0]; uint8_t*& outBuf_ = lease.b.out; size_t& outCap_ = lease.b.outCap;
        const size_t kPiece = (size_t)64 << 20;
        const size_t kMaxIn = (size_t)1 << 30;               // largest compressed frame (or bare stream) buffered whole
        const size_t kMaxContent = (size_t)4 << 30;          // largest content of one piece
        if (!buf_grow(&inBuf_, &inCap_, kPiece, 0)) return E_OUTOFMEMORY;
        uint64_t totalIn = 0, totalOut = 0;
        size_t have = 0;
        bool eof = false, bare = false, first = true;
        for (;;) {
            while (!eof && have < inCap_) {
                const size_t want = inCap_ - have;
                size_t got = want;
                HRESULT r = read_full(in, inBuf_ + have, &got);
                if (r != S_OK) return r;
                have += got; totalIn += got;
                if (got < want) eof = true;
            }
            if (have == 0) break;
            if (first && have >= 4) { uint32_t magic; memcpy(&magic, inBuf_, 4); bare = magic != 0x184D2A50u; }
            first = false;
            size_t consumed = 0, produced = 0;
            int rc;
            if (bare) {
                // a bare stream is one chunk: all of it has to be here, and the host has to have said how much it holds
                if (!eof) { if (inCap_ >= kMaxIn) return E_NOTIMPL; if (!buf_grow(&inBuf_, &inCap_, inCap_ * 2u, have)) return E_OUTOFMEMORY; continue; }
                if (!outSize || *outSize > 0xFFFF0000ull) return E_NOTIMPL;
                const size_t cap = (size_t)*outSize;
                if (!buf_grow(&outBuf_, &outCap_, cap ? cap : 1u, 0)) return E_OUTOFMEMORY;
                { std::lock_guard<std::mutex> g(shared()->gpu); rc = gc_brotli_decompress_host(ctx, inBuf_, have, outBuf_, cap, &produced); }
                consumed = have;
            } else {
                size_t nChunks = 0;
                rc = gc_brotli_scan_prefix(inBuf_, have, nullptr, 0, &nChunks, nullptr, &consumed);
                if (rc != GC_OK) return E_FAIL;
                if (consumed == 0) {                               // not even one whole frame in the buffer
                    if (eof) return E_FAIL;                        // the stream ends inside a frame
                    if (inCap_ >= kMaxIn) return E_NOTIMPL;
                    if (!buf_grow(&inBuf_, &inCap_, inCap_ * 2u, have)) return E_OUTOFMEMORY;
                    continue;
                }
                if (nChunks > chunksCap_) {
                    free(chunks_); chunksCap_ = 0;
                    chunks_ = (gc_brotli_chunk*)malloc((nChunks + 64u) * sizeof(gc_brotli_chunk));
                    if (!chunks_) return E_OUTOFMEMORY;
                    chunksCap_ = nChunks + 64u;
                }
                uint64_t capTotal = 0;
                rc = gc_brotli_scan_prefix(inBuf_, have, chunks_, chunksCap_, &nChunks, &capTotal, &consumed);
                if (rc != GC_OK) return E_FAIL;
                // the hints bound the content (brotli-mt_decompress.c:243); a piece whose chunks may regenerate more than kMaxContent is decoded a run of chunks at a time
                size_t cap = 0;
                for (size_t i = 0; i < nChunks; i++) {
                    if (cap + chunks_[i].capacity > kMaxContent) { if (i == 0) return E_NOTIMPL; consumed = (size_t)chunks_[i].src_off - 16u; break; }
                    cap += chunks_[i].capacity;
                }
                if (!buf_grow(&outBuf_, &outCap_, cap ? cap : 1u, 0)) return E_OUTOFMEMORY;
                { std::lock_guard<std::mutex> g(shared()->gpu); rc = gc_brotli_decompress_host(ctx, inBuf_, consumed, outBuf_, cap, &produced); }
            }
            if (rc != GC_OK) return rc == GC_ERR_UNSUPPORTED ? E_NOTIMPL : hresult_of(rc);
            HRESULT r = write_all(out, outBuf_, produced);
            if (r != S_OK) return r;
            totalOut += produced;
            have -= consumed;
            if (have) memmove(inBuf_, inBuf_ + consumed, have);
            if (progress) {
                const uint64_t pin = totalIn - have;
                r = progress->SetRatioInfo(&pin, &totalOut);
                if (r != S_OK) return r;
            }
            if (eof && have == 0) break;
        }
        return S_OK;
    }
};

// A pre-filter object: NCompress::NBranch::CCoder / CEncoder / CDecoder (BranchMisc.cpp:14-118), NCompress::NBcj::CCoder2 (BcjCoder.cpp:10-22) and
// NCompress::NDelta::CEncoder / CDecoder (DeltaFilter.cpp:28-119) over gc_filter_host: every Filter() call takes the host's buffer to the device, converts it
// there and brings it back; the program counter, the x86 converter's state word and the Delta filter's 256 bytes of history are carried from call to call.
class CGpuFilter final : public ICompressFilter, public ICompressSetCoderProperties, public ICompressWriteCoderProperties, public ICompressSetDecoderProperties2 {
    ULONG refs_ = 1;
    const int kind_; const bool encoding_; const unsigned align_;
    uint32_t pc_ = 0, pcInit_ = 0;
    unsigned delta_ = 1;
    unsigned char state_[256];
    bool hasPcProp() const { return kind_ == GC_BRA_ARM64 || kind_ == GC_BRA_RISCV; }
public:
    CGpuFilter(int kind, bool encoding, unsigned align) : kind_(kind), encoding_(encoding), align_(align) { memset(state_, 0, sizeof(state_)); }
    HRESULT QueryInterface(const GUID& iid, void** out) override
    {
        if (!out) return E_INVALIDARG;
        *out = nullptr;
        if (iid == IID_IUnknown || iid == IID_ICompressFilter) *out = static_cast<ICompressFilter*>(this);
        else if (encoding_ && (hasPcProp() || kind_ == GC_FILTER_DELTA) && iid == IID_ICompressSetCoderProperties) *out = static_cast<ICompressSetCoderProperties*>(this);
        else if (encoding_ && (hasPcProp() || kind_ == GC_FILTER_DELTA) && iid == IID_ICompressWriteCoderProperties) *out = static_cast<ICompressWriteCoderProperties*>(this);
        else if (!encoding_ && (hasPcProp() || kind_ == GC_FILTER_DELTA) && iid == IID_ICompressSetDecoderProperties2) *out = static_cast<ICompressSetDecoderProperties2*>(this);
        else return E_NOINTERFACE;
        ++refs_;
        return S_OK;
    }
    ULONG AddRef() override { return ++refs_; }
    ULONG Release() override { if (--refs_ != 0) return refs_; delete this; return 0; }

    // A return of 0 means "not enough data" to CFilterCoder, which then writes the bytes through UNFILTERED and reports S_OK (FilterCoder.cpp:172-174,
    // :383-387): a failure must never look like that.  Init() is the call with an HRESULT (CFilterCoder RINOKs it): no device -> E_FAIL before a byte is
    // read.  A failure in mid-stream is latched and answered with a size no buffer has: the host takes a return above what it handed over as "cannot
    // convert this" and ends with E_FAIL / S_FALSE (FilterCoder.cpp:176-201, :248-262, :389-396), never with an archive that claims the filter.
    bool failed_ = false;
    static const uint32_t kFilterFailed = 0xFFFFFFFFu;
    HRESULT Init() override
    {
        pc_ = pcInit_; memset(state_, 0, sizeof(state_)); failed_ = false;     // Z7_BRANCH_CONV_ST_X86_STATE_INIT_VAL = 0, Delta_Init: zeros
        if (test_env_u32("GC_PLUGIN_FILTER_NO_DEVICE")) return E_FAIL;          // (test hook: as on a machine without a usable GPU)
        return shared_dec_ctx() ? S_OK : E_FAIL;
    }
    uint32_t Filter(uint8_t* data, uint32_t size) override
    {
        if (!size) return 0;
        gc_ctx* const ctx = failed_ ? nullptr : shared_dec_ctx();
        if (!ctx) { failed_ = true; return kFilterFailed; }
        const unsigned failAt = test_env_u32("GC_PLUGIN_FILTER_FAIL_AT_PC");     // (test hook: the device fails once the stream has got this far)
        size_t done = 0; int rc;
        { std::lock_guard<std::mutex> g(shared()->gpu); rc = (failAt && pc_ - pcInit_ >= failAt) ? GC_ERR_HIP : gc_filter_host(ctx, kind_, data, size, pc_, encoding_ ? 1 : 0, delta_, state_, &done); }
        if (rc != GC_OK) { failed_ = true; return kFilterFailed; }
        pc_ += (uint32_t)done;
        return (uint32_t)done;
    }
    HRESULT SetCoderProperties(const PROPID* ids, const PROPVARIANT* props, uint32_t n) override
    {
        uint32_t pc = 0; unsigned delta = delta_;
        for (uint32_t i = 0; i < n; i++) {
            if (kind_ == GC_FILTER_DELTA) {                        // DeltaFilter.cpp:53-80
                if (ids[i] >= NCoderPropID::kReduceSize) continue;
                if (props[i].vt != VT_UI4) return E_INVALIDARG;
                if (ids[i] == NCoderPropID::kDefaultProp) { if (props[i].ulVal < 1 || props[i].ulVal > 256) return E_INVALIDARG; delta = props[i].ulVal; }
                else if (ids[i] != NCoderPropID::kNumThreads && ids[i] != NCoderPropID::kLevel) return E_INVALIDARG;
            } else if (ids[i] == NCoderPropID::kDefaultProp || ids[i] == NCoderPropID::kBranchOffset) {      // BranchMisc.cpp:44-63
                if (props[i].vt != VT_UI4) return E_INVALIDARG;
                pc = props[i].ulVal;
                if (pc & align_) return E_INVALIDARG;
            }
        }
        delta_ = delta; pcInit_ = pc;
        return S_OK;
    }
    HRESULT WriteCoderProperties(ISequentialOutStream* out) override
    {
        if (kind_ == GC_FILTER_DELTA) { const uint8_t p = (uint8_t)(delta_ - 1u); return write_all(out, &p, 1); }       // DeltaFilter.cpp:82-86
        if (pcInit_ == 0) return S_OK;                             // BranchMisc.cpp:66-73
        const uint8_t b[4] = { (uint8_t)pcInit_, (uint8_t)(pcInit_ >> 8), (uint8_t)(pcInit_ >> 16), (uint8_t)(pcInit_ >> 24) };
        return write_all(out, b, 4);
    }
    HRESULT SetDecoderProperties2(const uint8_t* p, uint32_t size) override
    {
        if (kind_ == GC_FILTER_DELTA) { if (size != 1) return E_INVALIDARG; delta_ = (unsigned)p[0] + 1u; return S_OK; }       // DeltaFilter.cpp:112-118
        uint32_t v = 0;                                            // BranchMisc.cpp:103-116
        if (size != 0) { if (size != 4) return E_NOTIMPL; v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); if (v & align_) return E_NOTIMPL; }
        pcInit_ = v;
        return S_OK;
    }
};

HRESULT create_filter(uint32_t index, bool encoding, const GUID* iid, void** out)
{
    if (!iid || !(*iid == IID_ICompressFilter)) return E_NOINTERFACE;          // CodecExports.cpp:127-150: filters are created through ICompressFilter
    CGpuFilter* f = new (std::nothrow) CGpuFilter(kMethods[index].filter, encoding, kMethods[index].align);
    IUnknown* obj = f ? static_cast<ICompressFilter*>(f) : nullptr;
    if (!obj) return E_OUTOFMEMORY;
    *out = obj;
    return S_OK;
}

HRESULT create_decoder(uint32_t index, const GUID* iid, void** out)
{
    if (!out) return E_INVALIDARG;
    *out = nullptr;
    if (index < kNumMethods && kMethods[index].kind == KIND_FILTER) return create_filter(index, false, iid, out);
    if (index >= kNumMethods || (kMethods[index].kind != KIND_ZSTD && kMethods[index].kind != KIND_BROTLI)) return CLASS_E_CLASSNOTAVAILABLE;     // (FLZMA2: the host's LZMA2 decoder)
    if (!iid || !(*iid == IID_ICompressCoder)) return E_NOINTERFACE;
    IUnknown* obj = nullptr;
    if (kMethods[index].kind == KIND_BROTLI) { CGpuBrotliDecoder* d = new (std::nothrow) CGpuBrotliDecoder(); obj = d ? static_cast<ICompressCoder*>(d) : nullptr; }
    else { CGpuZstdDecoder* d = new (std::nothrow) CGpuZstdDecoder(); obj = d ? static_cast<ICompressCoder*>(d) : nullptr; }
    if (!obj) return E_OUTOFMEMORY;
    *out = obj;
    return S_OK;
}

HRESULT create_encoder(uint32_t index, const GUID* iid, void** out)
{
    if (!out) return E_INVALIDARG;
    *out = nullptr;
    if (index >= kNumMethods) return CLASS_E_CLASSNOTAVAILABLE;
    if (kMethods[index].kind == KIND_FILTER) return create_filter(index, true, iid, out);
    if (!iid || !(*iid == IID_ICompressCoder)) return E_NOINTERFACE;     // the 1-stream codecs (CodecExports.cpp:127-150)
    warm_up_async();                                       // the devices are opened while the host sets the coder up and reads its first input
    CGpuEncoder* e = new (std::nothrow) CGpuEncoder(kMethods[index].kind);
    IUnknown* obj = e ? static_cast<ICompressCoder*>(e) : nullptr;
    if (!obj) return E_OUTOFMEMORY;
    *out = obj;          // pointer to the ICompressCoder sub-object, reference count 1 (RegisterCodec.h:25-26)
    return S_OK;
}

}  // namespace

GC_EXPORT HRESULT GetNumberOfMethods(uint32_t* n) { if (!n) return E_INVALIDARG; *n = kNumMethods; return S_OK; }

GC_EXPORT HRESULT GetMethodProperty(uint32_t index, PROPID propID, PROPVARIANT* value)
{
    if (!value) return E_INVALIDARG;
    gc_variant_clear(value);
    if (index >= kNumMethods) return E_INVALIDARG;
    const MethodInfo& m = kMethods[index];
    switch (propID) {
        case NMethodPropID::kID: value->vt = VT_UI8; value->uhVal = m.id; break;
        case NMethodPropID::kName: value->bstrVal = gc_bstr_ascii(m.name); if (!value->bstrVal) return E_OUTOFMEMORY; value->vt = VT_BSTR; break;
        case NMethodPropID::kEncoder: {
            GUID g = gc_codec_clsid(m.id, true);
            value->bstrVal = gc_bstr_bytes(&g, sizeof(g)); if (!value->bstrVal) return E_OUTOFMEMORY; value->vt = VT_BSTR; break;
        }
        case NMethodPropID::kEncoderIsAssigned: value->vt = VT_BOOL; value->boolVal = -1; break;   // VARIANT_TRUE
        case NMethodPropID::kDecoder:
            if (m.kind == KIND_ZSTD || m.kind == KIND_BROTLI || m.kind == KIND_FILTER) {
                GUID g = gc_codec_clsid(m.id, false);
                value->bstrVal = gc_bstr_bytes(&g, sizeof(g)); if (!value->bstrVal) return E_OUTOFMEMORY; value->vt = VT_BSTR;
            }
            break;
        case NMethodPropID::kDecoderIsAssigned: value->vt = VT_BOOL; value->boolVal = (m.kind == KIND_ZSTD || m.kind == KIND_BROTLI || m.kind == KIND_FILTER) ? -1 : 0; break;
        case NMethodPropID::kIsFilter: value->vt = VT_BOOL; value->boolVal = m.kind == KIND_FILTER ? -1 : 0; break;
        default: break;      // kPackStreams, ...: left VT_EMPTY
    }
    return S_OK;
}

GC_EXPORT HRESULT CreateEncoder(uint32_t index, const GUID* iid, void** out) { return create_encoder(index, iid, out); }

GC_EXPORT HRESULT CreateDecoder(uint32_t index, const GUID* iid, void** out) { return create_decoder(index, iid, out); }

GC_EXPORT HRESULT CreateObject(const GUID* clsid, const GUID* iid, void** out)
{
    if (!out) return E_INVALIDARG;
    *out = nullptr;
    if (!clsid) return E_INVALIDARG;
    for (uint32_t i = 0; i < kNumMethods; i++)
        if (*clsid == gc_codec_clsid(kMethods[i].id, true)) return create_encoder(i, iid, out);
    for (uint32_t i = 0; i < kNumMethods; i++)
        if ((kMethods[i].kind == KIND_ZSTD || kMethods[i].kind == KIND_BROTLI || kMethods[i].kind == KIND_FILTER) && *clsid == gc_codec_clsid(kMethods[i].id, false)) return create_decoder(i, iid, out);
    return CLASS_E_CLASSNOTAVAILABLE;
}

GC_EXPORT HRESULT GetModuleProp(PROPID propID, PROPVARIANT* value)
{
    if (!value) return E_INVALIDARG;
    gc_variant_clear(value);
    switch (propID) {
        case NModulePropID::kInterfaceType: value->vt = VT_UI4; value->ulVal = 0; break;            // no virtual destructor in IUnknown
        case NModulePropID::kVersion: value->vt = VT_UI4; value->ulVal = (26u << 16) + 1u; break;   // host ABI generation (CodecExports.cpp:372)
        default: break;
    }
    return S_OK;
}
