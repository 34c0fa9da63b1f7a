// tests/host/plugin_host.cpp -- TEST INFRASTRUCTURE: a minimal stand-in for the 7-Zip host's codec loader.
// Does to a codec module exactly what CPP/7zip/UI/Common/LoadCodecs.cpp:531-650 and CPP/7zip/Common/CreateCoder.cpp:160-232
// do: dlopen, GetModuleProp(kInterfaceType), enumerate GetNumberOfMethods/GetMethodProperty, CreateEncoder by index (or
// CreateObject by class id), QueryInterface for the property interfaces (7zEncode.cpp:186-197,298-304), Code() over
// pull/push streams that deliberately return short reads / short writes.
//
//   plugin_host <module.so> list
//   plugin_host <module.so> encode <method-name> <level> <in-file> <out-file> [props-out-file] [by-clsid]
//   plugin_host <module.so> decode <method-name> <props-file|-> <in-file> <out-file> [by-clsid]     (what 7zDecode.cpp:260-420 does with a decoder)
//   plugin_host <module.so> filter <method-name> <enc|dec> <prop|props-file|-> <in-file> <out-file> [buffer-bytes]
//                 (what CFilterCoder does with a pre-filter, CPP/7zip/Common/FilterCoder.cpp: Init, then Filter() on a buffer that is refilled behind the bytes left over;
//                  enc: <prop> = the number given as kDefaultProp / kBranchOffset, properties written to <out-file>.props; dec: <props-file> as written by enc)
#include "../../7-zip-zstd_amd/plugin/gc_7z_abi.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

typedef HRESULT (*Fn_GetNumberOfMethods)(uint32_t*);
typedef HRESULT (*Fn_GetMethodProperty)(uint32_t, PROPID, PROPVARIANT*);
typedef HRESULT (*Fn_CreateCoder)(uint32_t, const GUID*, void**);
typedef HRESULT (*Fn_CreateObject)(const GUID*, const GUID*, void**);
typedef HRESULT (*Fn_GetModuleProp)(PROPID, PROPVARIANT*);

struct RefCounted { ULONG refs = 1; };
struct FileIn final : ISequentialInStream, RefCounted {
    FILE* f; unsigned tick = 0;
    HRESULT QueryInterface(const GUID& iid, void** o) override { if (iid == IID_IUnknown || iid == IID_ISequentialInStream) { *o = this; refs++; return S_OK; } *o = nullptr; return E_NOINTERFACE; }
    ULONG AddRef() override { return ++refs; }
    ULONG Release() override { return --refs; }
    HRESULT Read(void* d, uint32_t size, uint32_t* done) override {
        // partial reads are legal (IStream.h:22-49): serve odd-sized pieces
        uint32_t cap = (tick++ % 3 == 0) ? 1000003u : 65536u * 7u + 13u;
        if (size > cap) size = cap;
        size_t n = fread(d, 1, size, f);
        if (done) *done = (uint32_t)n;
        return S_OK;
    }
};
struct FileOut final : ISequentialOutStream, RefCounted {
    FILE* f; unsigned tick = 0; uint64_t total = 0;
    HRESULT QueryInterface(const GUID& iid, void** o) override { if (iid == IID_IUnknown || iid == IID_ISequentialOutStream) { *o = this; refs++; return S_OK; } *o = nullptr; return E_NOINTERFACE; }
    ULONG AddRef() override { return ++refs; }
    ULONG Release() override { return --refs; }
    HRESULT Write(const void* d, uint32_t size, uint32_t* done) override {
        uint32_t cap = (tick++ % 2 == 0) ? 400001u : 1u << 20;      // partial writes are legal too
        if (size > cap) size = cap;
        size_t n = fwrite(d, 1, size, f);
        total += n;
        if (done) *done = (uint32_t)n;
        return n == size ? S_OK : E_FAIL;
    }
};
struct Progress final : ICompressProgressInfo, RefCounted {
    uint64_t in = 0, out = 0; unsigned calls = 0;
    HRESULT QueryInterface(const GUID& iid, void** o) override { if (iid == IID_IUnknown || iid == IID_ICompressProgressInfo) { *o = this; refs++; return S_OK; } *o = nullptr; return E_NOINTERFACE; }
    ULONG AddRef() override { return ++refs; }
    ULONG Release() override { return --refs; }
    HRESULT SetRatioInfo(const uint64_t* i, const uint64_t* o) override { if (i) in = *i; if (o) out = *o; calls++; return S_OK; }
};

static std::string narrow(BSTR b) { std::string s; for (const wchar_t* p = b; *p; p++) s += (char)*p; return s; }

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    auto getNum = (Fn_GetNumberOfMethods)dlsym(h, "GetNumberOfMethods");
    auto getProp = (Fn_GetMethodProperty)dlsym(h, "GetMethodProperty");
    auto createEnc = (Fn_CreateCoder)dlsym(h, "CreateEncoder");
    auto createDec = (Fn_CreateCoder)dlsym(h, "CreateDecoder");
    auto createObj = (Fn_CreateObject)dlsym(h, "CreateObject");
    auto modProp = (Fn_GetModuleProp)dlsym(h, "GetModuleProp");
    if (!getNum || !getProp || !createEnc || !createDec || !createObj || !modProp) { fprintf(stderr, "missing export\n"); return 4; }
    PROPVARIANT v; memset(&v, 0, sizeof(v));
    if (modProp(NModulePropID::kInterfaceType, &v) != S_OK || v.vt != VT_UI4 || v.ulVal != 0) { fprintf(stderr, "module is not compatible (interface type)\n"); return 5; }
    uint32_t n = 0;
    if (getNum(&n) != S_OK) return 6;
    std::string mode = argv[2];
    int found = -1; uint64_t foundId = 0;
    for (uint32_t i = 0; i < n; i++) {
        uint64_t id = 0; std::string name; bool enc = false, dec = false; GUID encId; memset(&encId, 0, sizeof(encId));
        memset(&v, 0, sizeof(v)); if (getProp(i, NMethodPropID::kID, &v) == S_OK && v.vt == VT_UI8) id = v.uhVal;
        memset(&v, 0, sizeof(v)); if (getProp(i, NMethodPropID::kName, &v) == S_OK && v.vt == VT_BSTR) { name = narrow(v.bstrVal); gc_variant_clear(&v); }
        memset(&v, 0, sizeof(v)); if (getProp(i, NMethodPropID::kEncoderIsAssigned, &v) == S_OK && v.vt == VT_BOOL) enc = v.boolVal != 0;
        memset(&v, 0, sizeof(v)); if (getProp(i, NMethodPropID::kDecoderIsAssigned, &v) == S_OK && v.vt == VT_BOOL) dec = v.boolVal != 0;
        memset(&v, 0, sizeof(v)); if (getProp(i, NMethodPropID::kEncoder, &v) == S_OK && v.vt == VT_BSTR) {
            uint32_t len; memcpy(&len, (uint8_t*)v.bstrVal - 4, 4);
            if (len != 16) { fprintf(stderr, "encoder class id has %u bytes\n", len); return 7; }
            memcpy(&encId, v.bstrVal, 16); gc_variant_clear(&v);
        }
        if (mode == "list") printf("%u %llX %s enc=%d dec=%d clsid=%08X-%04X-%04X\n", i, (unsigned long long)id, name.c_str(), enc, dec, encId.Data1, encId.Data2, encId.Data3);
        const bool wantDec = mode == "decode" || (mode == "filter" && argc > 4 && std::string(argv[4]) == "dec");
        if (argc > 3 && name == argv[3] && (wantDec ? dec : enc)) { found = (int)i; foundId = id; }
    }
    if (mode == "list") return 0;
    if (mode == "filter") {
        if (argc < 8 || found < 0) { fprintf(stderr, "method not found\n"); return 8; }
        const bool encoding = std::string(argv[4]) == "enc";
        memset(&v, 0, sizeof(v));
        if (getProp((uint32_t)found, NMethodPropID::kIsFilter, &v) != S_OK || v.vt != VT_BOOL || v.boolVal == 0) { fprintf(stderr, "not a filter\n"); return 9; }
        void* rawF = nullptr;
        { void* bad = nullptr; if ((encoding ? createEnc : createDec)((uint32_t)found, &IID_ICompressCoder, &bad) != E_NOINTERFACE || bad) { fprintf(stderr, "iid check\n"); return 10; } }
        HRESULT rf = (encoding ? createEnc : createDec)((uint32_t)found, &IID_ICompressFilter, &rawF);
        if (rf != S_OK || !rawF) { fprintf(stderr, "Create filter failed: %08X\n", (unsigned)rf); return 9; }
        ICompressFilter* flt = (ICompressFilter*)rawF;
        const std::string prop = argv[5];
        if (encoding) {
            ICompressSetCoderProperties* sp = nullptr; ICompressWriteCoderProperties* wp = nullptr;
            const bool hasProps = flt->QueryInterface(IID_ICompressSetCoderProperties, (void**)&sp) == S_OK;
            if (prop != "-") {
                if (!hasProps) { fprintf(stderr, "filter takes no properties\n"); return 11; }
                PROPID id = foundId == 3 ? (PROPID)NCoderPropID::kDefaultProp : (PROPID)NCoderPropID::kBranchOffset; PROPVARIANT pv; memset(&pv, 0, sizeof(pv)); pv.vt = VT_UI4; pv.ulVal = (uint32_t)strtoul(prop.c_str(), nullptr, 0);
                if (sp->SetCoderProperties(&id, &pv, 1) != S_OK) { fprintf(stderr, "SetCoderProperties refused\n"); return 12; }
            }
            if (hasProps) {
                if (flt->QueryInterface(IID_ICompressWriteCoderProperties, (void**)&wp) != S_OK) return 11;
                FileOut p; p.f = fopen((std::string(argv[7]) + ".props").c_str(), "wb"); if (!p.f || wp->WriteCoderProperties(&p) != S_OK) return 14; fclose(p.f);
                wp->Release(); sp->Release();
            }
        } else if (prop != "-") {
            ICompressSetDecoderProperties2* sp = nullptr;
            if (flt->QueryInterface(IID_ICompressSetDecoderProperties2, (void**)&sp) != S_OK) return 11;
            FILE* pf = fopen(prop.c_str(), "rb"); if (!pf) return 13;
            uint8_t pb[16]; size_t pn = fread(pb, 1, sizeof(pb), pf); fclose(pf);
            if (sp->SetDecoderProperties2(pb, (uint32_t)pn) != S_OK) { fprintf(stderr, "SetDecoderProperties2 refused %zu bytes\n", pn); return 12; }
            sp->Release();
        }
        FILE* fi = fopen(argv[6], "rb"); FILE* fo = fopen(argv[7], "wb");
        if (!fi || !fo) { fprintf(stderr, "file open\n"); return 13; }
        const size_t cap = argc > 8 ? (size_t)strtoul(argv[8], nullptr, 0) : (size_t)1 << 20;
        std::vector<uint8_t> buf(cap);
        if (flt->Init() != S_OK) return 15;
        size_t have = 0; bool eof = false; unsigned long long total = 0, calls = 0;
        for (;;) {
            while (!eof && have < cap) { const size_t got = fread(buf.data() + have, 1, cap - have, fi); if (got == 0) eof = true; have += got; }
            if (have == 0) break;
            uint32_t done = flt->Filter(buf.data(), (uint32_t)have); calls++;
            if (done > have) { fprintf(stderr, "filter asked for more bytes than a branch converter may\n"); return 16; }
            if (done == 0) {                                  // nothing more can be converted: the rest passes as it is (FilterCoder.cpp, end of stream)
                if (!eof && have < cap) continue;
                done = (uint32_t)have;
                if (!eof) { fprintf(stderr, "no progress on a full buffer\n"); return 16; }
            }
            if (fwrite(buf.data(), 1, done, fo) != done) return 14;
            total += done; have -= done;
            if (have) memmove(buf.data(), buf.data() + done, have);
        }
        fclose(fi); fclose(fo);
        if (flt->Release() != 0) { fprintf(stderr, "refcount leak\n"); return 18; }
        printf("ok bytes=%llu calls=%llu\n", total, calls);
        return 0;
    }
    if (mode == "decode") {
        if (argc < 7 || found < 0) { fprintf(stderr, "method not found\n"); return 8; }
        bool byClsidD = false; unsigned long long statedSize = 0; bool haveSize = false;      // [by-clsid] [size=<bytes>]: the unpack size a 7z folder knows (7zDecode.cpp hands it to Code())
        for (int i = 7; i < argc; i++) { const std::string a = argv[i]; if (a == "by-clsid") byClsidD = true; else if (a.rfind("size=", 0) == 0) { statedSize = strtoull(a.c_str() + 5, nullptr, 10); haveSize = true; } }
        void* rawD = nullptr; HRESULT rd;
        if (byClsidD) { GUID c = gc_codec_clsid(foundId, false); rd = createObj(&c, &IID_ICompressCoder, &rawD); }
        else rd = createDec((uint32_t)found, &IID_ICompressCoder, &rawD);
        if (rd != S_OK || !rawD) { fprintf(stderr, "CreateDecoder failed: %08X\n", (unsigned)rd); return 9; }
        ICompressCoder* dec = (ICompressCoder*)rawD;
        ICompressSetDecoderProperties2* sp = nullptr;
        if (dec->QueryInterface(IID_ICompressSetDecoderProperties2, (void**)&sp) != S_OK) return 11;
        if (std::string(argv[4]) != "-") {
            FILE* pf = fopen(argv[4], "rb"); if (!pf) return 13;
            uint8_t pb[16]; size_t pn = fread(pb, 1, sizeof(pb), pf); fclose(pf);
            if (sp->SetDecoderProperties2(pb, (uint32_t)pn) != S_OK) { fprintf(stderr, "SetDecoderProperties2 refused %zu bytes\n", pn); return 12; }
        }
        if (sp->SetDecoderProperties2(nullptr, 7) != E_NOTIMPL) { fprintf(stderr, "property size check\n"); return 12; }
        FileIn in; in.f = fopen(argv[5], "rb"); FileOut out; out.f = fopen(argv[6], "wb");
        if (!in.f || !out.f) { fprintf(stderr, "file open\n"); return 13; }
        Progress prog;
        const uint64_t outSizeArg = statedSize;
        HRESULT r2 = dec->Code(&in, &out, nullptr, haveSize ? &outSizeArg : nullptr, &prog);
        fclose(in.f); fclose(out.f);
        if (r2 != S_OK) { fprintf(stderr, "Code failed: %08X\n", (unsigned)r2); return 15; }
        if (prog.out != out.total) { fprintf(stderr, "progress accounting\n"); return 16; }
        if (in.refs != 1 || out.refs != 1 || prog.refs != 1) { fprintf(stderr, "coder kept a stream reference\n"); return 17; }
        sp->Release();
        if (dec->Release() != 0) { fprintf(stderr, "refcount leak\n"); return 18; }
        printf("ok in=%llu out=%llu\n", (unsigned long long)prog.in, (unsigned long long)prog.out);
        return 0;
    }
    if (mode != "encode" || argc < 7 || found < 0) { fprintf(stderr, "method not found\n"); return 8; }
    const bool byClsid = argc > 8 && std::string(argv[8]) == "by-clsid";
    void* raw = nullptr;
    HRESULT r;
    if (byClsid) { GUID c = gc_codec_clsid(foundId, true); r = createObj(&c, &IID_ICompressCoder, &raw); }
    else r = createEnc((uint32_t)found, &IID_ICompressCoder, &raw);
    if (r != S_OK || !raw) { fprintf(stderr, "CreateEncoder failed: %08X\n", (unsigned)r); return 9; }
    // wrong interface id must be refused, decoders exist for ZSTD and BROTLI only
    { void* bad = nullptr; if (createEnc((uint32_t)found, &IID_ISequentialInStream, &bad) != E_NOINTERFACE || bad) { fprintf(stderr, "iid check\n"); return 10; }
      if (foundId != 0x4F71101 && foundId != 0x4F71102 && (createDec((uint32_t)found, &IID_ICompressCoder, &bad) != CLASS_E_CLASSNOTAVAILABLE || bad)) { fprintf(stderr, "decoder check\n"); return 10; } }
    ICompressCoder* coder = (ICompressCoder*)raw;
    ICompressSetCoderProperties* setProps = nullptr; ICompressWriteCoderProperties* writeProps = nullptr;
    ICompressSetCoderMt* mt = nullptr; ICompressSetCoderPropertiesOpt* opt = nullptr; IUnknown* unk = nullptr; void* none = nullptr;
    if (coder->QueryInterface(IID_ICompressSetCoderProperties, (void**)&setProps) != S_OK) return 11;
    if (coder->QueryInterface(IID_ICompressWriteCoderProperties, (void**)&writeProps) != S_OK) return 11;
    if (coder->QueryInterface(IID_ICompressSetCoderMt, (void**)&mt) != S_OK) return 11;
    if (coder->QueryInterface(IID_ICompressSetCoderPropertiesOpt, (void**)&opt) != S_OK) return 11;
    if (coder->QueryInterface(IID_IUnknown, (void**)&unk) != S_OK) return 11;
    if (coder->QueryInterface(IID_ISequentialOutStream, &none) != E_NOINTERFACE) return 11;
    // "threads0" as the last argument: what the reference's bare .br handler does with its encoder (BrotliHandler.cpp:286-291) -- a plain stream
    const bool threads0 = std::string(argv[argc - 1]) == "threads0";
    mt->SetNumberOfThreads(threads0 ? 0 : 8);
    PROPID ids[2] = { NCoderPropID::kLevel, NCoderPropID::kNumThreads }; PROPVARIANT pv[2]; memset(pv, 0, sizeof(pv));
    pv[0].vt = VT_UI4; pv[0].ulVal = (uint32_t)atoi(argv[4]); pv[1].vt = VT_UI4; pv[1].ulVal = 8;
    if (setProps->SetCoderProperties(ids, pv, threads0 ? 1 : 2) != S_OK) return 12;
    FileIn in; in.f = fopen(argv[5], "rb"); FileOut out; out.f = fopen(argv[6], "wb");
    if (!in.f || !out.f) { fprintf(stderr, "file open\n"); return 13; }
    { fseek(in.f, 0, SEEK_END); uint64_t sz = (uint64_t)ftell(in.f); fseek(in.f, 0, SEEK_SET);
      PROPID oid = NCoderPropID::kExpectedDataSize; PROPVARIANT ov; memset(&ov, 0, sizeof(ov)); ov.vt = VT_UI8; ov.uhVal = sz; opt->SetCoderPropertiesOpt(&oid, &ov, 1); }
    if (argc > 7 && std::string(argv[7]) != "-") { FileOut p; p.f = fopen(argv[7], "wb"); if (!p.f || writeProps->WriteCoderProperties(&p) != S_OK) return 14; fclose(p.f); }
    Progress prog;
    r = coder->Code(&in, &out, nullptr, nullptr, &prog);
    fclose(in.f); fclose(out.f);
    if (r != S_OK) { fprintf(stderr, "Code failed: %08X\n", (unsigned)r); return 15; }
    if (prog.calls == 0 || prog.out != out.total) { fprintf(stderr, "progress accounting\n"); return 16; }
    if (in.refs != 1 || out.refs != 1 || prog.refs != 1) { fprintf(stderr, "coder kept a stream reference\n"); return 17; }
    unk->Release(); opt->Release(); mt->Release(); writeProps->Release(); setProps->Release();
    if (coder->Release() != 0) { fprintf(stderr, "refcount leak\n"); return 18; }
    printf("ok in=%llu out=%llu\n", (unsigned long long)prog.in, (unsigned long long)prog.out);
    return 0;
}
