// gc_7z_abi.h -- the slice of 7-Zip's binary plugin ABI that a codec module has to speak (Linux, Itanium C++ ABI).
//
// Written from the layout rules, not from the reference headers; every item names the reference line that fixes it
// (paths relative to /root/reference):
//   * HRESULT values                       CPP/Common/MyWindows.h:95-101, C/7zTypes.h:132-133
//   * GUID, PROPVARIANT (16 bytes), BSTR   CPP/Common/MyGuidDef.h, MyWindows.h:227-250, MyWindows.cpp:15-92
//   * IUnknown = 3 vtable slots, NO virtual destructor (module reports kInterfaceType 0)   MyWindows.h:170-182,
//     CPP/7zip/ICoder.h:422-440
//   * interface IDs {23170F69-40C1-278A-0000-00gg-00ss-0000}                               CPP/7zip/IDecl.h:9-27
//   * stream interfaces (group 3)          CPP/7zip/IStream.h:47-70
//   * coder interfaces (group 4) and the order of their methods                            CPP/7zip/ICoder.h:14-31,172-275
//   * property ids                         CPP/7zip/ICoder.h:104-160 (NCoderPropID), :405-420 (NMethodPropID), :443-449
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <wchar.h>

typedef int32_t  HRESULT;
typedef uint32_t ULONG;
typedef uint32_t PROPID;
typedef uint16_t VARTYPE;

#define S_OK            ((HRESULT)0)
#define S_FALSE         ((HRESULT)1)
#define E_NOTIMPL       ((HRESULT)0x80004001)
#define E_NOINTERFACE   ((HRESULT)0x80004002)
#define E_ABORT         ((HRESULT)0x80004004)
#define E_FAIL          ((HRESULT)0x80004005)
#define E_OUTOFMEMORY   ((HRESULT)0x8007000E)
#define E_INVALIDARG    ((HRESULT)0x80070057)
#define CLASS_E_CLASSNOTAVAILABLE ((HRESULT)0x80040111)

struct GUID { uint32_t Data1; uint16_t Data2; uint16_t Data3; uint8_t Data4[8]; };
static inline bool operator==(const GUID& a, const GUID& b) { return memcmp(&a, &b, sizeof(GUID)) == 0; }

enum { VT_EMPTY = 0, VT_BSTR = 8, VT_BOOL = 11, VT_UI4 = 19, VT_UI8 = 21 };
typedef wchar_t* BSTR;
struct PROPVARIANT {
    VARTYPE vt; uint16_t r1, r2, r3;
    union { uint32_t ulVal; uint64_t uhVal; int16_t boolVal; BSTR bstrVal; };
};
static_assert(sizeof(PROPVARIANT) == 16, "PROPVARIANT must be 16 bytes");

// BSTR = [u32 byte length][payload][zero wchar]; allocated with malloc, the HOST frees it with free()
static inline BSTR gc_bstr_bytes(const void* p, uint32_t len)
{
    const uint32_t size = (len + 2u * (uint32_t)sizeof(wchar_t) - 1u) & ~((uint32_t)sizeof(wchar_t) - 1u);
    uint8_t* m = (uint8_t*)malloc(size + 4u);
    if (!m) return nullptr;
    memcpy(m, &len, 4);
    memcpy(m + 4, p, len);
    memset(m + 4 + len, 0, size - len);
    return (BSTR)(m + 4);
}
static inline BSTR gc_bstr_ascii(const char* s)
{
    const uint32_t n = (uint32_t)strlen(s), bytes = n * (uint32_t)sizeof(wchar_t);
    uint8_t* m = (uint8_t*)malloc(bytes + 4u + sizeof(wchar_t));
    if (!m) return nullptr;
    memcpy(m, &bytes, 4);
    wchar_t* w = (wchar_t*)(m + 4);
    for (uint32_t i = 0; i < n; i++) w[i] = (wchar_t)(unsigned char)s[i];
    w[n] = 0;
    return w;
}
static inline void gc_variant_clear(PROPVARIANT* v)
{
    if (v->vt == VT_BSTR && v->bstrVal) free((uint8_t*)v->bstrVal - 4);
    memset(v, 0, sizeof(*v));
}

#define GC_7Z_IID(group, sub) GUID{ 0x23170F69u, 0x40C1, 0x278A, { 0, 0, 0, (group), 0, (sub), 0, 0 } }
static const GUID IID_IUnknown = { 0, 0, 0, { 0xC0, 0, 0, 0, 0, 0, 0, 0x46 } };
static const GUID IID_ISequentialInStream            = GC_7Z_IID(3, 0x01);
static const GUID IID_ISequentialOutStream           = GC_7Z_IID(3, 0x02);
static const GUID IID_ICompressProgressInfo          = GC_7Z_IID(4, 0x04);
static const GUID IID_ICompressCoder                 = GC_7Z_IID(4, 0x05);
static const GUID IID_ICompressSetCoderPropertiesOpt = GC_7Z_IID(4, 0x1F);
static const GUID IID_ICompressSetCoderProperties    = GC_7Z_IID(4, 0x20);
static const GUID IID_ICompressSetDecoderProperties2 = GC_7Z_IID(4, 0x22);
static const GUID IID_ICompressWriteCoderProperties  = GC_7Z_IID(4, 0x23);
static const GUID IID_ICompressSetCoderMt            = GC_7Z_IID(4, 0x25);
static const GUID IID_ICompressFilter                = GC_7Z_IID(4, 0x40);      // ICoder.h:362-365

// class ids: {23170F69-40C1-2791(encoder)/2790(decoder)-<method id as 8 little-endian bytes>}  CodecExports.cpp:44-52
static inline GUID gc_codec_clsid(uint64_t methodId, bool encoder)
{
    GUID g = { 0x23170F69u, 0x40C1, (uint16_t)(encoder ? 0x2791 : 0x2790), { 0 } };
    for (int i = 0; i < 8; i++) g.Data4[i] = (uint8_t)(methodId >> (8 * i));
    return g;
}

struct IUnknown {
    virtual HRESULT QueryInterface(const GUID& iid, void** out) = 0;
    virtual ULONG AddRef() = 0;
    virtual ULONG Release() = 0;
};
struct ISequentialInStream : IUnknown { virtual HRESULT Read(void* data, uint32_t size, uint32_t* processed) = 0; };
struct ISequentialOutStream : IUnknown { virtual HRESULT Write(const void* data, uint32_t size, uint32_t* processed) = 0; };
struct ICompressProgressInfo : IUnknown { virtual HRESULT SetRatioInfo(const uint64_t* inSize, const uint64_t* outSize) = 0; };
struct ICompressCoder : IUnknown {
    virtual HRESULT Code(ISequentialInStream* in, ISequentialOutStream* out, const uint64_t* inSize, const uint64_t* outSize,
                         ICompressProgressInfo* progress) = 0;
};
struct ICompressSetCoderPropertiesOpt : IUnknown { virtual HRESULT SetCoderPropertiesOpt(const PROPID* ids, const PROPVARIANT* props, uint32_t n) = 0; };
struct ICompressSetCoderProperties : IUnknown { virtual HRESULT SetCoderProperties(const PROPID* ids, const PROPVARIANT* props, uint32_t n) = 0; };
struct ICompressWriteCoderProperties : IUnknown { virtual HRESULT WriteCoderProperties(ISequentialOutStream* out) = 0; };
struct ICompressSetCoderMt : IUnknown { virtual HRESULT SetNumberOfThreads(uint32_t n) = 0; };
struct ICompressFilter : IUnknown { virtual HRESULT Init() = 0; virtual uint32_t Filter(uint8_t* data, uint32_t size) = 0; };      // ICoder.h:320-365
struct ICompressSetDecoderProperties2 : IUnknown { virtual HRESULT SetDecoderProperties2(const uint8_t* data, uint32_t size) = 0; };   // ICoder.h:187-189

namespace NCoderPropID { enum { kDefaultProp = 0, kDictionarySize = 1, kNumThreads = 13, kLevel = 15, kReduceSize = 16, kExpectedDataSize = 17, kBranchOffset = 23 }; }
namespace NMethodPropID { enum { kID = 0, kName, kDecoder, kEncoder, kPackStreams, kUnpackStreams, kDescription, kDecoderIsAssigned, kEncoderIsAssigned, kDigestSize, kIsFilter }; }
namespace NModulePropID { enum { kInterfaceType = 0, kVersion = 1 }; }
