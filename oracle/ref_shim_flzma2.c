/* oracle/ref_shim_flzma2.c -- TEST INFRASTRUCTURE ONLY.
 * Entry points over the REFERENCE's Fast-LZMA2 encoder (C/fast-lzma2) and the stock LZMA2
 * decoder the 7-Zip codec registers for FLZMA2 (C/Lzma2Dec.c; FastLzma2Register.cpp:15).
 * Parameters follow CFastEncoder (CPP/7zip/Compress/Lzma2Encoder.cpp:178-239): level table,
 * omitProperties=1 (the dict-size byte travels in the 7z coder props, :353-364).
 */
#include <stddef.h>
#include <stdlib.h>
#include "fast-lzma2.h"
#include "Lzma2Dec.h"
#include "Alloc.h"

static void* sz_alloc(ISzAllocPtr p, size_t n) { (void)p; return malloc(n); }
static void sz_free(ISzAllocPtr p, void* a) { (void)p; free(a); }
static const ISzAlloc g_alloc = { sz_alloc, sz_free };

/* returns compressed size; *prop receives the 1-byte dictionary-size property */
size_t ref_fl2_compress(void* dst, size_t cap, const void* src, size_t n, int level, unsigned threads,
                        unsigned char* prop)
{
    FL2_CCtx* c = FL2_createCCtxMt(threads);
    size_t r;
    if (!c) return (size_t)-1;
    FL2_CCtx_setParameter(c, FL2_p_compressionLevel, (size_t)level);
    FL2_CCtx_setParameter(c, FL2_p_omitProperties, 1);
    r = FL2_compressCCtx(c, dst, cap, src, n, 0);
    if (prop) *prop = FL2_getCCtxDictProp(c);
    FL2_freeCCtx(c);
    return FL2_isError(r) ? (size_t)-1 : r;
}

/* one-call LZMA2 decode (Lzma2Dec.c Lzma2Decode) */
size_t ref_lzma2_decode(void* dst, size_t cap, const void* src, size_t n, unsigned char prop)
{
    SizeT dl = cap, sl = n;
    ELzmaStatus st;
    SRes r = Lzma2Decode((Byte*)dst, &dl, (const Byte*)src, &sl, prop, LZMA_FINISH_END, &st, &g_alloc);
    if (r != SZ_OK) return (size_t)-1;
    if (st != LZMA_STATUS_FINISHED_WITH_MARK) return (size_t)-2;
    return dl;
}
