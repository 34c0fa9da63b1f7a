/* oracle/ref_shim_brotli.c -- TEST INFRASTRUCTURE ONLY.
 * Entry points over the REFERENCE's brotli 1.2.0 (C/brotli) and the brotli-mt framing
 * (C/zstdmt/brotli-mt_*.c) that defines the BROTLI wire format inside 7z
 * (CPP/7zip/Compress/BrotliEncoder.cpp:150-151: lgwin 24, chunk = 1 MiB x level).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include "encode.h"
#include "decode.h"
#include "brotli-mt.h"

size_t ref_brotli_compress(void* dst, size_t cap, const void* src, size_t n, int quality, int lgwin)
{
    size_t out = cap;
    if (!BrotliEncoderCompress(quality, lgwin, BROTLI_MODE_GENERIC, n, (const uint8_t*)src, &out, (uint8_t*)dst))
        return (size_t)-1;
    return out;
}

size_t ref_brotli_decompress(void* dst, size_t cap, const void* src, size_t n)
{
    size_t out = cap;
    if (BrotliDecoderDecompress(n, (const uint8_t*)src, &out, (uint8_t*)dst) != BROTLI_DECODER_RESULT_SUCCESS)
        return (size_t)-1;
    return out;
}

typedef struct { const uint8_t* p; size_t n, off; } rd_t;
typedef struct { uint8_t* p; size_t cap, off; int ovf; } wr_t;
static int rd_fn(void* a, BROTLIMT_Buffer* in)
{
    rd_t* r = (rd_t*)a; size_t take = r->n - r->off;
    if (take > in->size) take = in->size;
    memcpy(in->buf, r->p + r->off, take); r->off += take; in->size = take; return 0;
}
static int wr_fn(void* a, BROTLIMT_Buffer* out)
{
    wr_t* w = (wr_t*)a;
    if (w->off + out->size > w->cap) { w->ovf = 1; return -1; }
    memcpy(w->p + w->off, out->buf, out->size); w->off += out->size; return 0;
}

/* brotli-mt framed stream, as CEncoder::Code produces (BrotliEncoder.cpp:118-164) */
size_t ref_brotlimt_compress(void* dst, size_t cap, const void* src, size_t n, int level, int threads)
{
    rd_t r = { (const uint8_t*)src, n, 0 }; wr_t w = { (uint8_t*)dst, cap, 0, 0 };
    BROTLIMT_RdWr_t rw = { rd_fn, &r, wr_fn, &w };
    BROTLIMT_CCtx* c = BROTLIMT_createCCtx(threads, n, level, 0, 24);
    size_t rv;
    if (!c) return (size_t)-1;
    rv = BROTLIMT_compressCCtx(c, &rw);
    BROTLIMT_freeCCtx(c);
    return (BROTLIMT_isError(rv) || w.ovf) ? (size_t)-1 : w.off;
}

size_t ref_brotlimt_decompress(void* dst, size_t cap, const void* src, size_t n, int threads)
{
    rd_t r = { (const uint8_t*)src, n, 0 }; wr_t w = { (uint8_t*)dst, cap, 0, 0 };
    BROTLIMT_RdWr_t rw = { rd_fn, &r, wr_fn, &w };
    BROTLIMT_DCtx* c = BROTLIMT_createDCtx(threads, 0);
    size_t rv;
    if (!c) return (size_t)-1;
    rv = BROTLIMT_decompressDCtx(c, &rw);
    BROTLIMT_freeDCtx(c);
    return (BROTLIMT_isError(rv) || w.ovf) ? (size_t)-1 : w.off;
}
