/* oracle/ref_shim_zstd.c -- TEST INFRASTRUCTURE ONLY.
 * Thin C entry points over the REFERENCE's own zstd (compiled from /root/reference/C/zstd by
 * oracle/Makefile into oracle/_ref/libzstd_ref.so).  Only the reference's public API is called
 * (C/zstd/zstd.h); parameters mirror what the 7-Zip wrapper sets for a plain `-m0=zstd -mxN`
 * (CPP/7zip/Compress/ZstdEncoder.cpp:296-305: contentSizeFlag=1, checksum off, no dict).
 */
#include <stddef.h>
#include <string.h>
#define ZSTD_STATIC_LINKING_ONLY
#include "zstd.h"

/* one stream, like CEncoder::Code() -> ZSTD_compressStream2 (ZstdEncoder.cpp:398-461) */
size_t ref_zstd_compress(void* dst, size_t cap, const void* src, size_t n, int level, int nbWorkers)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r;
    if (!c) return (size_t)-1;
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_contentSizeFlag, 1);
    ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, 0);
    ZSTD_CCtx_setParameter(c, ZSTD_c_nbWorkers, nbWorkers);
    r = ZSTD_compress2(c, dst, cap, src, n);
    ZSTD_freeCCtx(c);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

/* independent frames of `piece` input bytes each, concatenated (BASELINE.md independence probe) */
size_t ref_zstd_compress_pieces(void* dst, size_t cap, const void* src, size_t n, int level, size_t piece)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t off = 0, out = 0;
    if (!c) return (size_t)-1;
    while (off < n || (n == 0 && off == 0)) {
        size_t take = n - off < piece ? n - off : piece;
        size_t r;
        ZSTD_CCtx_reset(c, ZSTD_reset_session_and_parameters);
        ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
        ZSTD_CCtx_setParameter(c, ZSTD_c_contentSizeFlag, 1);
        r = ZSTD_compress2(c, (char*)dst + out, cap - out, (const char*)src + off, take);
        if (ZSTD_isError(r)) { ZSTD_freeCCtx(c); return (size_t)-1; }
        out += r; off += take;
        if (n == 0) break;
    }
    ZSTD_freeCCtx(c);
    return out;
}

/* one frame with the options the decoder tests need: flags bit0 = content checksum, bit1 = streamed in 100 000-byte writes without a pledged
 * size (what CEncoder::Code does when the size is not known: no Frame_Content_Size field, window descriptor instead), bit2 = long-distance
 * matching on with windowLog 27 (large offsets) */
size_t ref_zstd_compress_opts(void* dst, size_t cap, const void* src, size_t n, int level, unsigned flags)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r = 0;
    if (!c) return (size_t)-1;
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, (flags & 1u) ? 1 : 0);
    if (flags & 4u) { ZSTD_CCtx_setParameter(c, ZSTD_c_enableLongDistanceMatching, 1); ZSTD_CCtx_setParameter(c, ZSTD_c_windowLog, 27); }
    if (flags & 2u) {
        ZSTD_inBuffer in; ZSTD_outBuffer out;
        size_t off = 0;
        out.dst = dst; out.size = cap; out.pos = 0;
        for (;;) {
            size_t take = n - off < 100000u ? n - off : 100000u;
            int last = off + take == n;
            in.src = (const char*)src + off; in.size = take; in.pos = 0;
            do {
                r = ZSTD_compressStream2(c, &out, &in, last ? ZSTD_e_end : ZSTD_e_continue);
                if (ZSTD_isError(r) || (out.pos == out.size && (r || in.pos < in.size))) { ZSTD_freeCCtx(c); return (size_t)-1; }
            } while (last ? r != 0 : in.pos < in.size);
            off += take;
            if (last) break;
        }
        r = out.pos;
    } else {
        ZSTD_CCtx_setParameter(c, ZSTD_c_contentSizeFlag, 1);
        r = ZSTD_compress2(c, dst, cap, src, n);
    }
    ZSTD_freeCCtx(c);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

/* multi-frame decode exactly as the 7-Zip decoder loop does (ZstdDecoder.cpp:145-158):
 * ZSTD_decompress() itself walks concatenated + skippable frames. */
size_t ref_zstd_decompress(void* dst, size_t cap, const void* src, size_t n)
{
    size_t r = ZSTD_decompress(dst, cap, src, n);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

const char* ref_zstd_last_error_name(size_t code) { return ZSTD_getErrorName(code); }
size_t ref_zstd_compress_bound(size_t n) { return ZSTD_compressBound(n); }
unsigned ref_zstd_version(void) { return ZSTD_versionNumber(); }

/* Let the reference entropy-code externally produced sequences (zstd.h:1680 ZSTD_compressSequences):
 * used to cross-check the GPU match finder independently of the GPU entropy stage. */
size_t ref_zstd_compress_sequences(void* dst, size_t cap, const unsigned* seq_off, const unsigned* seq_ll,
                                   const unsigned* seq_ml, size_t nseq, const void* src, size_t n, int level)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    ZSTD_Sequence* s;
    size_t i, r;
    if (!c) return (size_t)-1;
    s = (ZSTD_Sequence*)malloc((nseq + 1) * sizeof(*s));
    for (i = 0; i < nseq; i++) { s[i].offset = seq_off[i]; s[i].litLength = seq_ll[i]; s[i].matchLength = seq_ml[i]; s[i].rep = 0; }
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_blockDelimiters, ZSTD_sf_noBlockDelimiters);
    ZSTD_CCtx_setParameter(c, ZSTD_c_validateSequences, 1);
    r = ZSTD_compressSequences(c, dst, cap, s, nseq, src, n);
    free(s); ZSTD_freeCCtx(c);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}
