/* oracle/corpus_gen.c -- TEST / BENCH INPUT GENERATORS (no codec logic).
 *
 * enwik8 / enwik9 / Silesia are not on disk and cannot be downloaded, so the
 * harness generates deterministic stand-ins and labels every result "synthetic":
 *
 *   gc_corpus_text_zipf    Zipf-distributed pseudo-text: 50 000-word vocabulary (word length
 *                          U[2,10], letters ~ rank^-0.8 over "etaoinshrdlcumwfgypbvkjxqz"), word
 *                          rank ~ Zipf(1.07), separators ' ' / ". " 8% / ", " 8% / '\n' 2%.
 *   gc_corpus_lz7zip       the reference's own benchmark generator CBenchRandomGenerator::GenerateLz
 *                          restated from CPP/7zip/UI/Common/Bench.cpp:117-256 (MWC RNG
 *                          A1=362436069, A2=521288629; literal-or-match stream with log-uniform
 *                          distances up to 2^dictBits).
 *   gc_corpus_silesia_like 40% text_zipf, 30% lz7zip, 15% 16-bit AR(1) "PCM", 10% opcode soup,
 *                          5% uniform random, interleaved in 12 files-worth of segments.
 *   gc_corpus_webtext      text_zipf wrapped in <p>..</p> lines + 5% boilerplate from a 64 KiB pool
 *                          (the brotli "web-text" config).
 *
 * All generators are pure functions of (size, seed).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef struct { uint64_t s; } rng_t;
static uint64_t rng_next(rng_t* r)
{   /* splitmix64 */
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double rng_unit(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static uint32_t rng_below(rng_t* r, uint32_t n) { return (uint32_t)(((rng_next(r) >> 32) * (uint64_t)n) >> 32); }

/* ---------------------------------------------------------------- text-zipf */
#define VOCAB 50000
typedef struct { char w[VOCAB][11]; uint8_t len[VOCAB]; double cdf[VOCAB]; } vocab_t;

static vocab_t* make_vocab(uint64_t seed)
{
    static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
    double lcdf[26], tot = 0; int i, j;
    vocab_t* v = (vocab_t*)malloc(sizeof(*v));
    rng_t r = { seed ^ 0x5EED5EEDULL };
    for (i = 0; i < 26; i++) { tot += pow(i + 1.0, -0.8); lcdf[i] = tot; }
    for (i = 0; i < VOCAB; i++) {
        int L = 2 + (int)rng_below(&r, 9);
        for (j = 0; j < L; j++) {
            double u = rng_unit(&r) * tot; int k = 0;
            while (k < 25 && lcdf[k] < u) k++;
            v->w[i][j] = letters[k];
        }
        v->w[i][L] = 0; v->len[i] = (uint8_t)L;
    }
    tot = 0;
    for (i = 0; i < VOCAB; i++) { tot += pow(i + 1.0, -1.07); v->cdf[i] = tot; }
    for (i = 0; i < VOCAB; i++) v->cdf[i] /= tot;
    return v;
}
static int vocab_pick(const vocab_t* v, rng_t* r)
{
    double u = rng_unit(r); int lo = 0, hi = VOCAB - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (v->cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}

void gc_corpus_text_zipf(uint8_t* buf, size_t n, uint64_t seed)
{
    vocab_t* v = make_vocab(20260921ULL);     /* vocabulary is fixed; `seed` drives the word stream */
    rng_t r = { seed };
    size_t pos = 0;
    while (pos < n) {
        int w = vocab_pick(v, &r); int L = v->len[w], i; uint32_t sep = rng_below(&r, 100);
        for (i = 0; i < L && pos < n; i++) buf[pos++] = (uint8_t)v->w[w][i];
        if (sep < 8) { if (pos < n) buf[pos++] = '.'; if (pos < n) buf[pos++] = ' '; }
        else if (sep < 16) { if (pos < n) buf[pos++] = ','; if (pos < n) buf[pos++] = ' '; }
        else if (sep < 18) { if (pos < n) buf[pos++] = '\n'; }
        else if (pos < n) buf[pos++] = ' ';
    }
    free(v);
}

void gc_corpus_webtext(uint8_t* buf, size_t n, uint64_t seed)
{
    vocab_t* v = make_vocab(20260921ULL);
    rng_t r = { seed ^ 0x3E87E87ULL };
    size_t pos = 0, poolN = 64 * 1024;
    uint8_t* pool = (uint8_t*)malloc(poolN);
    gc_corpus_text_zipf(pool, poolN, 777);
    while (pos < n) {
        if (rng_below(&r, 100) < 5) {   /* boilerplate: a 200..1200-byte slice of the template pool */
            size_t L = 200 + rng_below(&r, 1000), o = rng_below(&r, (uint32_t)(poolN - L)), i;
            static const char open[] = "<div class=\"nav\">"; static const char close[] = "</div>\n";
            for (i = 0; open[i] && pos < n; i++) buf[pos++] = (uint8_t)open[i];
            for (i = 0; i < L && pos < n; i++) buf[pos++] = pool[o + i];
            for (i = 0; close[i] && pos < n; i++) buf[pos++] = (uint8_t)close[i];
        } else {
            int words = 8 + (int)rng_below(&r, 40), k; size_t i;
            static const char open[] = "<p>"; static const char close[] = "</p>\n";
            for (i = 0; open[i] && pos < n; i++) buf[pos++] = (uint8_t)open[i];
            for (k = 0; k < words; k++) {
                int w = vocab_pick(v, &r); int L = v->len[w], j;
                for (j = 0; j < L && pos < n; j++) buf[pos++] = (uint8_t)v->w[w][j];
                if (k + 1 < words && pos < n) buf[pos++] = rng_below(&r, 10) == 0 ? ',' : ' ';
            }
            for (i = 0; close[i] && pos < n; i++) buf[pos++] = (uint8_t)close[i];
        }
    }
    free(pool); free(v);
}

/* ---------------------------------------------------------------- lz-7zip (Bench.cpp:117-256) */
typedef struct { uint32_t a1, a2, salt; } mwc_t;
static uint32_t mwc_next(mwc_t* g)
{
    g->a1 = 36969u * (g->a1 & 0xffff) + (g->a1 >> 16);
    g->a2 = 18000u * (g->a2 & 0xffff) + (g->a2 >> 16);
    return g->salt ^ ((g->a1 << 16) + g->a2);
}
static uint32_t take_bits(uint32_t* r, unsigned nb) { uint32_t v = *r & ((1u << nb) - 1); *r >>= nb; return v; }
static uint32_t take_len(uint32_t* r) { unsigned l = take_bits(r, 2); return take_bits(r, 1 + l); }

void gc_corpus_lz7zip(uint8_t* buf, size_t n, unsigned dictBits, uint32_t salt)
{
    mwc_t g = { 362436069u, 521288629u, salt };
    size_t pos = 0, rep0 = 1; unsigned posBits = 1;
    while (pos < n) {
        uint32_t r = mwc_next(&g);
        if (take_bits(&r, 1) == 0 || pos < 1024) { buf[pos++] = (uint8_t)(r & 0xFF); continue; }
        {
            uint32_t len = 1 + take_len(&r);
            if (take_bits(&r, 3) != 0) {
                unsigned maxBits, logBits = 5;
                len += take_len(&r);
                while (((size_t)1 << posBits) < pos) posBits++;
                maxBits = dictBits < posBits ? dictBits : posBits;
                if (maxBits <= 15 + 6) logBits = 4;
                for (;;) {
                    uint32_t ppp = take_bits(&r, logBits) + 6;
                    r = mwc_next(&g);
                    if (ppp > maxBits) continue;
                    rep0 = r & (((size_t)1 << ppp) - 1);
                    if (rep0 < pos) break;
                    r = mwc_next(&g);
                }
                rep0++;
            }
            if (len > n - pos) len = (uint32_t)(n - pos);
            { size_t i; for (i = 0; i < len; i++) buf[pos + i] = buf[pos + i - rep0]; }
            pos += len;
        }
    }
}

/* ---------------------------------------------------------------- silesia-like mix */
static void gen_pcm(uint8_t* buf, size_t n, rng_t* r)
{   /* 16-bit little-endian AR(1) noise: strongly correlated high bytes, noisy low bytes */
    double x = 0; size_t i;
    for (i = 0; i + 1 < n; i += 2) {
        double e = (rng_unit(r) + rng_unit(r) + rng_unit(r) - 1.5) * 900.0;
        int v; x = 0.985 * x + e; v = (int)x;
        if (v > 32767) v = 32767;
        if (v < -32768) v = -32768;
        buf[i] = (uint8_t)(v & 0xFF); buf[i + 1] = (uint8_t)((v >> 8) & 0xFF);
    }
    if (n & 1) buf[n - 1] = 0;
}
static void gen_opcodes(uint8_t* buf, size_t n, rng_t* r)
{   /* x86-like: small opcode alphabet, modrm bytes, 4-byte little-endian displacements that repeat */
    static const uint8_t ops[] = { 0x8B, 0x89, 0xE8, 0x48, 0x83, 0xFF, 0x0F, 0x74, 0x75, 0xC3, 0x55, 0x5D, 0x8D, 0x85, 0x31, 0xEB };
    uint32_t targets[256]; size_t pos = 0; int i;
    for (i = 0; i < 256; i++) targets[i] = (uint32_t)rng_next(r) & 0x000FFFFF;
    while (pos < n) {
        uint8_t op = ops[rng_below(r, 16)];
        buf[pos++] = op;
        if (op == 0xE8 || op == 0x8D) { uint32_t t = targets[rng_below(r, 256)]; int k; for (k = 0; k < 4 && pos < n; k++) buf[pos++] = (uint8_t)(t >> (8 * k)); }
        else if (op == 0x8B || op == 0x89 || op == 0x83) { if (pos < n) buf[pos++] = (uint8_t)(0x40 | rng_below(r, 64)); if (pos < n) buf[pos++] = (uint8_t)(rng_below(r, 16) * 8); }
        else if (op == 0x74 || op == 0x75 || op == 0xEB) { if (pos < n) buf[pos++] = (uint8_t)rng_below(r, 128); }
    }
}

void gc_corpus_silesia_like(uint8_t* buf, size_t n, uint64_t seed)
{
    /* 20 segments in a fixed interleave so that every 1/20th of the corpus has one kind */
    static const uint8_t kind[20] = { 0,1,0,2,1,0,3,0,1,2, 0,1,0,4,1,0,3,2,1,0 };  /* 8 text,6 lz,3 pcm,2 op,1 rnd */
    rng_t r = { seed };
    size_t seg = n / 20, pos = 0; int i;
    for (i = 0; i < 20; i++) {
        size_t len = (i == 19) ? n - pos : seg;
        switch (kind[i]) {
        case 0: gc_corpus_text_zipf(buf + pos, len, seed + 100 + (uint64_t)i); break;
        case 1: gc_corpus_lz7zip(buf + pos, len, 24, (uint32_t)(seed + (uint64_t)i)); break;
        case 2: gen_pcm(buf + pos, len, &r); break;
        case 3: gen_opcodes(buf + pos, len, &r); break;
        default: { size_t k; for (k = 0; k < len; k++) buf[pos + k] = (uint8_t)rng_next(&r); } break;
        }
        pos += len;
    }
}
