/* oracle/lzma2_dec.c -- TEST INFRASTRUCTURE ONLY (checker; never linked into the product).
 *
 * Plain-C restatement of the LZMA2 decoder the 7-Zip codec registers for FLZMA2
 * (NCompress::NLzma2::CDecoder, CPP/7zip/Compress/FastLzma2Register.cpp:15), following the reference's
 * C/Lzma2Dec.c (chunk walk, control bytes :97-220, property/dictionary rules :60-95) and C/LzmaDec.c
 * (symbol grammar and state machine :229-560: literal / matched literal, match, rep0..3, short rep; length coder; position
 * slot + reverse bit trees + direct bits + align; probability update kNumMoveBits = 5; range decoder normalisation).
 * One pass, whole buffers, no streaming states: the point is an independent statement of the format, pinned against the
 * reference decoder and the reference encoder's output in tests/test_oracle.py.
 *
 * Returns the number of bytes produced, or (size_t)-1 - k on error k (k = 0 generic, others identify the check that failed).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define kNumStates 12
#define kNumLitStates 7
#define LEN_CHOICE 0
#define LEN_CHOICE2 1
#define LEN_LOW 2          /* [16 posStates][8] */
#define LEN_MID (LEN_LOW + 16 * 8)
#define LEN_HIGH (LEN_MID + 16 * 8)
#define LEN_SIZE (LEN_HIGH + 256)

typedef struct {
    const uint8_t* in; size_t inPos, inEnd;
    uint32_t range, code;
    int err;
} Rc;

typedef struct {
    uint16_t isMatch[kNumStates][16], isRep[kNumStates], isRepG0[kNumStates], isRepG1[kNumStates], isRepG2[kNumStates],
             isRep0Long[kNumStates][16];
    uint16_t lenC[LEN_SIZE], repLenC[LEN_SIZE];
    uint16_t posSlot[4][64], specPos[128], align[16];
    uint16_t lit[0x300 << 4];      /* lc + lp <= 4 in LZMA2 (Lzma2Dec.c:60-75) */
} Probs;

static void probs_init(Probs* p) { uint16_t* q = (uint16_t*)p; size_t n = sizeof(Probs) / 2, i; for (i = 0; i < n; i++) q[i] = 1024; }

static void rc_init(Rc* rc)
{
    int i;
    if (rc->inPos + 5 > rc->inEnd) { rc->err = 1; return; }
    if (rc->in[rc->inPos] != 0) { rc->err = 2; return; }        /* first byte of a range coder stream is 0 (LzmaDec.c:1100) */
    rc->code = 0; rc->range = 0xFFFFFFFFu;
    for (i = 1; i < 5; i++) rc->code = (rc->code << 8) | rc->in[rc->inPos + i];
    rc->inPos += 5;
}
static void rc_norm(Rc* rc)
{
    if (rc->range < (1u << 24)) {
        if (rc->inPos >= rc->inEnd) { rc->err = 3; rc->range <<= 8; rc->code <<= 8; return; }
        rc->range <<= 8; rc->code = (rc->code << 8) | rc->in[rc->inPos++];
    }
}
/* optional observation hooks (tools/lzma2_stats.c defines them before including this file; the checker builds without) */
#ifndef GCO_ON_BIT
#define GCO_ON_BIT(p, bit)
#define GCO_ON_DIRECT(n)
#define GCO_ON_SYM(kind, len, dist)
#endif
static unsigned rc_bit(Rc* rc, uint16_t* prob)
{
    uint32_t bound;
    rc_norm(rc);
    bound = (rc->range >> 11) * *prob;
    GCO_ON_BIT(*prob, rc->code >= bound);
    if (rc->code < bound) { rc->range = bound; *prob = (uint16_t)(*prob + ((2048 - *prob) >> 5)); return 0; }
    rc->range -= bound; rc->code -= bound; *prob = (uint16_t)(*prob - (*prob >> 5)); return 1;
}
static unsigned rc_direct(Rc* rc, unsigned n)
{
    unsigned v = 0;
    GCO_ON_DIRECT(n);
    while (n--) { rc_norm(rc); rc->range >>= 1; if (rc->code >= rc->range) { rc->code -= rc->range; v = (v << 1) | 1; } else v <<= 1; }
    return v;
}
static unsigned bittree(Rc* rc, uint16_t* probs, unsigned nbits)
{
    unsigned m = 1, i;
    for (i = 0; i < nbits; i++) m = (m << 1) | rc_bit(rc, &probs[m]);
    return m - (1u << nbits);
}
static unsigned bittree_rev(Rc* rc, uint16_t* probs, unsigned nbits)
{
    unsigned m = 1, v = 0, i;
    for (i = 0; i < nbits; i++) { unsigned b = rc_bit(rc, &probs[m]); m = (m << 1) | b; v |= b << i; }
    return v;
}
static unsigned len_decode(Rc* rc, uint16_t* lc, unsigned posState)
{
    if (!rc_bit(rc, &lc[LEN_CHOICE])) return bittree(rc, &lc[LEN_LOW + posState * 8], 3);
    if (!rc_bit(rc, &lc[LEN_CHOICE2])) return 8 + bittree(rc, &lc[LEN_MID + posState * 8], 3);
    return 16 + bittree(rc, &lc[LEN_HIGH], 8);
}

typedef struct { unsigned lc, lp, pb; uint32_t reps[4]; unsigned state; int propsSet; } Lz;

/* decode `usize` bytes of one LZMA chunk from in[inPos, inPos+csize) into dst at *outPos; dicStart = position of the last
 * dictionary reset (matches may not reach before it), dictSize = declared dictionary size */
static int lzma_chunk(Lz* z, Probs* P, const uint8_t* in, size_t inPos, size_t csize, uint8_t* dst, size_t* outPos, size_t usize,
                      size_t dicStart, uint32_t dictSize)
{
    Rc rc; size_t end = *outPos + usize, pos = *outPos;
    const unsigned pbMask = (1u << z->pb) - 1, lpMask = (1u << z->lp) - 1;
    rc.in = in; rc.inPos = inPos; rc.inEnd = inPos + csize; rc.err = 0;
    rc_init(&rc);
    if (rc.err) return 10 + rc.err;
    while (pos < end) {
        const size_t processed = pos - dicStart;
        const unsigned posState = (unsigned)processed & pbMask;
        if (!rc_bit(&rc, &P->isMatch[z->state][posState])) {
            const unsigned prev = processed ? dst[pos - 1] : 0;
            uint16_t* lp = &P->lit[0x300u * ((((unsigned)processed & lpMask) << z->lc) + (prev >> (8 - z->lc)))];
            unsigned sym = 1;
            if (z->state < kNumLitStates) { do sym = (sym << 1) | rc_bit(&rc, &lp[sym]); while (sym < 0x100); }
            else {
                unsigned matchByte = dst[pos - z->reps[0] - 1], offs = 0x100;
                do {
                    unsigned bit, mb;
                    matchByte <<= 1; mb = matchByte & offs;
                    bit = rc_bit(&rc, &lp[offs + mb + sym]);
                    sym = (sym << 1) | bit;
                    if (bit) offs &= matchByte; else offs &= ~matchByte;
                } while (sym < 0x100);
            }
            dst[pos++] = (uint8_t)sym;
            GCO_ON_SYM(0, 1, 0);
            z->state = z->state < 4 ? 0 : (z->state < 10 ? z->state - 3 : z->state - 6);
        } else {
            unsigned len; uint32_t dist; int kind = 1;
            if (!rc_bit(&rc, &P->isRep[z->state])) {
                unsigned slot, lenState;
                len = len_decode(&rc, P->lenC, posState);
                lenState = len < 4 ? len : 3;
                slot = bittree(&rc, P->posSlot[lenState], 6);
                if (slot < 4) dist = slot;
                else {
                    const unsigned nd = (slot >> 1) - 1;
                    dist = (2u | (slot & 1u)) << nd;
                    if (slot < 14) dist += bittree_rev(&rc, &P->specPos[dist - slot], nd);     /* probs + base - slot (- 1 + m with m >= 1) */
                    else { dist += rc_direct(&rc, nd - 4) << 4; dist += bittree_rev(&rc, P->align, 4); }
                }
                z->reps[3] = z->reps[2]; z->reps[2] = z->reps[1]; z->reps[1] = z->reps[0]; z->reps[0] = dist;
                z->state = z->state < kNumLitStates ? 7 : 10;
                if (dist == 0xFFFFFFFFu) return 20;       /* end marker is not used inside LZMA2 chunks */
            } else {
                if (!rc_bit(&rc, &P->isRepG0[z->state])) {
                    if (!rc_bit(&rc, &P->isRep0Long[z->state][posState])) {
                        if (pos - dicStart <= z->reps[0]) return 21;
                        dst[pos] = dst[pos - z->reps[0] - 1]; pos++;
                        GCO_ON_SYM(2, 1, z->reps[0]);
                        z->state = z->state < kNumLitStates ? 9 : 11;
                        continue;
                    }
                    kind = 3;
                } else {
                    uint32_t d;
                    if (!rc_bit(&rc, &P->isRepG1[z->state])) { d = z->reps[1]; kind = 4; }
                    else {
                        if (!rc_bit(&rc, &P->isRepG2[z->state])) { d = z->reps[2]; kind = 5; }
                        else { d = z->reps[3]; z->reps[3] = z->reps[2]; kind = 6; }
                        z->reps[2] = z->reps[1];
                    }
                    z->reps[1] = z->reps[0]; z->reps[0] = d;
                }
                len = len_decode(&rc, P->repLenC, posState);
                z->state = z->state < kNumLitStates ? 8 : 11;
            }
            len += 2; dist = z->reps[0];
            GCO_ON_SYM(kind, len, dist);
            if (pos - dicStart <= dist || dist >= dictSize) return 22;
            if (pos + len > end) return 23;                /* a match may not cross the chunk end (LzmaDec.c remainLen handling is for streaming) */
            while (len--) { dst[pos] = dst[pos - dist - 1]; pos++; }
        }
        if (rc.err) return 10 + rc.err;
    }
    rc_norm(&rc);
    if (rc.err) return 10 + rc.err;
    if (rc.inPos != rc.inEnd) return 30;                   /* Lzma2Dec.c: packSize must be consumed exactly */
    if (rc.code != 0) return 31;
    *outPos = pos;
    return 0;
}

/* diagnostics of the last failing call: chunk index, input / output position at the chunk start, sizes */
static size_t g_diag[6];
const size_t* gco_lzma2_last_diag(void) { return g_diag; }

/* cf. Lzma2Dec_GetOldProps / LZMA2_DIC_SIZE_FROM_PROP (Lzma2Dec.c:60-75) */
size_t gco_lzma2_decode(const uint8_t* in, size_t inSize, uint8_t* dst, size_t dstCap, unsigned char prop)
{
    static Probs P;                 /* 64 KB; the checker is single-threaded */
    Lz z; size_t ip = 0, op = 0, dicStart = 0, nChunk = 0;
    uint32_t dictSize;
    int needDictReset = 1, needProps = 1;
    if (prop > 40) return (size_t)-1 - 1;
    dictSize = prop == 40 ? 0xFFFFFFFFu : ((uint32_t)(2 | (prop & 1)) << (prop / 2 + 11));
    memset(&z, 0, sizeof(z));
    for (;; nChunk++) {
        unsigned ctl;
        if (ip >= inSize) return (size_t)-1 - 2;
        ctl = in[ip++];
        if (ctl == 0) return op;
        if (ctl < 0x80) {
            size_t n;
            if (ctl > 2) return (size_t)-1 - 3;
            if (ctl == 1) { dicStart = op; needDictReset = 0; }
            else if (needDictReset) return (size_t)-1 - 4;
            if (ip + 2 > inSize) return (size_t)-1 - 2;
            n = (((size_t)in[ip] << 8) | in[ip + 1]) + 1; ip += 2;
            if (ip + n > inSize || op + n > dstCap) return (size_t)-1 - 5;
            memcpy(dst + op, in + ip, n); ip += n; op += n;
            /* LzmaDec_InitDicAndState(initDic, initState = False): the coder state survives a stored chunk */
        } else {
            const unsigned reset = (ctl >> 5) & 3;
            size_t usize, csize; int r;
            if (ip + 4 > inSize) return (size_t)-1 - 2;
            usize = ((size_t)(ctl & 0x1F) << 16) + ((size_t)in[ip] << 8) + in[ip + 1] + 1;
            csize = ((size_t)in[ip + 2] << 8) + in[ip + 3] + 1; ip += 4;
            if (reset == 3) { dicStart = op; needDictReset = 0; }
            else if (needDictReset) return (size_t)-1 - 4;
            if (reset >= 2) {
                unsigned pr;
                if (ip >= inSize) return (size_t)-1 - 2;
                pr = in[ip++];
                if (pr >= 9 * 5 * 5) return (size_t)-1 - 6;
                z.lc = pr % 9; pr /= 9; z.lp = pr % 5; z.pb = pr / 5;
                if (z.lc + z.lp > 4) return (size_t)-1 - 6;
                needProps = 0;
            } else if (needProps) return (size_t)-1 - 7;
            if (reset >= 1) { probs_init(&P); z.state = 0; z.reps[0] = z.reps[1] = z.reps[2] = z.reps[3] = 0; }
            if (ip + csize > inSize || op + usize > dstCap) return (size_t)-1 - 5;
            r = lzma_chunk(&z, &P, in, ip, csize, dst, &op, usize, dicStart, dictSize);
            if (r) { g_diag[0] = nChunk; g_diag[1] = ip; g_diag[2] = op; g_diag[3] = usize; g_diag[4] = csize; g_diag[5] = (size_t)r; return (size_t)-1 - 100 - (size_t)r; }
            ip += csize;
        }
    }
}
