// zstd_parse_lab.c -- CPU lab (NOT product, NOT test infrastructure): what a change of the level-3 finder's SHAPE would cost in size before it is written for the device.
//
// Models the candidate sources of the windowed finder (gc_lz_window.hip: per position the most recent earlier position of its 8 MiB frame with the same 8-byte
// and the same 5-byte hash, direct-mapped tables of 2^20 + 2^19 slots per frame), the greedy + one-step-lazy parse of W6 (pz_seg) and an order-0 estimate of what
// the zstd entropy stage makes of the result (literals: byte entropy per 128 KiB block; sequences: entropy of the LL / ML / OF codes + their extra bits, repeat
// offsets as ZSTD_updateRep does, C/zstd/zstd_compress_internal.h:818).  Sizes are estimates: only differences between policies on the same bytes mean anything.
//
//   policy 0  the shipped shape: every position listed in the frame-wide tables
//   policy 1  two tiers (round 6): NEAR = exact most-recent tables over [tile start - H, p) for every position (LDS tables per 8 KiB tile, H bytes of history staged with
//             the tile), FAR = the frame-wide tables over a content-defined SAMPLE of the positions (1 in 2^R, chosen by the position's 5-byte hash: both ends of a repeat
//             are sampled or neither), a far match extended up to BK bytes backwards onto the positions in front (C/zstd/zstd_ldm.c:34-56 samples the same way in front of
//             the block matchers)
//   usage: zstd_parse_lab file [policy] [H] [nearLogL] [nearLogS] [R] [BK] [unit]      (unit != 0: near tables live for `unit` bytes (a workgroup takes that many tiles in order))
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define BLK (128u * 1024u)
#define FRAME (8u << 20)
#define TILE 8192u
static const uint8_t* S; static size_t N;
static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t hL(const uint8_t* p) { return ld32(p) * 0x9E3779B1u + ld32(p + 4) * 0x85EBCA77u; }
static inline uint32_t hS(const uint8_t* p) { return ld32(p) * 0x9E3779B1u + (ld32(p + 4) & 0xFFu) * 0xC2B2AE3Du; }
static inline int hib(uint32_t x) { return 31 - __builtin_clz(x); }
static inline int gain(uint32_t len, uint32_t off) { return (int)(len * 4u) - hib(off + 1u); }
static inline uint32_t mlen(size_t a, size_t b, uint32_t maxLen) { uint32_t l = 0; while (l < maxLen && S[a + l] == S[b + l]) l++; return l; }

typedef struct { uint32_t off, len; } Rec;
static Rec* rec;

static void consider(size_t p, size_t c, uint32_t maxLen, uint32_t minLen)
{
    if (c >= p) return;
    const uint32_t l = mlen(p, c, maxLen);
    if (l < minLen) return;
    const uint32_t off = (uint32_t)(p - c);
    if (rec[p].len == 0 || gain(l, off) > gain(rec[p].len, rec[p].off)) { rec[p].off = off; rec[p].len = l; }
}

// ---- cost model
static const uint8_t LLc[64] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 };
static const uint8_t MLc[128] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
    40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42 };
static const uint8_t LLb[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const uint8_t MLb[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
static inline uint32_t llcode(uint32_t ll) { return ll > 63 ? (uint32_t)hib(ll) + 19 : LLc[ll]; }
static inline uint32_t mlcode(uint32_t mlb) { return mlb > 127 ? (uint32_t)hib(mlb) + 36 : MLc[mlb]; }
static double ent(const uint32_t* h, int n) { uint64_t t = 0; for (int i = 0; i < n; i++) t += h[i]; if (!t) return 0; double b = 0; for (int i = 0; i < n; i++) if (h[i]) b += h[i] * log2((double)t / h[i]); return b; }

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    FILE* f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); N = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t* buf = malloc(N + 64); if (fread(buf, 1, N, f) != N) return 1; memset(buf + N, 0, 64); fclose(f); S = buf;
    const int policy = argc > 2 ? atoi(argv[2]) : 0;
    const uint32_t H = argc > 3 ? (uint32_t)atoi(argv[3]) : 8192u;
    const uint32_t nLL = argc > 4 ? (uint32_t)atoi(argv[4]) : 12u, nLS = argc > 5 ? (uint32_t)atoi(argv[5]) : 12u;
    const uint32_t R = argc > 6 ? (uint32_t)atoi(argv[6]) : 3u, BK = argc > 7 ? (uint32_t)atoi(argv[7]) : 15u;
    const uint32_t unit = argc > 8 ? (uint32_t)atoi(argv[8]) : 0u;
    const uint32_t farMin = argc > 9 ? (uint32_t)atoi(argv[9]) : 5u;
    rec = calloc(N + 1, sizeof(Rec));
    uint32_t* fL = malloc(sizeof(uint32_t) << 20); uint32_t* fS = malloc(sizeof(uint32_t) << 19);
    uint32_t* nL = malloc(sizeof(uint32_t) << nLL); uint32_t* nS = malloc(sizeof(uint32_t) << nLS);
    uint64_t nSampled = 0, nListed = 0;
    for (size_t fs = 0; fs < N; fs += FRAME) {
        const size_t fe = fs + FRAME < N ? fs + FRAME : N;
        memset(fL, 0xFF, sizeof(uint32_t) << 20); memset(fS, 0xFF, sizeof(uint32_t) << 19);
        // ---- far / frame-wide tier
        for (size_t p = fs; p + 64 + 16 <= fe; p++) {
            const size_t bend = (p / BLK + 1) * BLK < fe ? (p / BLK + 1) * BLK : fe;
            const uint32_t maxLen = (uint32_t)(bend - p);
            if (p > fs && ld64(S + p) == ((ld64(S + p) << 8) | S[p - 1])) { consider(p, p - 1, maxLen, 5); continue; }      // byte run: candidate P - 1, not listed
            const uint32_t a = hL(S + p), b = hS(S + p);
            if ((policy == 1 || policy == 4) && ((b ^ (b >> 15)) * 0x2C1B3C6Du) >> (32u - R) != 0u) continue;      // (policy 4: the sampled listing alone + the extension backwards)
            nListed++;
            const uint32_t ia = a >> 12, ib = b >> 13;
            const uint32_t cL = fL[ia], cS = fS[ib];
            fL[ia] = (uint32_t)(p - fs); fS[ib] = (uint32_t)(p - fs);
            Rec before = rec[p]; rec[p].len = 0;
            if (cL != 0xFFFFFFFFu) consider(p, fs + cL, maxLen, policy == 1 ? farMin : 5);
            if (rec[p].len < 8 && cS != 0xFFFFFFFFu) consider(p, fs + cS, maxLen, policy == 1 ? farMin : 5);
            if ((policy == 1 || policy == 3 || policy == 4) && rec[p].len) {      // (policy 3: the shipped shape + every match extended backwards onto the positions in front -- ZSTD_compressBlock_doubleFast's catch-up, zstd_double_fast.c:255-262, made a candidate of the earlier position)
                nSampled++;
                const Rec far = rec[p];
                for (uint32_t k = 1; k <= BK; k++) {                  // backwards onto the positions in front (inside the block and the frame)
                    if (p < fs + k + far.off || (p - k) / BLK != p / BLK) break;
                    if (S[p - k] != S[p - k - far.off]) break;
                    Rec* r = &rec[p - k];
                    if (r->len == 0 || gain(far.len + k, far.off) > gain(r->len, r->off)) { r->off = far.off; r->len = far.len + k; }
                }
            }
            if (before.len && (rec[p].len == 0 || gain(before.len, before.off) > gain(rec[p].len, rec[p].off))) rec[p] = before;      // (an extension from a later... earlier sampled position)
        }
        if (policy != 1) continue;
        // ---- near tier: exact most-recent tables over [start - H, p)
        const uint32_t step = unit ? unit : TILE;
        for (size_t ts = fs; ts < fe; ts += step) {
            const size_t te = ts + step < fe ? ts + step : fe;
            memset(nL, 0xFF, sizeof(uint32_t) << nLL); memset(nS, 0xFF, sizeof(uint32_t) << nLS);
            const size_t h0 = ts >= fs + H ? ts - H : fs;
            for (size_t p = h0; p < te && p + 64 + 16 <= fe; p++) {
                const uint32_t a = hL(S + p), b = hS(S + p);
                const uint32_t ia = a >> (32u - nLL), ib = b >> (32u - nLS);
                const uint32_t cL = nL[ia], cS = nS[ib];
                nL[ia] = (uint32_t)(p - fs); nS[ib] = (uint32_t)(p - fs);
                if (p < ts) continue;
                if (p > fs && ld64(S + p) == ((ld64(S + p) << 8) | S[p - 1])) continue;
                const size_t bend = (p / BLK + 1) * BLK < fe ? (p / BLK + 1) * BLK : fe;
                const uint32_t maxLen = (uint32_t)(bend - p);
                const uint32_t l0 = rec[p].len;
                if (cL != 0xFFFFFFFFu) consider(p, fs + cL, maxLen, 5);
                if (cS != 0xFFFFFFFFu) consider(p, fs + cS, maxLen, 5);
                (void)l0;
            }
        }
    }
    // ---- parse (greedy + one-step lazy, pz_seg) and cost per block
    double bits = 0; uint64_t nSeq = 0, nLit = 0;
    uint32_t rep[3] = { 1, 4, 8 };
    for (size_t bs = 0; bs < N; bs += BLK) {
        const size_t be = bs + BLK < N ? bs + BLK : N;
        if (bs % FRAME == 0) { rep[0] = 1; rep[1] = 4; rep[2] = 8; }
        uint32_t hl[36] = { 0 }, hm[53] = { 0 }, ho[32] = { 0 }, hb[256] = { 0 };
        double extra = 0;
        size_t p = bs, litStart = bs;
        while (p < be) {
            const Rec r0 = rec[p];
            int take = r0.len != 0;
            if (take && p + 1 < be) { const Rec r1 = rec[p + 1]; if (r1.len > r0.len && gain(r1.len, r1.off) > gain(r0.len, r0.off) + 4) take = 0; }
            if (!take) { hb[S[p]]++; p++; continue; }
            uint32_t len = r0.len; if (p + len > be) len = (uint32_t)(be - p);
            const uint32_t ll = (uint32_t)(p - litStart);
            uint32_t offBase;
            if (ll) {
                if (r0.off == rep[0]) offBase = 1; else if (r0.off == rep[1]) { offBase = 2; rep[1] = rep[0]; rep[0] = r0.off; }
                else if (r0.off == rep[2]) { offBase = 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = r0.off; } else { offBase = r0.off + 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = r0.off; }
            } else {
                if (r0.off == rep[1]) { offBase = 1; rep[1] = rep[0]; rep[0] = r0.off; } else if (r0.off == rep[2]) { offBase = 2; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = r0.off; }
                else if (r0.off == rep[0] - 1 && rep[0] > 1) { offBase = 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = r0.off; } else { offBase = r0.off + 3; rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = r0.off; }
            }
            const uint32_t lc = llcode(ll), mc = mlcode(len - 3), oc = (uint32_t)hib(offBase);
            hl[lc]++; hm[mc]++; ho[oc]++; extra += LLb[lc] + MLb[mc] + oc;
            nSeq++;
            p += len; litStart = p;
        }
        uint64_t lits = 0; for (int i = 0; i < 256; i++) lits += hb[i];
        nLit += lits;
        bits += ent(hb, 256) + ent(hl, 36) + ent(hm, 53) + ent(ho, 32) + extra + 8.0 * 120;
    }
    printf("%s policy %d H %u near 2^%u/2^%u R %u BK %u unit %u farMin %u: est %.0f bytes (%.4f of input), %llu seqs, %llu lits, listed %.3f of positions, far matches %llu\n", argv[1], policy, H, nLL, nLS, R, BK, unit, farMin,
           bits / 8, bits / 8 / N, (unsigned long long)nSeq, (unsigned long long)nLit, (double)nListed / N, (unsigned long long)nSampled);
    return 0;
}
