/* tools/lzma_parse_lab.c -- RESEARCH TOOL (not product, not oracle): what does each parse / candidate / reset choice of the
 * FLZMA2 GPU path cost in compressed size?  Simulates the GPU finder's candidates on the CPU (two "most recent wins" hash
 * tables over 8 MiB frames, 64-byte compare cap, link following), runs several parse strategies over them and prices the result
 * with an exact LZMA model (lc3 lp0 pb2, adaptive 11-bit probabilities; cost = sum of -log2 p).  Sizes are printed next to
 * each other so that a design can be chosen before any kernel is written.
 *   gcc -O2 -o /tmp/lab tools/lzma_parse_lab.c -lm && /tmp/lab file [segLog] [mode...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>

typedef uint8_t u8; typedef uint32_t u32; typedef uint64_t u64;
static u8* S; static u32 N;

/* ---------------------------------------------------------------- LZMA model */
#define NPROB (1380 + (0x300 << 3))
#define P_ISMATCH 0
#define P_ISREP 48
#define P_ISREPG0 60
#define P_ISREPG1 72
#define P_ISREPG2 84
#define P_ISREP0LONG 96
#define P_LEN 144
#define P_REPLEN 466
#define P_POSSLOT 788
#define P_SPECPOS 1044
#define P_ALIGN 1364
#define P_LIT 1380
static float g_cost[2048 + 1];          /* -log2(p/2048) */
typedef struct { uint16_t p[NPROB]; } Model;
static void model_reset(Model* m) { for (int i = 0; i < NPROB; i++) m->p[i] = 1024; }
typedef struct { Model* m; int update; double bits; } Enc;
int g_counting = 0; static u32 cnt0[NPROB], cnt1[NPROB];
static inline void ebit(Enc* e, u32 idx, u32 bit)
{
    u32 p = e->m->p[idx];
    if (g_counting) { if (bit) cnt1[idx]++; else cnt0[idx]++; }
    e->bits += bit ? g_cost[2048 - p] : g_cost[p];
    if (e->update) e->m->p[idx] = (uint16_t)(bit ? p - (p >> 5) : p + ((2048 - p) >> 5));
}
static void e_len(Enc* e, u32 base, u32 len, u32 ps)
{
    u32 v = len - 2;
    if (v < 8) { ebit(e, base, 0); u32 tb = base + 2 + ps * 8, m = 1; for (int i = 2; i >= 0; i--) { u32 b = (v >> i) & 1; ebit(e, tb + m, b); m = (m << 1) | b; } }
    else if (v < 16) { ebit(e, base, 1); ebit(e, base + 1, 0); v -= 8; u32 tb = base + 34 + ps * 8, m = 1; for (int i = 2; i >= 0; i--) { u32 b = (v >> i) & 1; ebit(e, tb + m, b); m = (m << 1) | b; } }
    else { ebit(e, base, 1); ebit(e, base + 1, 1); v -= 16; u32 tb = base + 66, m = 1; for (int i = 7; i >= 0; i--) { u32 b = (v >> i) & 1; ebit(e, tb + m, b); m = (m << 1) | b; } }
}
static inline u32 hibit(u32 x) { return 31 - __builtin_clz(x); }
static void e_dist(Enc* e, u32 dist /* distance-1 */, u32 len)
{
    u32 ls = len - 2 < 3 ? len - 2 : 3, slot;
    if (dist < 4) slot = dist; else { u32 hb = hibit(dist); slot = 2 * hb + ((dist >> (hb - 1)) & 1); }
    u32 tb = P_POSSLOT + ls * 64, m = 1;
    for (int i = 5; i >= 0; i--) { u32 b = (slot >> i) & 1; ebit(e, tb + m, b); m = (m << 1) | b; }
    if (slot >= 4) {
        u32 footer = (slot >> 1) - 1, base = (2 | (slot & 1)) << footer, red = dist - base;
        if (slot < 14) { u32 t2 = P_SPECPOS + (slot - 4) * 32; m = 1; for (u32 i = 0; i < footer; i++) { u32 b = (red >> i) & 1; ebit(e, t2 + m, b); m = (m << 1) | b; } }
        else { e->bits += footer - 4; m = 1; for (u32 i = 0; i < 4; i++) { u32 b = (red >> i) & 1; ebit(e, P_ALIGN + m, b); m = (m << 1) | b; } }
    }
}
static void e_lit(Enc* e, u32 cur, u32 prev, u32 mb, u32 st)
{
    u32 pb = P_LIT + 0x300 * (prev >> 5);
    if (st < 7) { u32 m = 1; for (int i = 7; i >= 0; i--) { u32 b = (cur >> i) & 1; ebit(e, pb + m, b); m = (m << 1) | b; } }
    else { u32 offs = 0x100, sym = cur | 0x100; for (int i = 0; i < 8; i++) { mb <<= 1; ebit(e, pb + offs + (mb & offs) + (sym >> 8), (sym >> 7) & 1); sym <<= 1; offs &= ~(mb ^ sym); } }
}
typedef struct { u32 st; u32 rep[4]; } St;
static inline u32 st_lit(u32 s) { return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6); }
/* symbol kinds: 0 literal, 1 match (dist = offset), 2 shortrep, 3+k rep k */
static void e_symbol(Enc* e, St* s, u32 pos, u32 kind, u32 len, u32 off)
{
    u32 ps = pos & 3;
    if (kind == 0) {
        ebit(e, P_ISMATCH + s->st * 4 + ps, 0);
        e_lit(e, S[pos], pos ? S[pos - 1] : 0, s->st >= 7 ? S[pos - s->rep[0]] : 0, s->st);
        s->st = st_lit(s->st); return;
    }
    ebit(e, P_ISMATCH + s->st * 4 + ps, 1);
    if (kind == 1) {
        ebit(e, P_ISREP + s->st, 0); e_len(e, P_LEN, len, ps); e_dist(e, off - 1, len);
        s->rep[3] = s->rep[2]; s->rep[2] = s->rep[1]; s->rep[1] = s->rep[0]; s->rep[0] = off; s->st = s->st < 7 ? 7 : 10; return;
    }
    ebit(e, P_ISREP + s->st, 1);
    if (kind == 2) { ebit(e, P_ISREPG0 + s->st, 0); ebit(e, P_ISREP0LONG + s->st * 4 + ps, 0); s->st = s->st < 7 ? 9 : 11; return; }
    u32 k = kind - 3;
    if (k == 0) { ebit(e, P_ISREPG0 + s->st, 0); ebit(e, P_ISREP0LONG + s->st * 4 + ps, 1); }
    else { ebit(e, P_ISREPG0 + s->st, 1);
        if (k == 1) ebit(e, P_ISREPG1 + s->st, 0);
        else { ebit(e, P_ISREPG1 + s->st, 1); ebit(e, P_ISREPG2 + s->st, k - 2); }
        u32 d = s->rep[k]; for (u32 i = k; i > 0; i--) s->rep[i] = s->rep[i - 1]; s->rep[0] = d; }
    e_len(e, P_REPLEN, len, ps);
    s->st = s->st < 7 ? 8 : 11;
}

/* ---------------------------------------------------------------- simulated finder */
#define CAP 64
#define FRAME (8u << 20)
static u32 *recOff; static u8 *recLen;            /* best candidate per position (len <= CAP) */
static u32 *rec3Off; static u8 *rec3Len;          /* optional short candidate (hash of 3 bytes, nearest) */
static u32 *rec2Off; static u8 *rec2Len;          /* LAB_TWO=1: second-best main candidate (by gain, other distance) */
static inline u32 mlen(u32 a, u32 b, u32 max) { u32 l = 0; while (l < max && S[a + l] == S[b + l]) l++; return l; }
static inline int gain(u32 len, u32 off) { return (int)(len * 4) - (int)hibit(off + 1); }
static void finder(int depth, int minMatch, int use3)
{
    recOff = calloc(N, 4); recLen = calloc(N, 1); rec3Off = calloc(N, 4); rec3Len = calloc(N, 1); rec2Off = calloc(N, 4); rec2Len = calloc(N, 1);
    int two = getenv("LAB_TWO") ? atoi(getenv("LAB_TWO")) : 0;   /* 1: second by gain; 2: the LONGEST other candidate */
    u32* tL = malloc(4u << 20), *tS = malloc(4u << 19), *t3 = malloc(4u << 16);
    int nx = 0, xk[16]; u32* tX[16]; int tbits = getenv("LAB_TBITS") ? atoi(getenv("LAB_TBITS")) : 20;              /* LAB_MULTI=12,16,24: extra "most recent wins" tables keyed on longer prefixes */
    if (getenv("LAB_MULTI")) { const char* e = getenv("LAB_MULTI"); while (*e && nx < 16) { xk[nx] = atoi(e); tX[nx] = malloc((size_t)4u << tbits); nx++; while (*e && *e != ',') e++; if (*e) e++; } }
    for (u32 f = 0; f < N; f += FRAME) {
        u32 fe = f + FRAME < N ? f + FRAME : N;
        memset(tL, 0xFF, 4u << 20); memset(tS, 0xFF, 4u << 19); memset(t3, 0xFF, 4u << 16);
        for (int j = 0; j < nx; j++) memset(tX[j], 0xFF, (size_t)4u << tbits);
        for (u32 p = f; p + CAP + 16 <= fe; p++) {
            u64 x; memcpy(&x, S + p, 8);
            u32 lo = (u32)x, hi = (u32)(x >> 32);
            u32 h3 = ((lo & 0xFFFFFF) * 0x9E3779B1u) >> 16;
            if (use3) { u32 c = t3[h3]; if (c != 0xFFFFFFFFu) { u32 l = mlen(p, c, 16); if (l >= 2) { rec3Off[p] = p - c; rec3Len[p] = l; } } t3[h3] = p; }
            int run = p > f && (((x << 8) | S[p - 1]) == x);
            if (run) { recOff[p] = 1; recLen[p] = mlen(p, p - 1, CAP); continue; }
            u32 hL = (lo * 0x9E3779B1u + hi * 0x85EBCA77u) >> 12, hS = (lo * 0x9E3779B1u + (hi & 0xFF) * 0xC2B2AE3Du) >> 13;
            u32 cL = tL[hL], cS = tS[hS]; tL[hL] = p; tS[hS] = p;
            u32 bl = 0, bo = 0;
            if (getenv("LAB_NOBASE")) cL = cS = 0xFFFFFFFFu;
            if (cL != 0xFFFFFFFFu) { u32 l = mlen(p, cL, CAP); if (l >= (u32)minMatch) { bl = l; bo = p - cL; } }
            if (bl < 8 && cS != 0xFFFFFFFFu && cS != cL) { u32 l = mlen(p, cS, CAP); if (l >= (u32)minMatch && (bl == 0 || gain(l, p - cS) > gain(bl, bo))) { bl = l; bo = p - cS; } }
            for (int j = 0; j < nx; j++) {
                u64 h = 0xcbf29ce484222325ull; for (int q = 0; q < xk[j]; q++) h = (h ^ S[p + q]) * 0x100000001b3ull;
                u32 hx = (u32)(h >> (64 - tbits)), c = tX[j][hx]; tX[j][hx] = p;
                if (c != 0xFFFFFFFFu) { u32 l = mlen(p, c, CAP); if (l >= (u32)minMatch) {
                    u32 o = p - c;
                    if (bl == 0 || gain(l, o) > gain(bl, bo)) { if (two && bl && bo != o && (two == 1 || bl > rec2Len[p])) { rec2Len[p] = bl; rec2Off[p] = bo; } bl = l; bo = o; }
                    else if (two && o != bo && (rec2Len[p] == 0 || (two == 1 ? gain(l, o) > gain(rec2Len[p], rec2Off[p]) : l > rec2Len[p]))) { rec2Len[p] = l; rec2Off[p] = o; } } }
            }
            recOff[p] = bo; recLen[p] = bl;
        }
    }
    if (depth) {                                   /* W5b: follow links */
        u32* o2 = malloc(4u * N); u8* l2 = malloc(N);
        for (u32 p = 0; p < N; p++) {
            u32 bl = recLen[p], bo = recOff[p];
            if (bl && bo != 1) { int bg = gain(bl, bo); u32 c = p - bo; u32 fs = p / FRAME * FRAME;
                for (int d = 0; d < depth; d++) { if (!recLen[c]) break; u32 c2 = c - recOff[c]; if (c2 < fs) break; u32 l = mlen(p, c2, CAP); if (l >= (u32)minMatch) { int g = gain(l, p - c2); if (g > bg) { bg = g; bl = l; bo = p - c2; } } c = c2; } }
            o2[p] = bo; l2[p] = bl;
        }
        free(recOff); free(recLen); recOff = o2; recLen = l2;
    }
    free(tL); free(tS); free(t3);
    if (getenv("LAB_INHERIT")) {                   /* left-neighbour inheritance: the match that covers p-1 also covers p */
        u32 cnt = 0;
        for (u32 p = 1; p < N; p++) { u32 l = recLen[p - 1], d = recOff[p - 1]; if (l < 3 || (p % FRAME) == 0) continue; u32 l1 = l - 1;
            if (l == CAP) l1 = mlen(p, p - d, CAP);   /* capped: the real remainder */
            if (l1 >= 2 && (recLen[p] == 0 || gain(l1, d) > gain(recLen[p], recOff[p]))) { recLen[p] = l1; recOff[p] = d; cnt++; } }
        fprintf(stderr, "   inherited %u records\n", cnt);
    }
}

/* reference-like candidates: longest match (nearest among equals) by a deep hash-chain search on 3 bytes; replaces rec[] */
static void finder_hc(int depth)
{
    u32* head = malloc(4u << 16), *chain = malloc(4u * N);
    for (u32 f = 0; f < N; f += FRAME) {
        u32 fe = f + FRAME < N ? f + FRAME : N;
        memset(head, 0xFF, 4u << 16);
        for (u32 p = f; p + 4 <= fe; p++) {
            u32 h = ((S[p] | (S[p + 1] << 8) | (S[p + 2] << 16)) * 0x9E3779B1u) >> 16;
            u32 c = head[h]; chain[p] = c; head[h] = p;
            u32 bl = 0, bo = 0, max = fe - p < 273 ? fe - p : 273;
            for (int d = 0; d < depth && c != 0xFFFFFFFFu; d++, c = chain[c]) { u32 l = mlen(p, c, max); if (l > bl) { bl = l; bo = p - c; if (l == max) break; } }
            if (bl >= 3) { recOff[p] = bo; recLen[p] = bl > CAP ? CAP : bl; } else { recOff[p] = 0; recLen[p] = 0; }
        }
    }
    free(head); free(chain);
}


/* exact candidates: for every position the LONGEST earlier match inside its frame (compare cap CAP), the nearest one among equally
 * long ones -- what RMF_buildTable resolves to (radix_engine.h:920: depth 42 at level 5, then extended) -- from a suffix sort of
 * the frame by the first CAP bytes (ties by position) and nearest-smaller-position scans in rank order.  LAB_LPM=1 */
static const u8* g_sortBase; static u32 g_sortEnd;
static int lpm_cmp(const void* a, const void* b)
{
    u32 x = *(const u32*)a, y = *(const u32*)b;
    u32 mx = g_sortEnd - x < CAP ? g_sortEnd - x : CAP, my = g_sortEnd - y < CAP ? g_sortEnd - y : CAP, m = mx < my ? mx : my;
    int c = memcmp(g_sortBase + x, g_sortBase + y, m);
    if (c) return c;
    if (mx != my) return mx < my ? -1 : 1;
    return x < y ? -1 : 1;
}
static void finder_lpm(int minMatch)
{
    for (u32 f = 0; f < N; f += FRAME) {
        u32 fe = f + FRAME < N ? f + FRAME : N, n = fe - f;
        u32* sa = malloc(4u * n); u8* lcp = malloc(n);            /* lcp[r] = common prefix of sa[r-1], sa[r], capped */
        for (u32 i = 0; i < n; i++) sa[i] = f + i;
        g_sortBase = S; g_sortEnd = fe;
        qsort(sa, n, 4, lpm_cmp);
        lcp[0] = 0;
        for (u32 r = 1; r < n; r++) { u32 a = sa[r - 1], b = sa[r], m = fe - (a > b ? a : b); if (m > CAP) m = CAP; lcp[r] = (u8)mlen(a, b, m); }
        /* previous smaller position in rank order, with the minimum lcp on the way: stack scan; then the same from the other side */
        u32* bestPos = malloc(4u * n); u8* bestLen = calloc(n, 1);
        u32* stR = malloc(4u * n); u8* stL = malloc(n); u32 sp;
        for (int dir = 0; dir < 2; dir++) {
            sp = 0;
            for (u32 k = 0; k < n; k++) {
                u32 r = dir ? n - 1 - k : k;
                /* lcp between rank r and the previous visited rank */
                u32 l = k == 0 ? 0 : (dir ? lcp[r + 1] : lcp[r]);
                /* stack holds ranks with increasing positions...: pop entries whose position is larger than ours, carrying the min lcp */
                u32 cur = l;
                while (sp && sa[stR[sp - 1]] > sa[r]) { if (stL[sp - 1] < cur) cur = stL[sp - 1]; sp--; }
                if (sp) { u32 q = sa[stR[sp - 1]]; u32 p = sa[r]; u32 i = p - f;
                    if (cur > bestLen[i] || (cur == bestLen[i] && cur && q > bestPos[i])) { bestLen[i] = (u8)cur; bestPos[i] = q; } }
                /* push: the lcp stored with an entry is the min lcp between it and the entry below it */
                stR[sp] = r; stL[sp] = (u8)cur; sp++;
            }
        }
        for (u32 i = 0; i < n; i++) { u32 p = f + i; if (p + CAP + 16 > fe) { recOff[p] = 0; recLen[p] = 0; continue; }
            if (bestLen[i] >= (u32)minMatch) { recOff[p] = p - bestPos[i]; recLen[p] = bestLen[i]; } else { recOff[p] = 0; recLen[p] = 0; } }
        free(sa); free(lcp); free(bestPos); free(bestLen); free(stR); free(stL);
    }
}

/* ---------------------------------------------------------------- parses: produce symbol list */
typedef struct { u32 pos; u32 len; u32 off; } Sym;   /* len 0 = literal run marker not used; we list only matches */
static Sym* syms; static u32 nSyms;
static void push(u32 pos, u32 len, u32 off) { syms[nSyms].pos = pos; syms[nSyms].len = len; syms[nSyms].off = off; nSyms++; }

/* W6: greedy + lazy over capped records; chains of capped records merge into one match (cut at 273 later) */
static void parse_greedy(int lazy)
{
    nSyms = 0;
    u32 p = 0;
    while (p < N) {
        u32 len = recLen[p];
        int take = len != 0;
        if (take && p + 1 < N) { u32 l1 = recLen[p + 1]; if (l1 > len && gain(l1, recOff[p + 1]) > gain(len, recOff[p]) + 4) take = 0; }
        if (take && lazy >= 2 && p + 2 < N) { u32 l2 = recLen[p + 2]; if (l2 > len + 1 && gain(l2, recOff[p + 2]) > gain(len, recOff[p]) + 8) take = 0; }
        if (!take) { p++; continue; }
        if (nSyms && syms[nSyms - 1].pos + syms[nSyms - 1].len == p && syms[nSyms - 1].off == recOff[p]) syms[nSyms - 1].len += len; else push(p, len, recOff[p]);
        p += len;
    }
}

/* ---------------------------------------------------------------- pricing a symbol list with the real model */
static double price_syms(int segLog, int rcLog, double* hdrBytes)
{
    static Model m; Enc e; e.m = &m; e.update = 1; e.bits = 0;
    { u32 j = 0; for (u32 i = 0; i < nSyms; i++) { if (j && syms[j - 1].pos + syms[j - 1].len == syms[i].pos && syms[j - 1].off == syms[i].off) syms[j - 1].len += syms[i].len; else syms[j++] = syms[i]; } nSyms = j; }   /* L1 merges adjacent pieces of one match */
    St s; u32 p = 0, k = 0; u32 segSize = 1u << segLog;
    u32 nRep = 0, nMatch = 0, nLit = 0, nShort = 0;
    for (u32 ss = 0; ss < N; ss += segSize) {
        u32 se = ss + segSize < N ? ss + segSize : N;
        model_reset(&m); s.st = 0; s.rep[0] = s.rep[1] = s.rep[2] = s.rep[3] = 1;
        while (p < se) {
            if (k < nSyms && syms[k].pos == p) {
                u32 len = syms[k].len, off = syms[k].off;
                u32 lim = se - p; if (len > lim) { syms[k].pos += lim; syms[k].len -= lim; len = lim; } else k++;
                while (len) {
                    u32 take = len < 273 ? len : 273; if (len - take == 1) take--;
                    if (take < 2) { e_symbol(&e, &s, p, 0, 0, 0); p++; len--; nLit++; continue; }
                    u32 kind = 1; for (u32 r = 0; r < 4; r++) if (s.rep[r] == off) { kind = 3 + r; break; }
                    if (kind == 1) nMatch++; else nRep++;
                    e_symbol(&e, &s, p, kind, take, off); p += take; len -= take;
                }
            } else if (k < nSyms && syms[k].pos < p) { k++; }
            else { e_symbol(&e, &s, p, 0, 0, 0); p++; nLit++; }
        }
    }
    u32 nRc = (N + (1u << rcLog) - 1) >> rcLog;
    *hdrBytes = nRc * 10.0;
    fprintf(stderr, "   [lit %u match %u rep %u short %u]\n", nLit, nMatch, nRep, nShort);
    return e.bits / 8.0;
}

/* ---------------------------------------------------------------- rep-aware refinement (parallelisable design)
 * Given the path of a first parse: rep0(p) = offset of the last path match that ends at or before p.  Every position gets a
 * rep candidate (length of the match at distance rep0(p), >= 2) and the record is replaced when LZMA's fast-mode rules
 * prefer the rep.  Then the greedy parse is run again. */
static u32 *repOff; static u8* repLen;
static void refine_records(void)
{
    if (!repOff) { repOff = calloc(N, 4); repLen = calloc(N, 1); }
    memset(repLen, 0, N);
    u32 k = 0, r0 = 0, r1 = 0;
    for (u32 p = 0; p < N; p++) {
        while (k < nSyms && syms[k].pos + syms[k].len <= p) { if (syms[k].off != r0) { r1 = r0; r0 = syms[k].off; } k++; }
        u32 fs = p / FRAME * FRAME;
        u32 best = 0, bo = 0;
        if (r0 && p >= fs + r0) { u32 l = mlen(p, p - r0, (N - p) < CAP ? N - p : CAP); if (l >= 2) { best = l; bo = r0; } }
        if (r1 && p >= fs + r1) { u32 l = mlen(p, p - r1, (N - p) < CAP ? N - p : CAP); if (l >= 2 && l > best + 1) { best = l; bo = r1; } }
        repLen[p] = best; repOff[p] = bo;
    }
    for (u32 p = 0; p < N; p++) {
        u32 rl = repLen[p], ml = recLen[p], mo = recOff[p];
        if (!rl) continue;
        if (mo == repOff[p]) continue;
        int pick = 0;
        if (ml == 0) pick = 1;
        else if (rl + 1 >= ml) pick = 1;
        else if (rl + 2 >= ml && mo >= (1u << 9)) pick = 1;
        else if (rl + 3 >= ml && mo >= (1u << 15)) pick = 1;
        if (pick) { recLen[p] = rl; recOff[p] = repOff[p]; }
    }
}

/* ---------------------------------------------------------------- optimal parse (upper bound for these candidates) */
typedef struct { float cost; u32 prev; u32 len; u32 off; u8 kind; St s; } Node;
static u32 full_len(u32 p, u32 off, u32 have, u32 lim) { u32 l = have; while (l < lim && S[p + l] == S[p + l - off]) l++; return l; }
static Sym* g_optG; static u32 g_optNG;      /* LAB_OPT_STATIC: a greedy parse whose block statistics price the optimal parse (instead of the live model) */
static void count_block(u32 b0, u32 b1);
static Model g_blockModel;
static void parse_optimal(int segLog, int use3, int allLens)
{
    static Model m; Enc pe; pe.m = &m; pe.update = 0;
    Enc ue; ue.m = &m; ue.update = 1; ue.bits = 0;
    nSyms = 0;
    u32 segSize = 1u << segLog;
    const u32 W = 2048, SPAN = W + 280;
    Node* nd = malloc(sizeof(Node) * (SPAN + 1));
    for (u32 ss = 0; ss < N; ss += segSize) {
        u32 se = ss + segSize < N ? ss + segSize : N;
        model_reset(&m);
        if (g_optG) { Sym* keep = syms; u32 kn = nSyms; syms = g_optG; nSyms = g_optNG; count_block(ss, se); syms = keep; nSyms = kn; pe.m = &g_blockModel; }
        St cur; cur.st = 0; cur.rep[0] = cur.rep[1] = cur.rep[2] = cur.rep[3] = 1;
        u32 s0 = ss;
        while (s0 < se) {
            u32 wEnd = s0 + W < se ? s0 + W : se;              /* nodes s0 .. lim */
            u32 lim = wEnd + 273 < se ? wEnd + 273 : se;
            u32 n = lim - s0;
            for (u32 i = 0; i <= n; i++) nd[i].cost = 1e30f;
            nd[0].cost = 0; nd[0].s = cur;
            for (u32 i = 0; i < n && s0 + i < wEnd; i++) {
                if (nd[i].cost > 1e29f) continue;
                u32 p = s0 + i; St s = nd[i].s; u32 fs = p / FRAME * FRAME;
                /* literal */
                { pe.bits = 0; St t = s; e_symbol(&pe, &t, p, 0, 0, 0); float c = nd[i].cost + (float)pe.bits;
                  if (c < nd[i + 1].cost) { nd[i + 1].cost = c; nd[i + 1].prev = i; nd[i + 1].kind = 0; nd[i + 1].s = t; } }
                u32 maxl = lim - p; if (maxl > 273) maxl = 273;
                if (maxl < 2) { if (maxl == 1 && p >= fs + s.rep[0] && S[p] == S[p - s.rep[0]]) { pe.bits = 0; St t = s; e_symbol(&pe, &t, p, 2, 1, 0); float c = nd[i].cost + (float)pe.bits; if (c < nd[i + 1].cost) { nd[i + 1].cost = c; nd[i + 1].prev = i; nd[i + 1].kind = 2; nd[i + 1].s = t; } } continue; }
                /* shortrep */
                if (p >= fs + s.rep[0] && S[p] == S[p - s.rep[0]]) { pe.bits = 0; St t = s; e_symbol(&pe, &t, p, 2, 1, 0); float c = nd[i].cost + (float)pe.bits; if (c < nd[i + 1].cost) { nd[i + 1].cost = c; nd[i + 1].prev = i; nd[i + 1].kind = 2; nd[i + 1].s = t; } }
                /* reps */
                for (u32 r = 0; r < 4; r++) {
                    u32 d = s.rep[r]; if (p < fs + d) continue;
                    u32 l = mlen(p, p - d, maxl); if (l < 2) continue;
                    for (u32 x = allLens ? 2 : l; x <= l; x++) { pe.bits = 0; St t = s; e_symbol(&pe, &t, p, 3 + r, x, d); float c = nd[i].cost + (float)pe.bits;
                        if (c < nd[i + x].cost) { nd[i + x].cost = c; nd[i + x].prev = i; nd[i + x].kind = 3 + r; nd[i + x].len = x; nd[i + x].off = d; nd[i + x].s = t; } }
                }
                /* main candidates */
                for (int w = 0; w < 3; w++) {
                    u32 d = w == 2 ? rec2Off[p] : w ? rec3Off[p] : recOff[p], l = w == 2 ? rec2Len[p] : w ? rec3Len[p] : recLen[p];
                    if (w == 1 && !use3) continue; if (!l) continue;
                    l = full_len(p, d, l, maxl); if (l > maxl) l = maxl; if (l < 2) continue;
                    for (u32 x = allLens ? 2 : l; x <= l; x++) { pe.bits = 0; St t = s; e_symbol(&pe, &t, p, 1, x, d); float c = nd[i].cost + (float)pe.bits;
                        if (c < nd[i + x].cost) { nd[i + x].cost = c; nd[i + x].prev = i; nd[i + x].kind = 1; nd[i + x].len = x; nd[i + x].off = d; nd[i + x].s = t; } }
                }
            }
            /* end node: at the window end if reachable, else the cheapest (per byte) reachable behind it */
            u32 eI = wEnd - s0;
            if (wEnd < se) { float best = 1e30f; u32 bi = eI; float avg = nd[eI].cost < 1e29f ? nd[eI].cost / (float)(eI ? eI : 1) : 2.0f;
                for (u32 i = eI; i <= n; i++) if (nd[i].cost < 1e29f) { float c = nd[i].cost - avg * (float)(i - eI); if (c < best) { best = c; bi = i; } } eI = bi; }
            else eI = n;
            /* backtrack, then encode forward with updates */
            static u32 stack[4096]; u32 sp = 0; for (u32 i = eI; i != 0; i = nd[i].prev) stack[sp++] = i;
            u32 at = 0;
            while (sp) { u32 i = stack[--sp]; u32 p = s0 + at;
                if (nd[i].kind == 0) e_symbol(&ue, &cur, p, 0, 0, 0);
                else if (nd[i].kind == 2) e_symbol(&ue, &cur, p, 2, 1, 0);
                else { e_symbol(&ue, &cur, p, nd[i].kind, nd[i].len, nd[i].off); push(p, nd[i].len, nd[i].off); }
                at = i; }
            s0 += eI;
        }
    }
    fprintf(stderr, "   optimal direct size %.0f\n", ue.bits / 8.0);
    free(nd);
}


/* ---------------------------------------------------------------- the GPU-feasible DP ("W7") restated on the CPU
 * Differences from parse_optimal that the GPU design needs:
 *   - static prices: one probability set per 128 KiB block, counted from the symbols of a first greedy parse of that block
 *   - independent windows of 4 KiB (= rc chunks; matches never cross them): coder state and repeat distances unknown at the start
 *   - repeat distances are only recognised by equality with the candidate's distance (no path-dependent memory compares), plus
 *     the precomputed continuation behind every record: match(d, L) + literal + rep0(d, c)
 *   - no short rep; matched literals priced exactly or as plain literals (flag) */
typedef struct { int staticPrices, win, repCompare, composite, shortRep, litPlain, use3, maxLenCap, repDetect, d3max, simpleState; } DpCfg;
static u32 g_blockModelFor = 0xFFFFFFFFu;
typedef struct { Model* m; } CountSink;
static void count_block(u32 b0, u32 b1)          /* events of the greedy symbols inside [b0, b1) -> static probabilities */
{
    memset(cnt0, 0, sizeof cnt0); memset(cnt1, 0, sizeof cnt1);
    /* run the symbol list with a model that records counts: emulate by encoding with update into a scratch model while counting */
    static Model scratch; model_reset(&scratch);
    Enc e; e.m = &scratch; e.update = 0; e.bits = 0;
    /* we need the event indices: re-implement via a hook -- simplest: temporarily use probabilities as event recorders */
    /* (hook: p[idx] is left at 1024; counts are collected in ebit_count below) */
    g_counting = 1;
    St s; s.st = 0; s.rep[0] = s.rep[1] = s.rep[2] = s.rep[3] = 1;
    u32 k = 0; while (k < nSyms && syms[k].pos + syms[k].len <= b0) k++;
    u32 p = b0;
    while (p < b1) {
        if (k < nSyms && syms[k].pos <= p) {
            u32 sp = syms[k].pos, len = syms[k].len, off = syms[k].off;
            if (sp < p) { len -= p - sp; }
            if (len > b1 - p) len = b1 - p;
            k++;
            while (len >= 2) { u32 take = len < 273 ? len : 273; if (len - take == 1) take--; u32 kind = 1; for (u32 r = 0; r < 4; r++) if (s.rep[r] == off) { kind = 3 + r; break; }
                e_symbol(&e, &s, p, kind, take, off); p += take; len -= take; }
            if (len == 1) { e_symbol(&e, &s, p, 0, 0, 0); p++; }
        } else { e_symbol(&e, &s, p, 0, 0, 0); p++; }
    }
    g_counting = 0;
    for (int i = 0; i < NPROB; i++) { u32 a = cnt0[i], b = cnt1[i]; u32 pr = (a + b) ? (u32)(2048.0 * (a + 0.4) / (a + b + 0.8)) : 1024; if (pr < 31) pr = 31; if (pr > 2017) pr = 2017; g_blockModel.p[i] = (uint16_t)pr; }
}
typedef struct { float cost; u32 prev; u32 len; u32 off; u32 len2; u8 kind; St s; } Node2;   /* kind 9 = composite: match(off,len) lit rep0(len2) */
static void dp_relax(Node2* nd, u32 to, float c, u32 from, u8 kind, u32 len, u32 off, u32 len2, const St* t)
{ if (c < nd[to].cost) { nd[to].cost = c; nd[to].prev = from; nd[to].kind = kind; nd[to].len = len; nd[to].off = off; nd[to].len2 = len2; nd[to].s = *t; } }
static int g_dpIter = 0;     /* > 0: the symbol list already in syms[] (a previous DP's parse) supplies the statistics instead of the greedy parse */
static void parse_dp_gpu(DpCfg cfg, int greedyLazy)
{
    static Model madapt;
    if (!g_dpIter) parse_greedy(greedyLazy);
    Sym* g = malloc(sizeof(Sym) * (nSyms + 1)); u32 ng = nSyms; memcpy(g, syms, sizeof(Sym) * nSyms);
    Sym* out = malloc(sizeof(Sym) * (N / 2 + 16)); u32 nOut = 0;
    Node2* nd = malloc(sizeof(Node2) * (cfg.win + 2));
    Enc pe; pe.update = 0;
    /* repCompare >= 2: rep candidates only at distances named by HINTS = the last 4 distinct distances of the previous parse's matches that start before p
     * (repCompare 3: matches that END at or before p) -- position-only data, computable in parallel before the DP */
    u32* hint = NULL;
    static Sym* gGreedy = NULL; static u32 nGreedy = 0;
    if (!g_dpIter) { gGreedy = malloc(sizeof(Sym) * (nSyms + 1)); memcpy(gGreedy, syms, sizeof(Sym) * nSyms); nGreedy = nSyms; }
    if (cfg.repCompare >= 2) { Sym* g = gGreedy && getenv("LAB_HINT_GREEDY") ? gGreedy : syms; u32 ng = gGreedy && getenv("LAB_HINT_GREEDY") ? nGreedy : nSyms;
        hint = calloc((size_t)N * 4, 4); u32 lru[4] = {0,0,0,0}; u32 k = 0; int nh = getenv("LAB_NHINT") ? atoi(getenv("LAB_NHINT")) : 4;
        for (u32 p = 0; p < N; p++) {
            if (p % FRAME == 0) lru[0] = lru[1] = lru[2] = lru[3] = 0;
            while (k < ng && (cfg.repCompare == 3 ? g[k].pos + g[k].len <= p : g[k].pos < p)) { u32 d = g[k].off; int j = 0; for (j = 0; j < 3; j++) if (lru[j] == d) break; for (; j > 0; j--) lru[j] = lru[j - 1]; lru[0] = d; k++; }
            for (int j = 0; j < 4; j++) hint[(size_t)p * 4 + j] = j < nh ? lru[j] : 0; } }
    u32 pblk = getenv("LAB_PBLK") ? (u32)atoi(getenv("LAB_PBLK")) : (128u << 10), pctx = getenv("LAB_PCTX") ? (u32)atoi(getenv("LAB_PCTX")) : 0;
    for (u32 b0 = 0; b0 < N; b0 += pblk) {
        u32 b1 = b0 + pblk < N ? b0 + pblk : N;
        memcpy(syms, g, sizeof(Sym) * ng); nSyms = ng;
        if (cfg.staticPrices) { u32 c0 = b0 > pctx ? b0 - pctx : 0, c1 = b1 + pctx < N ? b1 + pctx : N; count_block(c0, c1); pe.m = &g_blockModel; } else { model_reset(&madapt); pe.m = &madapt; }
        for (u32 w0 = b0; w0 < b1; w0 += cfg.win) {
            u32 w1 = w0 + cfg.win < b1 ? w0 + cfg.win : b1, n = w1 - w0;
            for (u32 i = 0; i <= n; i++) nd[i].cost = 1e30f;
            nd[0].cost = 0; nd[0].s.st = 0; nd[0].s.rep[0] = nd[0].s.rep[1] = nd[0].s.rep[2] = nd[0].s.rep[3] = 0;   /* 0 = unknown */
            if (hint && getenv("LAB_INITREP")) for (int j = 0; j < 4; j++) nd[0].s.rep[j] = hint[(size_t)w0 * 4 + j];
            u32 skipTo = 0; int nice = getenv("LAB_NICE2") ? atoi(getenv("LAB_NICE2")) : 0; int hcap = getenv("LAB_HCAP") ? atoi(getenv("LAB_HCAP")) : 273;
            for (u32 i = 0; i < n; i++) {
                u32 p = w0 + i; St s = nd[i].s; u32 fs = p / FRAME * FRAME; float c0 = nd[i].cost;
                if (i < skipTo || c0 > 1e29f) continue;
                if (nice) { u32 d = recOff[p], l = recLen[p]; if (l) { u32 mx = n - i; l = full_len(p, d, l, mx < 64 ? mx : 64); if (l >= (u32)nice) {
                    u32 kind = 1; for (u32 r = 0; r < (u32)cfg.repDetect; r++) if (s.rep[r] == d) { kind = 3 + r; break; }
                    pe.bits = 0; St t = s; for (int q = 0; q < 4; q++) if (!t.rep[q]) t.rep[q] = 0xFFFFFFFFu; e_symbol(&pe, &t, p, kind, l, d); for (int q = 0; q < 4; q++) if (t.rep[q] == 0xFFFFFFFFu) t.rep[q] = 0;
                    for (u32 x = i + 1; x <= n; x++) nd[x].cost = 1e30f;
                    dp_relax(nd, i + l, c0 + (float)pe.bits, i, kind, l, d, 0, &t); skipTo = i + l; continue; } } }
                if (cfg.simpleState) s.st = s.st >= 7 ? 7 : 0;
                St sl = s; if (!sl.rep[0]) sl.rep[0] = 1;
                { pe.bits = 0; St t = sl; if (cfg.litPlain && t.st >= 7) { u32 keep = t.st; t.st = 0; e_symbol(&pe, &t, p, 0, 0, 0); t.st = st_lit(keep); pe.bits += 0; } else e_symbol(&pe, &t, p, 0, 0, 0);
                  t.rep[0] = s.rep[0]; dp_relax(nd, i + 1, c0 + (float)pe.bits, i, 0, 0, 0, 0, &t); }
                u32 maxl = n - i; if (maxl > 273) maxl = 273;
                if (cfg.shortRep && s.rep[0] && p >= fs + s.rep[0] && S[p] == S[p - s.rep[0]] && (!hint || getenv("LAB_SREP_FREE") || hint[(size_t)p*4]==s.rep[0] || hint[(size_t)p*4+1]==s.rep[0] || hint[(size_t)p*4+2]==s.rep[0] || hint[(size_t)p*4+3]==s.rep[0])) { pe.bits = 0; St t = s; e_symbol(&pe, &t, p, 2, 1, 0); dp_relax(nd, i + 1, c0 + (float)pe.bits, i, 2, 1, 0, 0, &t); }
                if (maxl < 2) continue;
                if (cfg.repCompare) for (u32 r = 0; r < 4; r++) { u32 d = s.rep[r]; if (!d || p < fs + d) continue;
                    if (hint) { int ok = 0; for (int j = 0; j < 4; j++) if (hint[(size_t)p * 4 + j] == d) ok = 1; if (!ok) continue; }
                    u32 l = mlen(p, p - d, maxl < (u32)hcap ? maxl : (u32)hcap); if (l < 2) continue;
                    for (u32 x = 2; x <= l; x++) { pe.bits = 0; St t = s; e_symbol(&pe, &t, p, 3 + r, x, d); dp_relax(nd, i + x, c0 + (float)pe.bits, i, 3 + r, x, d, 0, &t); } }
                for (int w = 0; w < 3; w++) {
                    u32 d = w == 2 ? rec2Off[p] : w ? rec3Off[p] : recOff[p], l = w == 2 ? rec2Len[p] : w ? rec3Len[p] : recLen[p];
                    if (w == 1 && !cfg.use3) continue; if (w == 2 && !getenv("LAB_TWO")) continue; if (!l) continue;
                    u32 cap = cfg.maxLenCap ? (u32)cfg.maxLenCap : 273;
                    l = full_len(p, d, l, maxl < cap ? maxl : cap); if (l < 2) continue;
                    u32 kind = 1; for (u32 r = 0; r < (u32)cfg.repDetect; r++) if (s.rep[r] == d) { kind = 3 + r; break; }
                    if (w == 1 && cfg.d3max && d > (u32)cfg.d3max) continue;
                    St tl; float cl = 0;
                    for (u32 x = 2; x <= l; x++) { pe.bits = 0; St t = s; for (int q = 0; q < 4; q++) if (!t.rep[q]) t.rep[q] = 0xFFFFFFFFu; e_symbol(&pe, &t, p, kind, x, d); for (int q = 0; q < 4; q++) if (t.rep[q] == 0xFFFFFFFFu) t.rep[q] = 0;
                        dp_relax(nd, i + x, c0 + (float)pe.bits, i, kind, x, d, 0, &t); if (x == l) { tl = t; cl = c0 + (float)pe.bits; } }
                    if (cfg.composite && w == 0 && i + l + 3 <= n && l < 273) {   /* match + literal + rep0 */
                        u32 q = p + l; u32 c = mlen(q + 1, q + 1 - d, (n - i - l - 1) < 273 ? (n - i - l - 1) : 273);
                        if (c >= 2) { pe.bits = 0; St t = tl; e_symbol(&pe, &t, q, 0, 0, 0); float c1 = cl + (float)pe.bits;
                            for (u32 x = 2; x <= c; x++) { pe.bits = 0; St t2 = t; e_symbol(&pe, &t2, q + 1, 3, x, d); dp_relax(nd, i + l + 1 + x, c1 + (float)pe.bits, i, 9, l, d, x, &t2); } }
                    }
                }
            }
            static u32 stack[140000]; u32 sp = 0; for (u32 i = n; i != 0; i = nd[i].prev) stack[sp++] = i;
            u32 at = 0;
            while (sp) { u32 i = stack[--sp]; u32 p = w0 + at;
                if (nd[i].kind == 9) { out[nOut].pos = p; out[nOut].len = nd[i].len; out[nOut].off = nd[i].off; nOut++; out[nOut].pos = p + nd[i].len + 1; out[nOut].len = nd[i].len2; out[nOut].off = nd[i].off; nOut++; }
                else if (nd[i].kind == 1 || nd[i].kind >= 3) { out[nOut].pos = p; out[nOut].len = nd[i].len; out[nOut].off = nd[i].off; nOut++; }
                at = i; }
        }
    }
    memcpy(syms, out, sizeof(Sym) * nOut); nSyms = nOut; free(out); free(g); free(nd);
}


/* ---------------------------------------------------------------- W7 as it would run on the GPU: entropy-style static prices
 * from simple per-block statistics (no LZMA event counting), no state, candidates main (2..L<=64) and short. */
static void parse_dp_simple(int litFromAll, int d3max, int win, int lsClasses)
{
    parse_greedy(2);
    Sym* g = malloc(sizeof(Sym) * (nSyms + 1)); u32 ng = nSyms; memcpy(g, syms, sizeof(Sym) * nSyms);
    Sym* out = malloc(sizeof(Sym) * (N / 2 + 16)); u32 nOut = 0;
    float* cost = malloc(4 * (win + 2)); u32* bk = malloc(4 * (win + 2)); u8* bkind = malloc(win + 2);
    static float litP[8][256], lenP[66], slotP[4][64];
    u32 k = 0;
    for (u32 b0 = 0; b0 < N; b0 += (128u << 10)) {
        u32 b1 = b0 + (128u << 10) < N ? b0 + (128u << 10) : N;
        /* statistics of the block */
        static u32 lc[8][256], ls[8], lenH[66], slotH[4][64], slotN[4]; u32 nLit = 0, nMat = 0;
        memset(lc, 0, sizeof lc); memset(ls, 0, sizeof ls); memset(lenH, 0, sizeof lenH); memset(slotH, 0, sizeof slotH); memset(slotN, 0, sizeof slotN);
        u32 p = b0; u32 kk = k;
        while (p < b1) {
            if (kk < ng && g[kk].pos <= p) { u32 sp = g[kk].pos, len = g[kk].len, off = g[kk].off; if (sp < p) len -= p - sp; if (len > b1 - p) len = b1 - p; kk++;
                u32 q = p; p += len;
                while (len >= 2) { u32 t = len < 64 ? len : 64; if (len - t == 1) t--; lenH[t]++; nMat++; u32 dd = off - 1, slot; if (dd < 4) slot = dd; else { u32 hb = hibit(dd); slot = 2 * hb + ((dd >> (hb - 1)) & 1); }
                    u32 c = lsClasses == 1 ? 0 : (t - 2 < 3 ? t - 2 : 3); slotH[c][slot]++; slotN[c]++; len -= t; }
                if (litFromAll) for (; q < p; q++) { u32 ctx = q ? S[q - 1] >> 5 : 0; lc[ctx][S[q]]++; ls[ctx]++; }
            } else { u32 ctx = p ? S[p - 1] >> 5 : 0; lc[ctx][S[p]]++; ls[ctx]++; nLit++; p++; }
        }
        while (k < ng && g[k].pos + g[k].len <= b1) k++;
        for (int c = 0; c < 8; c++) for (int b = 0; b < 256; b++) litP[c][b] = (float)-log2((lc[c][b] + 0.3) / (ls[c] + 0.3 * 256));
        float fl = (float)-log2((nLit + 1.0) / (nLit + nMat + 2.0)), fm = (float)-log2((nMat + 1.0) / (nLit + nMat + 2.0));
        for (int l = 2; l <= 65; l++) lenP[l] = (float)-log2((lenH[l < 64 ? l : 64] + 0.5) / (nMat + 0.5 * 63));
        for (int c = 0; c < 4; c++) for (int sl = 0; sl < 64; sl++) slotP[c][sl] = (float)-log2((slotH[c][sl] + 0.5) / (slotN[c] + 0.5 * 44));
        for (u32 w0 = b0; w0 < b1; w0 += win) {
            u32 w1 = w0 + win < b1 ? w0 + win : b1, n = w1 - w0;
            for (u32 i = 0; i <= n; i++) cost[i] = 1e30f; cost[0] = 0;
            u32 skipTo = 0, contAt = 0xFFFFFFFFu, contOff = 0;
            for (u32 i = 0; i < n; i++) {
                u32 pp = w0 + i; float c0 = cost[i];
                if (i < skipTo) continue;
                if (getenv("LAB_NICE") && recLen[pp] == 64 && n - i >= 64) { u32 d = recOff[pp]; u32 M = full_len(pp, d, 64, (n - i) < 273 ? (n - i) : 273); if (atoi(getenv("LAB_NICE")) >= 2) M = 64;
                    if (atoi(getenv("LAB_NICE")) == 3) { float c = c0 + ((contAt == i && contOff == d) ? 0.25f : fm + lenP[64] + slotP[3][(d - 1) < 4 ? (d - 1) : 2 * hibit(d - 1) + (((d - 1) >> (hibit(d - 1) - 1)) & 1)] + ((d - 1) >= 4 ? (float)(hibit(d - 1) - 1) : 0));
                        if (c < cost[i + M]) { cost[i + M] = c; bk[i + M] = M; bkind[i + M] = 1; } skipTo = i + M; contAt = i + M; contOff = d; continue; }
                    u32 dd = d - 1, slot; if (dd < 4) slot = dd; else { u32 hb = hibit(dd); slot = 2 * hb + ((dd >> (hb - 1)) & 1); }
                    float c = c0 + fm + lenP[64] + slotP[3][slot] + (slot >= 4 ? (float)((slot >> 1) - 1) : 0);
                    if (c < cost[i + M]) { cost[i + M] = c; bk[i + M] = M; bkind[i + M] = 1; } skipTo = i + M; continue; }
                { float c = c0 + fl + litP[pp ? S[pp - 1] >> 5 : 0][S[pp]]; if (c < cost[i + 1]) { cost[i + 1] = c; bk[i + 1] = 1; bkind[i + 1] = 0; } }
                for (int w = 0; w < 2; w++) {
                    u32 d = w ? rec3Off[pp] : recOff[pp], l = w ? rec3Len[pp] : recLen[pp];
                    if (!l || (w && d3max && d > (u32)d3max)) continue;
                    if (l > n - i) l = n - i; if (l > 64) l = 64;
                    u32 dd = d - 1, slot; if (dd < 4) slot = dd; else { u32 hb = hibit(dd); slot = 2 * hb + ((dd >> (hb - 1)) & 1); }
                    float foot = slot >= 4 ? (float)((slot >> 1) - 1) : 0;
                    for (u32 x = 2; x <= l; x++) { u32 c4 = lsClasses == 1 ? 0 : (x - 2 < 3 ? x - 2 : 3); float c = c0 + fm + lenP[x] + slotP[c4][slot] + foot; if (!w && contAt == i && contOff == d) c = c0 + 0.25f; if (c < cost[i + x]) { cost[i + x] = c; bk[i + x] = x; bkind[i + x] = 1 + w; } }
                }
            }
            static u32 stack[8192]; u32 sp = 0; for (u32 i = n; i != 0; i -= bk[i]) stack[sp++] = i;
            u32 at = 0;
            while (sp) { u32 i = stack[--sp]; u32 pp = w0 + at; if (bkind[i]) { out[nOut].pos = pp; out[nOut].len = bk[i]; out[nOut].off = bkind[i] == 1 ? recOff[pp] : rec3Off[pp]; nOut++; } at = i; }
        }
    }
    memcpy(syms, out, sizeof(Sym) * nOut); nSyms = nOut; free(out); free(g); free(cost); free(bk); free(bkind);
}

int main(int argc, char** argv)
{
    for (int i = 1; i <= 2048; i++) g_cost[i] = (float)(-log2((double)i / 2048.0)); g_cost[0] = 20;
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); N = ftell(f); fseek(f, 0, SEEK_SET);
    S = malloc(N + 64); if (fread(S, 1, N, f) != N) return 1; fclose(f);
    int segLog = argc > 2 ? atoi(argv[2]) : 15;
    int depth = argc > 3 ? atoi(argv[3]) : 2;
    syms = malloc(sizeof(Sym) * (N / 2 + 16));
    double h;
    finder(depth, getenv("LAB_MINMATCH") ? atoi(getenv("LAB_MINMATCH")) : 5, 1);
    if (getenv("LAB_HC")) finder_hc(atoi(getenv("LAB_HC")));
    if (getenv("LAB_LPM")) finder_lpm(atoi(getenv("LAB_LPM")));
    if (getenv("LAB_OPT")) {
        if (getenv("LAB_OPT_STATIC")) { parse_greedy(2); g_optG = malloc(sizeof(Sym) * (nSyms + 1)); memcpy(g_optG, syms, sizeof(Sym) * nSyms); g_optNG = nSyms; }
        parse_optimal(atoi(getenv("LAB_OPT")), 1, 1);
        for (int it = 0; it < (getenv("LAB_OPT_STATIC") ? atoi(getenv("LAB_OPT_STATIC")) - 1 : 0); it++) { free(g_optG); g_optG = malloc(sizeof(Sym) * (nSyms + 1)); memcpy(g_optG, syms, sizeof(Sym) * nSyms); g_optNG = nSyms; parse_optimal(atoi(getenv("LAB_OPT")), 1, 1); }
        return 0; }
    if (getenv("LAB_DP")) {
        parse_greedy(2); double a0 = price_syms(segLog, 12, &h); printf("greedy-lazy2 seg %d: %.0f\n", segLog, a0);
        int v[11] = {1, 4096, 0, 1, 0, 0, 0, 0, 4, 0, 0}; const char* e = getenv("LAB_DP"); sscanf(e, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%d", v, v+1, v+2, v+3, v+4, v+5, v+6, v+7, v+8, v+9, v+10);
        DpCfg c = { v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10] };
        parse_dp_gpu(c, 2); double a = price_syms(segLog, 12, &h); printf("dp [%s] seg %d: %.0f  (%.4f of greedy)\n", e, segLog, a, a / a0);
        for (int it = 0; it < (getenv("LAB_ITER") ? atoi(getenv("LAB_ITER")) : 0); it++) { g_dpIter = 1; parse_dp_gpu(c, 2); a = price_syms(segLog, 12, &h); printf("  iteration %d: %.0f\n", it + 2, a); }
        return 0; }
    if (getenv("LAB_SIMPLE")) {
        parse_greedy(2); double a0 = price_syms(segLog, 12, &h); printf("greedy-lazy2 seg %d: %.0f\n", segLog, a0);
        int v[4] = {0, 4096, 4096, 4}; const char* e = getenv("LAB_SIMPLE"); sscanf(e, "%d,%d,%d,%d", v, v+1, v+2, v+3);
        parse_dp_simple(v[0], v[1], v[2], v[3]); double a = price_syms(segLog, 12, &h); printf("simple [%s] seg %d: %.0f  (%.4f of greedy)\n", e, segLog, a, a / a0); return 0; }
    if (getenv("LAB_QUICK")) { parse_greedy(2); double a = price_syms(segLog, 12, &h); printf("greedy-lazy2 seg %d: %.0f (+hdr %.0f)\n", segLog, a, h);
        parse_optimal(segLog, 1, 1); parse_optimal(23, 1, 1); return 0; }
    parse_greedy(2); double a = price_syms(segLog, 12, &h); printf("greedy-lazy2 seg %d: %.0f (+hdr %.0f)\n", segLog, a, h);
    parse_greedy(2); a = price_syms(17, 12, &h); printf("greedy-lazy2 seg 17: %.0f\n", a);
    parse_greedy(2); a = price_syms(23, 12, &h); printf("greedy-lazy2 seg 23: %.0f\n", a);
    /* refinement */
    { u32* so = malloc(4u * N); u8* sl = malloc(N); memcpy(so, recOff, 4u * N); memcpy(sl, recLen, N);
      for (int it = 0; it < 2; it++) { parse_greedy(2); memcpy(recOff, so, 4u * N); memcpy(recLen, sl, N); refine_records(); }
      parse_greedy(2); a = price_syms(segLog, 12, &h); printf("refined x2 seg %d: %.0f\n", segLog, a);
      parse_greedy(2); a = price_syms(23, 12, &h); printf("refined x2 seg 23: %.0f\n", a);
      memcpy(recOff, so, 4u * N); memcpy(recLen, sl, N); }
    parse_optimal(segLog, 0, 1); parse_optimal(segLog, 1, 1); parse_optimal(segLog, 1, 0);
    parse_optimal(17, 1, 1); parse_optimal(23, 1, 1);
    return 0;
}
