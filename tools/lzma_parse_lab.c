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
static inline void ebit(Enc* e, u32 idx, u32 bit)
{
    u32 p = e->m->p[idx];
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
static inline u32 mlen(u32 a, u32 b, u32 max) { u32 l = 0; while (l < max && S[a + l] == S[b + l]) l++; return l; }
static inline int gain(u32 len, u32 off) { return (int)(len * 4) - (int)hibit(off + 1); }
static void finder(int depth, int minMatch, int use3)
{
    recOff = calloc(N, 4); recLen = calloc(N, 1); rec3Off = calloc(N, 4); rec3Len = calloc(N, 1);
    u32* tL = malloc(4u << 20), *tS = malloc(4u << 19), *t3 = malloc(4u << 16);
    for (u32 f = 0; f < N; f += FRAME) {
        u32 fe = f + FRAME < N ? f + FRAME : N;
        memset(tL, 0xFF, 4u << 20); memset(tS, 0xFF, 4u << 19); memset(t3, 0xFF, 4u << 16);
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
            if (cL != 0xFFFFFFFFu) { u32 l = mlen(p, cL, CAP); if (l >= (u32)minMatch) { bl = l; bo = p - cL; } }
            if (bl < 8 && cS != 0xFFFFFFFFu && cS != cL) { u32 l = mlen(p, cS, CAP); if (l >= (u32)minMatch && (bl == 0 || gain(l, p - cS) > gain(bl, bo))) { bl = l; bo = p - cS; } }
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
                for (int w = 0; w < 2; w++) {
                    u32 d = w ? rec3Off[p] : recOff[p], l = w ? rec3Len[p] : recLen[p];
                    if (w && !use3) break; if (!l) continue;
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

int main(int argc, char** argv)
{
    for (int i = 1; i <= 2048; i++) g_cost[i] = (float)(-log2((double)i / 2048.0)); g_cost[0] = 20;
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); N = ftell(f); fseek(f, 0, SEEK_SET);
    S = malloc(N + 64); if (fread(S, 1, N, f) != N) return 1; fclose(f);
    int segLog = argc > 2 ? atoi(argv[2]) : 15;
    int depth = argc > 3 ? atoi(argv[3]) : 2;
    syms = malloc(sizeof(Sym) * (N / 2 + 16));
    double h;
    finder(depth, 5, 1);
    if (getenv("LAB_HC")) finder_hc(atoi(getenv("LAB_HC")));
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
