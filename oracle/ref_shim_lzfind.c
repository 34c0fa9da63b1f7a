/* oracle/ref_shim_lzfind.c -- TEST INFRASTRUCTURE ONLY.  Thin driver around the reference's mainline match finders (C/LzFind.c, compiled from
 * /root/reference into oracle/_ref/liblzfind_ref.so): runs MatchFinder_Create / Init / GetMatches over a buffer the way LzmaEnc's
 * ReadMatchDistances does (C/LzmaEnc.c: p->matchFinder.GetMatches(p->matchFinderObj, p->matches)) and records, for every position, the
 * (length, distance - 1) pairs it returns.  This is the pinned oracle of SURVEY.md 8(f3) / a20; nothing in the product links it. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "LzFind.h"

static void* shim_alloc(ISzAllocPtr p, size_t size) { (void)p; return size ? malloc(size) : NULL; }
static void shim_free(ISzAllocPtr p, void* a) { (void)p; free(a); }
static const ISzAlloc g_shimAlloc = { shim_alloc, shim_free };

/* bt: 0 = hash chain (HC4 / HC5), 1 = binary tree (BT2..BT5); numHashBytes 2..5; cut = cutValue; niceLen = matchMaxLen (fb in LzmaEnc terms).
 * counts[i] = number of UInt32 values GetMatches wrote at position i (2 per pair), pairs = all of them in order.  Returns 0, -1 no memory,
 * -2 pairs capacity too small. */
int ref_lzfind_matches(const uint8_t* data, size_t n, uint32_t historySize, int bt, int numHashBytes, uint32_t cut, uint32_t niceLen,
                       uint32_t* counts, uint32_t* pairs, size_t pairCap, size_t* pairsUsed)
{
    CMatchFinder mf;
    IMatchFinder2 vt;
    UInt32 tmp[2 * 273 + 16];
    size_t used = 0, i;
    MatchFinder_Construct(&mf);
    mf.btMode = (Byte)(bt ? 1 : 0);
    mf.numHashBytes = (UInt32)numHashBytes;
    mf.cutValue = cut;
    MatchFinder_SET_DIRECT_INPUT_BUF(&mf, data, n)
    if (!MatchFinder_Create(&mf, historySize, 0, niceLen, 273 + 1, &g_shimAlloc)) return -1;      /* keepAddBufferAfter as LzmaEnc: LZMA_MATCH_LEN_MAX + 1 */
    MatchFinder_CreateVTable(&mf, &vt);
    vt.Init(&mf);
    for (i = 0; i < n; i++) {
        UInt32 avail = vt.GetNumAvailableBytes(&mf);
        UInt32 cnt;
        if (!avail) break;
        cnt = (UInt32)(vt.GetMatches(&mf, tmp) - tmp);
        counts[i] = cnt;
        if (used + cnt > pairCap) { MatchFinder_Free(&mf, &g_shimAlloc); return -2; }
        memcpy(pairs + used, tmp, cnt * sizeof(UInt32));
        used += cnt;
    }
    for (; i < n; i++) counts[i] = 0;
    *pairsUsed = used;
    MatchFinder_Free(&mf, &g_shimAlloc);
    return 0;
}
