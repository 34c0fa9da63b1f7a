/* oracle/lzfind_hc4.c -- TEST INFRASTRUCTURE ONLY (the checker of SURVEY.md 8(f3) / a20; never linked into the product).
 *
 * Plain-C restatement of what the reference's mainline hash-chain match finder HC4 returns for every position of a buffer
 * (C/LzFind.c: Hc4_MatchFinder_GetMatches :1362-1425, Hc_GetMatchesSpec :880-946, HASH4_CALC :49-54, SET_mmm :1171-1174,
 * lenLimit of MatchFinder_SetLimits, hash mask of MatchFinder_GetHashMask :345-372, cyclicBufferSize = historySize + 1 :449),
 * written in the data-parallel form a device kernel would take -- three passes, of which only the second is a scan:
 *
 *   1. per position i (independent): the three hash values h2 (10 bits of crc[b0] ^ b1), h3 (16 bits, + b2 << 8), hv (& hashMask, + crc[b3] << 5);
 *      a position whose remaining bytes are fewer than 4 takes no part (the reference only moves on there: MatchFinder_MovePos);
 *   2. per hash table: prev[i] = the latest earlier position with the same hash value (what the table holds when position i looks it up);
 *      on a device: sort (hash, position) pairs, neighbours in the sorted order;
 *   3. per position (independent again, given the prev arrays): the 2- and 3-byte candidates from prev2 / prev3, then the walk along
 *      prevV (the "son" links) for at most cutValue steps, inside the window, reporting strictly growing lengths.
 *
 * Output per position: the UInt32 values GetMatches writes (length, distance - 1, ...), and their number.
 * Pinned against the reference itself (oracle/_ref/liblzfind_ref.so, ref_shim_lzfind.c) by tests/test_oracle.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>

static uint32_t hc4_hash_mask(uint32_t historySize)          /* MatchFinder_GetHashMask for numHashBytes = 4 */
{
    uint32_t hs = historySize;
    if (hs) hs--;
    hs |= hs >> 1; hs |= hs >> 2; hs |= hs >> 4; hs |= hs >> 8;
    hs >>= 1;
    if (hs >= (1u << 24)) hs >>= 1;
    return hs | 0xFFFFu;
}

int gc_oracle_hc4_matches(const uint8_t* data, size_t n, uint32_t historySize, uint32_t cut, uint32_t niceLen,
                          uint32_t* counts, uint32_t* pairs, size_t pairCap, size_t* pairsUsed)
{
    uint32_t crc[256];
    const uint32_t mask = hc4_hash_mask(historySize), window = historySize + 1u;      /* cyclicBufferSize */
    uint32_t *h2 = NULL, *h3 = NULL, *hv = NULL, *prev2 = NULL, *prev3 = NULL, *prevV = NULL, *head = NULL;
    size_t i, used = 0;
    int rc = 0;
    if (n >= 0xFFFF0000u) return -3;
    for (i = 0; i < 256; i++) { uint32_t r = (uint32_t)i; int k; for (k = 0; k < 8; k++) r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1u))); crc[i] = r; }
    h2 = (uint32_t*)malloc((n + 1) * 4); h3 = (uint32_t*)malloc((n + 1) * 4); hv = (uint32_t*)malloc((n + 1) * 4);
    prev2 = (uint32_t*)malloc((n + 1) * 4); prev3 = (uint32_t*)malloc((n + 1) * 4); prevV = (uint32_t*)malloc((n + 1) * 4);
    head = (uint32_t*)calloc((size_t)mask + 1u > 65536u ? (size_t)mask + 1u : 65536u, 4);
    if (!h2 || !h3 || !hv || !prev2 || !prev3 || !prevV || !head) { rc = -1; goto done; }
    /* pass 1 */
    for (i = 0; i + 4 <= n; i++) {
        uint32_t t = crc[data[i]] ^ data[i + 1];
        h2[i] = t & 1023u;
        t ^= (uint32_t)data[i + 2] << 8;
        h3[i] = t & 65535u;
        hv[i] = (t ^ (crc[data[i + 3]] << 5)) & mask;
    }
    /* pass 2: positions are numbered from 1 (0 = "none"), as the reference's pos */
    { const uint32_t* hh[3] = { h2, h3, hv }; uint32_t* pp[3] = { prev2, prev3, prevV }; const size_t sz[3] = { 1024, 65536, (size_t)mask + 1u }; int t;
      for (t = 0; t < 3; t++) { memset(head, 0, sz[t] * 4); for (i = 0; i + 4 <= n; i++) { pp[t][i] = head[hh[t][i]]; head[hh[t][i]] = (uint32_t)i + 1u; } } }
    /* pass 3 */
    for (i = 0; i < n; i++) {
        const uint8_t* cur = data + i;
        const uint32_t pos = (uint32_t)i + 1u;
        const uint32_t lenLimit = n - i < niceLen ? (uint32_t)(n - i) : niceLen;
        uint32_t out[2 * 273 + 8], no = 0, maxLen = 3, mmm, d2, d3, ext = 0;
        counts[i] = 0;
        if (lenLimit < 4) continue;
        mmm = pos < window ? pos : window;
        d2 = pos - prev2[i]; d3 = pos - prev3[i];
        {   /* the 2- and 3-byte tables: at most two pairs; `ext` = the distance whose match is then measured beyond 3 bytes */
            const int c2 = d2 < mmm && cur[-(ptrdiff_t)d2] == cur[0], c3 = d3 < mmm && cur[-(ptrdiff_t)d3] == cur[0];
            if (c2) {
                out[no++] = 2; out[no++] = d2 - 1u;
                if (cur[2 - (ptrdiff_t)d2] == cur[2]) ext = d2;
                else if (c3) { out[no++] = 0; out[no++] = d3 - 1u; ext = d3; }
            } else if (c3) { out[no++] = 0; out[no++] = d3 - 1u; ext = d3; }
        }
        if (ext) {
            uint32_t l = 3;
            while (l < lenLimit && cur[l - (ptrdiff_t)ext] == cur[l]) l++;
            maxLen = l; out[no - 2] = l;
        }
        if (!(ext && maxLen == lenLimit)) {
            /* the chain: prevV links, newest first */
            uint32_t m = prevV[i], steps = cut;
            while (steps && m) {
                const uint32_t delta = pos - m;
                uint32_t l = 0;
                if (delta >= window) break;
                if (cur[maxLen] == cur[(ptrdiff_t)maxLen - (ptrdiff_t)delta]) {
                    while (l < lenLimit && cur[l] == cur[l - (ptrdiff_t)delta]) l++;
                    if (l == lenLimit) { out[no++] = l; out[no++] = delta - 1u; break; }
                    if (l > maxLen) { maxLen = l; out[no++] = l; out[no++] = delta - 1u; }
                }
                m = prevV[m - 1u];
                steps--;
            }
        }
        if (used + no > pairCap) { rc = -2; goto done; }
        memcpy(pairs + used, out, no * 4u); used += no; counts[i] = no;
    }
    *pairsUsed = used;
done:
    free(h2); free(h3); free(hv); free(prev2); free(prev3); free(prevV); free(head);
    return rc;
}
