// gc_lzfind.hip -- the mainline LZMA match finders HC4 and BT4 on the device: for every position of a buffer exactly
// the (length, distance - 1) values the reference's GetMatches writes there.
//
// Replaces Hc4_MatchFinder_GetMatches / Hc_GetMatchesSpec (C/LzFind.c:1362-1425, :880-946), Bt4_MatchFinder_GetMatches / GetMatchesSpec1 /
// SkipMatchesSpec (:1219-1282, :962-1029, :1032-1085), the hashes of HASH4_CALC (:49-54), SET_mmm (:1171-1174), the length limit of
// MatchFinder_SetLimits (:500-536) and the hash mask of MatchFinder_GetHashMask (:345-372); what the multithreaded front end splits into a
// hash thread and a tree thread (C/LzFindMt.c:448 HashThreadFunc, :761 BtThreadFunc) is here the split into kernels.  Interface mirrored:
// IMatchFinder2::GetMatches (C/LzFind.h:127-140), one call for all positions.
//
// What is sequential in the reference and how it is taken apart (every step below is exact, not an approximation):
//   - the three hash tables (2-byte, 3-byte, main) hold "the latest earlier position with the same hash value".  That is a property of the
//     multiset of (hash value, position) pairs: sort the positions by hash value (stable LSD radix sort, 8 bits per pass) and every
//     position's predecessor in the sorted order is its table lookup -- no table, no order of insertion.
//   - HC4: the chain `son` is that same predecessor relation of the main hash, so the walk of a position reads links that no other
//     position's walk writes: one thread per position.
//   - BT4: a position's walk REBUILDS the binary tree of its hash bucket (the new position becomes the root, the nodes it passes are
//     re-hung left and right), so the positions of one bucket must go in order -- but buckets never touch each other's nodes, and a buffer
//     has hundreds of thousands of them: one LANE per bucket, each running the reference's control flow, restated, over its bucket's
//     positions in order.  The tree's nodes live in an array indexed by absolute position; the reference's cyclic buffer re-uses a slot
//     `cyclicBufferSize` positions later, but never follows a link that old (`delta >= _cyclicBufferSize` ends the walk before the slot
//     is read), so absolute slots hold the same values whenever they are read.
// Output: counts[i] values at pairs[i * stride ...] (length, distance - 1, ...), as the reference's distances array after GetMatches.
#include "gc_common.h"
#include "gc_device.h"
#include "gc_lzfind.h"

#define LZF_T 256u
#define LZF_PER 16u                              // items per thread and tile of the sort
#define LZF_TILE (LZF_T * LZF_PER)

__device__ __forceinline__ uint32_t lzf_crc_entry(uint32_t i)       // the CRC-32 table entry HASH4_CALC reads (g_CrcTable, C/7zCrc.c: polynomial 0xEDB88320)
{
    uint32_t r = i;
#pragma unroll
    for (int k = 0; k < 8; k++) r = (r >> 1) ^ (0xEDB88320u & (0u - (r & 1u)));
    return r;
}

// ------------------------------------------------------------------------------------------------ items = (hash value << 32 | position)
// which: 0 = 2-byte table (10 bits), 1 = 3-byte table (16 bits), 2 = main table (hashMask).  Positions i with i + 4 <= n take part.
extern "C" __global__ void __launch_bounds__(LZF_T)
gc_lzf_items_kernel(const uint8_t* __restrict__ src, uint32_t nPart, uint32_t which, uint32_t hashMask, uint64_t* __restrict__ items)
{
    __shared__ uint32_t sCrc[256];
    sCrc[threadIdx.x] = lzf_crc_entry(threadIdx.x);
    __syncthreads();
    const uint32_t i = blockIdx.x * LZF_T + threadIdx.x;
    if (i >= nPart) return;
    uint32_t t = sCrc[src[i]] ^ src[i + 1u];
    uint32_t key = t & 1023u;
    if (which >= 1u) { t ^= (uint32_t)src[i + 2u] << 8; key = t & 65535u; }
    if (which >= 2u) key = (t ^ (sCrc[src[i + 3u]] << 5)) & hashMask;
    items[i] = ((uint64_t)key << 32) | i;
}

// ------------------------------------------------------------------------------------------------ stable LSD radix sort, 8 bits per pass
// pass = count -> offsets -> scatter.  cnt layout: [digit][tile] (tile fastest), so that the exclusive scan over the whole array is the
// scatter offset of (digit, tile) directly.
extern "C" __global__ void __launch_bounds__(LZF_T)
gc_lzf_count_kernel(const uint64_t* __restrict__ items, uint32_t n, uint32_t shift, uint32_t nTiles, uint32_t* __restrict__ cnt)
{
    __shared__ uint32_t sH[256];
    const uint32_t t = threadIdx.x, tile = blockIdx.x;
    sH[t] = 0;
    __syncthreads();
    const uint32_t base = tile * LZF_TILE;
    for (uint32_t k = 0; k < LZF_PER; k++) {
        const uint32_t i = base + k * LZF_T + t;
        if (i < n) atomicAdd(&sH[(uint32_t)(items[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    cnt[t * nTiles + tile] = sH[t];
}

// exclusive scan of a uint32 array in three steps: sums of chunks of LZF_TILE, scan of the sums by one workgroup, chunks again
extern "C" __global__ void __launch_bounds__(LZF_T)
gc_lzf_scan_sums_kernel(const uint32_t* __restrict__ a, uint32_t n, uint32_t* __restrict__ sums)
{
    __shared__ uint32_t sW[LZF_T / 64u];
    const uint32_t t = threadIdx.x, base = blockIdx.x * LZF_TILE;
    uint32_t s = 0;
    for (uint32_t k = 0; k < LZF_PER; k++) { const uint32_t i = base + t * LZF_PER + k; if (i < n) s += a[i]; }
    s = gc_wave_sum(s);
    if ((t & 63u) == 0u) sW[t >> 6] = s;
    __syncthreads();
    if (t == 0) { uint32_t all = 0; for (uint32_t w = 0; w < LZF_T / 64u; w++) all += sW[w]; sums[blockIdx.x] = all; }
}
extern "C" __global__ void __launch_bounds__(1024)
gc_lzf_scan_top_kernel(uint32_t* __restrict__ sums, uint32_t m)       // one workgroup: exclusive scan of sums[0..m) in place
{
    __shared__ uint32_t sW[16];
    __shared__ uint32_t sCarry;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) sCarry = 0;
    __syncthreads();
    for (uint32_t b = 0; b < m; b += 1024u) {
        const uint32_t i = b + t;
        const uint32_t v = i < m ? sums[i] : 0u;
        const uint32_t incl = gc_wave_incl_sum(v);
        if (lane == 63u) sW[wave] = incl;
        __syncthreads();
        uint32_t before = sCarry, all = 0;
        for (uint32_t w = 0; w < 16u; w++) { const uint32_t c = sW[w]; if (w < wave) before += c; all += c; }
        if (i < m) sums[i] = before + incl - v;
        __syncthreads();
        if (t == 0) sCarry += all;
        __syncthreads();
    }
}
extern "C" __global__ void __launch_bounds__(LZF_T)
gc_lzf_scan_apply_kernel(uint32_t* __restrict__ a, uint32_t n, const uint32_t* __restrict__ sums)
{
    __shared__ uint32_t sW[LZF_T / 64u];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, base = blockIdx.x * LZF_TILE;
    uint32_t v[LZF_PER], s = 0;
    for (uint32_t k = 0; k < LZF_PER; k++) { const uint32_t i = base + t * LZF_PER + k; v[k] = i < n ? a[i] : 0u; s += v[k]; }
    const uint32_t incl = gc_wave_incl_sum(s);
    if (lane == 63u) sW[wave] = incl;
    __syncthreads();
    uint32_t run = sums[blockIdx.x] + incl - s;
    for (uint32_t w = 0; w < LZF_T / 64u; w++) if (w < wave) run += sW[w];
    for (uint32_t k = 0; k < LZF_PER; k++) { const uint32_t i = base + t * LZF_PER + k; if (i < n) a[i] = run; run += v[k]; }
}

// Stable scatter of one tile: wave w owns the w-th quarter of the tile and takes it 64 items at a time in order; the rank of an item among the
// items of its digit is what a returning LDS add hands back (the LDS unit serves the lanes that hit one counter in lane order, lanes are
// items in order, rounds follow each other in program order -- the property the windowed finder's W3 and W4 rely on as well).
extern "C" __global__ void __launch_bounds__(LZF_T)
gc_lzf_scatter_kernel(const uint64_t* __restrict__ in, uint32_t n, uint32_t shift, uint32_t nTiles, const uint32_t* __restrict__ offs,
                      uint64_t* __restrict__ out)
{
    constexpr uint32_t NW = LZF_T / 64u, PERW = LZF_TILE / NW;
    __shared__ uint32_t sRun[NW][256];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6, tile = blockIdx.x;
    for (uint32_t w = 0; w < NW; w++) sRun[w][t] = 0;
    __syncthreads();
    const uint32_t base = tile * LZF_TILE + wave * PERW;
    for (uint32_t r = 0; r < PERW / 64u; r++) {                   // per-wave histograms
        const uint32_t i = base + r * 64u + lane;
        if (i < n) atomicAdd(&sRun[wave][(uint32_t)(in[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {   // digit t: where each wave's share of it starts
        uint32_t run = offs[t * nTiles + tile];
        for (uint32_t w = 0; w < NW; w++) { const uint32_t c = sRun[w][t]; sRun[w][t] = run; run += c; }
    }
    __syncthreads();
    for (uint32_t r = 0; r < PERW / 64u; r++) {
        const uint32_t i = base + r * 64u + lane;
        if (i < n) { const uint64_t it = in[i]; out[atomicAdd(&sRun[wave][(uint32_t)(it >> shift) & 255u], 1u)] = it; }
        gc_wave_step();
    }
}

// ------------------------------------------------------------------------------------------------ links from the sorted order
// prev[position] = position + 1 of the latest earlier position with the same hash value, 0 if none: the value the reference reads from its
// table before it overwrites the slot (positions count from 1 there, 0 = kEmptyHashValue).  heads != nullptr: the sorted indices at which
// a hash bucket starts are collected as well (in any order: the buckets are independent).
extern "C" __global__ void __launch_bounds__(LZF_T)
gc_lzf_link_kernel(const uint64_t* __restrict__ sorted, uint32_t n, uint32_t* __restrict__ prev, uint32_t* __restrict__ heads, uint32_t* __restrict__ nHeads)
{
    const uint32_t s = blockIdx.x * LZF_T + threadIdx.x;
    bool head = false;
    if (s < n) {
        const uint64_t it = sorted[s];
        const uint64_t before = s ? sorted[s - 1u] : ~0ull;
        head = s == 0u || (uint32_t)(before >> 32) != (uint32_t)(it >> 32);
        prev[(uint32_t)it] = head ? 0u : (uint32_t)before + 1u;
    }
    if (heads) {
        const uint64_t bal = __ballot(head);
        uint32_t base = 0;
        if ((threadIdx.x & 63u) == 0u && bal) base = atomicAdd(nHeads, (uint32_t)__popcll(bal));
        base = __shfl(base, 0);
        if (head) heads[base + (uint32_t)__popcll(bal & gc_lanemask_lt())] = s;
    }
}

// ------------------------------------------------------------------------------------------------ what both finders do in front of the walk
struct LzfPos {
    uint32_t lenLimit, pos, mmm, maxLen, no;
    bool full;                  // the 2- / 3-byte match reached lenLimit
};
// the 2- and 3-byte tables (Hc4 / Bt4_MatchFinder_GetMatches up to the walk): at most two pairs, the second one measured beyond 3 bytes
__device__ __forceinline__ LzfPos lzf_front(const uint8_t* __restrict__ src, uint32_t n, uint32_t i, uint32_t window, uint32_t niceLen,
                                            uint32_t p2, uint32_t p3, uint32_t* __restrict__ out)
{
    LzfPos P;
    const uint8_t* cur = src + i;
    P.lenLimit = n - i < niceLen ? n - i : niceLen;
    P.pos = i + 1u; P.mmm = P.pos < window ? P.pos : window; P.maxLen = 3u; P.no = 0u; P.full = false;
    const uint32_t d2 = P.pos - p2, d3 = P.pos - p3;
    const bool c2 = d2 < P.mmm && cur[-(int64_t)d2] == cur[0], c3 = d3 < P.mmm && cur[-(int64_t)d3] == cur[0];
    uint32_t ext = 0;
    if (c2) {
        out[P.no++] = 2u; out[P.no++] = d2 - 1u;
        if (cur[2 - (int64_t)d2] == cur[2]) ext = d2;
        else if (c3) { out[P.no++] = 0u; out[P.no++] = d3 - 1u; ext = d3; }
    } else if (c3) { out[P.no++] = 0u; out[P.no++] = d3 - 1u; ext = d3; }
    if (ext) {
        uint32_t l = 3u;
        while (l < P.lenLimit && cur[(int64_t)l - (int64_t)ext] == cur[l]) l++;
        P.maxLen = l; out[P.no - 2u] = l;
        P.full = l == P.lenLimit;
    }
    return P;
}

// ------------------------------------------------------------------------------------------------ HC4: one thread per position
extern "C" __global__ void __launch_bounds__(LZF_T)
gc_lzf_hc4_kernel(const uint8_t* __restrict__ src, uint32_t n, uint32_t window /* cyclicBufferSize = historySize + 1 */, uint32_t cut, uint32_t niceLen,
                  const uint32_t* __restrict__ prev2, const uint32_t* __restrict__ prev3, const uint32_t* __restrict__ prevV,
                  uint32_t stride, uint32_t* __restrict__ counts, uint32_t* __restrict__ pairs, uint32_t* __restrict__ overflow)
{
    const uint32_t i = blockIdx.x * LZF_T + threadIdx.x;
    if (i >= n) return;
    if (n - i < 4u) { counts[i] = 0; return; }                    // (lenLimit < 4: the reference only moves on)
    uint32_t tmp[4];
    LzfPos P = lzf_front(src, n, i, window, niceLen, prev2[i], prev3[i], tmp);
    uint32_t* out = pairs + (uint64_t)i * stride;
    uint32_t no = P.no;
    bool over = no > stride;
    for (uint32_t k = 0; k < no && k < stride; k++) out[k] = tmp[k];
    if (!P.full) {
        const uint8_t* cur = src + i;
        uint32_t m = prevV[i], steps = cut, maxLen = P.maxLen;
        while (steps && m) {                                      // Hc_GetMatchesSpec: newest first, strictly growing lengths
            const uint32_t delta = P.pos - m;
            if (delta >= window) break;
            if (cur[maxLen] == cur[(int64_t)maxLen - (int64_t)delta]) {
                uint32_t l = 0;
                while (l < P.lenLimit && cur[l] == cur[(int64_t)l - (int64_t)delta]) l++;
                if (l == P.lenLimit || l > maxLen) {
                    if (no + 2u <= stride) { out[no] = l; out[no + 1u] = delta - 1u; } else over = true;
                    no += 2u;
                    if (l == P.lenLimit) break;
                    maxLen = l;
                }
            }
            m = prevV[m - 1u];
            steps--;
        }
    }
    counts[i] = no < stride ? no : stride;
    if (over) atomicOr(overflow, 1u);
}

// ------------------------------------------------------------------------------------------------ BT4: one lane per hash bucket
// son[2 * i] / son[2 * i + 1] = the pair of position i (the reference's son[cyclicBufferPos << 1], + 1): subtree of smaller / larger suffixes.
__device__ __forceinline__ void lzf_bt_walk(const uint8_t* __restrict__ src, uint32_t i, uint32_t lenLimit, uint32_t curMatch, uint32_t window, uint32_t cut,
                                            uint32_t* __restrict__ son, uint32_t maxLen, bool report, uint32_t* __restrict__ out, uint32_t stride, uint32_t& no, bool& over)
{
    const uint8_t* cur = src + i;
    const uint32_t pos = i + 1u;
    uint32_t* ptr0 = son + 2ull * i + 1u;
    uint32_t* ptr1 = son + 2ull * i;
    uint32_t len0 = 0, len1 = 0;
    const uint32_t cmCheck = pos < window ? 0u : pos - window;
    if (cmCheck < curMatch) {
        do {
            const uint32_t delta = pos - curMatch;
            uint32_t* pair = son + 2ull * (curMatch - 1u);
            const uint8_t* pb = cur - delta;
            uint32_t len = len0 < len1 ? len0 : len1;
            const uint32_t pair0 = pair[0];
            if (pb[len] == cur[len]) {
                while (++len != lenLimit) if (pb[len] != cur[len]) break;
                if (report ? maxLen < len : len == lenLimit) {
                    if (report) {
                        maxLen = len;
                        if (no + 2u <= stride) { out[no] = len; out[no + 1u] = delta - 1u; } else over = true;
                        no += 2u;
                    }
                    if (len == lenLimit) { *ptr1 = pair0; *ptr0 = pair[1]; return; }
                }
            }
            if (pb[len] < cur[len]) { *ptr1 = curMatch; curMatch = pair[1]; ptr1 = pair + 1; len1 = len; }
            else { *ptr0 = curMatch; curMatch = pair0; ptr0 = pair; len0 = len; }
        } while (--cut && cmCheck < curMatch);
    }
    *ptr0 = 0; *ptr1 = 0;
}

extern "C" __global__ void __launch_bounds__(64)
gc_lzf_bt4_kernel(const uint8_t* __restrict__ src, uint32_t n, uint32_t window, uint32_t cut, uint32_t niceLen,
                  const uint32_t* __restrict__ prev2, const uint32_t* __restrict__ prev3, const uint64_t* __restrict__ sortedV, uint32_t nPart,
                  const uint32_t* __restrict__ heads, const uint32_t* __restrict__ nHeads, uint32_t* __restrict__ ticket, uint32_t* __restrict__ son,
                  uint32_t stride, uint32_t* __restrict__ counts, uint32_t* __restrict__ pairs, uint32_t* __restrict__ overflow)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t nb = *nHeads;
    bool over = false;
    for (;;) {                                                    // buckets are drawn 64 at a time
        uint32_t b0 = 0;
        if (lane == 0u) b0 = atomicAdd(ticket, 64u);
        b0 = __shfl(b0, 0);
        if (b0 >= nb) break;
        const uint32_t b = b0 + lane;
        if (b < nb) {
            uint32_t s = heads[b];
            const uint32_t key = (uint32_t)(sortedV[s] >> 32);
            uint32_t curMatch = 0;                                // what the main table holds: the bucket's previous position (+ 1)
            for (; s < nPart; s++) {
                const uint64_t it = sortedV[s];
                if ((uint32_t)(it >> 32) != key) break;
                const uint32_t i = (uint32_t)it;
                uint32_t tmp[4];
                LzfPos P = lzf_front(src, n, i, window, niceLen, prev2[i], prev3[i], tmp);
                uint32_t* out = pairs + (uint64_t)i * stride;
                uint32_t no = P.no;
                if (no > stride) over = true;
                for (uint32_t k = 0; k < no && k < stride; k++) out[k] = tmp[k];
                lzf_bt_walk(src, i, P.lenLimit, curMatch, window, cut, son, P.maxLen, !P.full, out, stride, no, over);   // (full: SkipMatchesSpec)
                counts[i] = no < stride ? no : stride;
                curMatch = i + 1u;
            }
        }
    }
    if (over) atomicOr(overflow, 1u);
}

// positions that take no part (fewer than 4 bytes left): no values
extern "C" __global__ void __launch_bounds__(LZF_T)
gc_lzf_tail_kernel(uint32_t n, uint32_t nPart, uint32_t* __restrict__ counts)
{
    const uint32_t i = nPart + blockIdx.x * LZF_T + threadIdx.x;
    if (i < n) counts[i] = 0;
}

// ---------------------------------------------------------------------------------------------------- host side
#include "gpucodec.h"
#ifdef HIPEMU
#include "hip_runtime_stub.h"
#else
#define GC_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif

namespace {
struct LzfWs {                                                    // device workspace of one call
    uint64_t *itA = nullptr, *itB = nullptr;
    uint32_t *cnt = nullptr, *sums = nullptr, *prev2 = nullptr, *prev3 = nullptr, *prevV = nullptr, *heads = nullptr, *son = nullptr, *misc = nullptr;
    ~LzfWs() { hipFree(itA); hipFree(itB); hipFree(cnt); hipFree(sums); hipFree(prev2); hipFree(prev3); hipFree(prevV); hipFree(heads); hipFree(son); hipFree(misc); }
};
// exclusive scan of a[0..m) in place
static void lzf_scan(uint32_t* a, uint32_t m, uint32_t* sums, hipStream_t st)
{
    const uint32_t chunks = (m + LZF_TILE - 1u) / LZF_TILE;
    GC_LAUNCH(gc_lzf_scan_sums_kernel, chunks, LZF_T, st, (const uint32_t*)a, m, sums);
    GC_LAUNCH(gc_lzf_scan_top_kernel, 1, 1024, st, sums, chunks);
    GC_LAUNCH(gc_lzf_scan_apply_kernel, chunks, LZF_T, st, a, m, (const uint32_t*)sums);
}
// positions sorted by the hash value of table `which`; returns the buffer that holds the result
static uint64_t* lzf_sort(const uint8_t* src, uint32_t nPart, uint32_t which, uint32_t hashMask, LzfWs& W, hipStream_t st)
{
    const uint32_t bits = which == 0u ? 10u : (which == 1u ? 16u : 32u - (uint32_t)__builtin_clz(hashMask));
    const uint32_t nTiles = (nPart + LZF_TILE - 1u) / LZF_TILE;
    GC_LAUNCH(gc_lzf_items_kernel, (nPart + LZF_T - 1u) / LZF_T, LZF_T, st, src, nPart, which, hashMask, W.itA);
    uint64_t *in = W.itA, *out = W.itB;
    for (uint32_t sh = 0; sh < bits; sh += 8u) {
        GC_LAUNCH(gc_lzf_count_kernel, nTiles, LZF_T, st, (const uint64_t*)in, nPart, 32u + sh, nTiles, W.cnt);
        lzf_scan(W.cnt, 256u * nTiles, W.sums, st);
        GC_LAUNCH(gc_lzf_scatter_kernel, nTiles, LZF_T, st, (const uint64_t*)in, nPart, 32u + sh, nTiles, (const uint32_t*)W.cnt, out);
        uint64_t* t = in; in = out; out = t;
    }
    return in;
}
}

// IMatchFinder2::GetMatches for every position of a buffer in device memory (C/LzFind.h:127-140; see the top of this file).
//   bt            0 = HC4 (hash chain), 1 = BT4 (binary tree)
//   historySize   the reference's historySize: links older than historySize (cyclicBufferSize = historySize + 1) are not followed
//   cut, niceLen  cutValue and matchMaxLen of MatchFinder_Create
//   d_counts[i]   number of uint32 values of position i (2 per match), at d_pairs[i * stride ...]: length, distance - 1, ...
// GC_ERR_DST_SMALL if some position has more than `stride` values (its list is cut at stride).  Synchronous, default stream of the current device.
extern "C" int gc_lzfind_get_matches_device(const void* d_src, size_t n, int bt, uint32_t historySize, uint32_t cut, uint32_t niceLen,
                                            uint32_t* d_counts, uint32_t* d_pairs, uint32_t stride)
{
    if ((!d_src && n) || !d_counts || !d_pairs || cut == 0u || niceLen < 4u || niceLen > 273u || stride < 4u || n >= 0x7FFFFFF0ull || historySize == 0u) return GC_ERR_PARAM;
    if (n == 0) return GC_OK;
    const uint8_t* src = (const uint8_t*)d_src;
    const uint32_t N = (uint32_t)n, nPart = N >= 4u ? N - 3u : 0u;             // positions with at least 4 bytes left take part
    const uint32_t window = historySize + 1u, hashMask = gc_lzf_hash_mask(historySize);
    hipStream_t st = (hipStream_t)0;
    if (nPart == 0u) { GC_LAUNCH(gc_lzf_tail_kernel, 1, LZF_T, st, N, 0u, d_counts); return hipStreamSynchronize(st) == hipSuccess ? GC_OK : GC_ERR_HIP; }
    LzfWs W;
    const uint32_t nTiles = (nPart + LZF_TILE - 1u) / LZF_TILE;
    const size_t cntWords = (size_t)256u * nTiles, sumWords = (cntWords + LZF_TILE - 1u) / LZF_TILE + 1u;
    if (hipMalloc((void**)&W.itA, (size_t)nPart * 8u) != hipSuccess || hipMalloc((void**)&W.itB, (size_t)nPart * 8u) != hipSuccess ||
        hipMalloc((void**)&W.cnt, cntWords * 4u) != hipSuccess || hipMalloc((void**)&W.sums, sumWords * 4u) != hipSuccess ||
        hipMalloc((void**)&W.prev2, (size_t)N * 4u) != hipSuccess || hipMalloc((void**)&W.prev3, (size_t)N * 4u) != hipSuccess ||
        hipMalloc((void**)&W.prevV, (size_t)N * 4u) != hipSuccess || hipMalloc((void**)&W.misc, 64u) != hipSuccess ||
        (bt && (hipMalloc((void**)&W.heads, (size_t)nPart * 4u) != hipSuccess || hipMalloc((void**)&W.son, (size_t)N * 8u) != hipSuccess)))
        return GC_ERR_NOMEM;
    if (hipMemsetAsync(W.misc, 0, 64u, st) != hipSuccess) return GC_ERR_HIP;
    uint32_t* overflow = W.misc; uint32_t* nHeads = W.misc + 1; uint32_t* ticket = W.misc + 2;
    const uint32_t gP = (nPart + LZF_T - 1u) / LZF_T;
    for (uint32_t which = 0; which < 3u; which++) {
        const uint64_t* sorted = lzf_sort(src, nPart, which, hashMask, W, st);
        uint32_t* prev = which == 0u ? W.prev2 : (which == 1u ? W.prev3 : W.prevV);
        const bool wantHeads = bt && which == 2u;
        GC_LAUNCH(gc_lzf_link_kernel, gP, LZF_T, st, sorted, nPart, prev, wantHeads ? W.heads : (uint32_t*)nullptr, nHeads);
        if (wantHeads)      // (the sorted main-table items stay where they are: nothing is sorted after them)
            GC_LAUNCH(gc_lzf_bt4_kernel, 2048, 64, st, src, N, window, cut, niceLen, (const uint32_t*)W.prev2, (const uint32_t*)W.prev3, sorted, nPart,
                      (const uint32_t*)W.heads, (const uint32_t*)nHeads, ticket, W.son, stride, d_counts, d_pairs, overflow);
    }
    if (!bt) GC_LAUNCH(gc_lzf_hc4_kernel, (N + LZF_T - 1u) / LZF_T, LZF_T, st, src, N, window, cut, niceLen, (const uint32_t*)W.prev2, (const uint32_t*)W.prev3,
                       (const uint32_t*)W.prevV, stride, d_counts, d_pairs, overflow);
    else GC_LAUNCH(gc_lzf_tail_kernel, 1, LZF_T, st, N, nPart, d_counts);
    uint32_t ov = 0;
    if (hipMemcpyAsync(&ov, overflow, 4u, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) return GC_ERR_HIP;
    return ov ? GC_ERR_DST_SMALL : GC_OK;
}
