// gc_lzfind.h -- shared definitions of the mainline LZMA match finders on the device (gc_lzfind.hip).
#pragma once
#include <stdint.h>

// MatchFinder_GetHashMask (C/LzFind.c:345-372) for numHashBytes = 4: the main table has hashMask + 1 slots
static inline uint32_t gc_lzf_hash_mask(uint32_t historySize)
{
    uint32_t hs = historySize;
    if (hs) hs--;
    hs |= hs >> 1; hs |= hs >> 2; hs |= hs >> 4; hs |= hs >> 8;
    hs >>= 1;
    if (hs >= (1u << 24)) hs >>= 1;
    return hs | 0xFFFFu;
}
