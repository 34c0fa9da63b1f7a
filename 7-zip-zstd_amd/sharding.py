"""Range sharding of one input over the GPUs of a node.

The zstd path produces one independent frame per 128 KiB block, and the reference decoder accepts any number of
concatenated frames (CPP/7zip/Compress/ZstdDecoder.cpp:145-158).  So N ranks can compress N contiguous,
grain-aligned ranges with no data-path collective; the host concatenates the compressed ranges in rank order.  This is
the same job split the reference's own multi-threaded front end makes (ZSTDMT jobs, C/zstd/zstdmt_compress.c:1184-1247;
brotli-mt chunks, C/zstdmt/brotli-mt_compress.c:209-333), with GPUs in place of worker threads.

torch.distributed is used for the gather of the compressed byte strings only (gloo on CPU tests, RCCL on GPUs).
"""
import numpy as np

GRAIN_ZSTD = 128 * 1024
FRAME_ZSTD = 64 * GRAIN_ZSTD      # level >= 3: 8 MiB frames (windowed match finder, csrc/gc_mf.h GC_MF_MAX_FRAME_BLOCKS)


def zstd_grain(level, n, world):
    """Independence grain of the zstd path: one block at levels 1-2; a whole 8 MiB frame at level >= 3 when every rank gets at
    least one (a range that starts inside a frame would only cut that frame's window short, the stream stays valid)."""
    if level >= 3 and n >= world * FRAME_ZSTD:
        return FRAME_ZSTD
    return GRAIN_ZSTD


def shard_ranges(n, world, grain=GRAIN_ZSTD):
    """Contiguous [start, end) per rank, every boundary a multiple of `grain`, sizes as even as the grain allows.
    The concatenation of the per-rank streams is a valid stream for the whole input (independent frames)."""
    units = (n + grain - 1) // grain
    base, extra = divmod(units, world)
    out, u = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        s, e = min(u * grain, n), min((u + cnt) * grain, n)
        out.append((s, e))
        u += cnt
    return out


def compress_sharded(encoder, data, rank, world, dist=None, grain=GRAIN_ZSTD):
    """Every rank compresses its own range of `data` (numpy uint8, identical on all ranks or at least valid on its own
    range); rank 0 returns the concatenated stream, the others return None."""
    s, e = shard_ranges(data.size, world, grain)[rank]
    if e > s or (rank == 0 and data.size == 0):
        mine = np.asarray(encoder.code(data[s:e]), dtype=np.uint8)
    else:
        mine = np.empty(0, dtype=np.uint8)     # an empty range contributes nothing (not even an empty frame)
    if world == 1 or dist is None:
        return mine
    pieces = [None] * world if rank == 0 else None
    dist.gather_object(mine.tobytes(), pieces, dst=0)
    if rank != 0:
        return None
    return np.frombuffer(b"".join(pieces), dtype=np.uint8)
