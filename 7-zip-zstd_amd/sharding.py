"""Range sharding of one input over the GPUs of a node, one process per GPU.

All three codecs produce streams that concatenate: zstd frames (the reference decoder accepts any number of them,
CPP/7zip/Compress/ZstdDecoder.cpp:145-158), brotli-mt frames (C/zstdmt/brotli-mt_decompress.c:240-351) and runs of LZMA2
chunks that start with a dictionary reset and share ONE end marker (C/Lzma2Dec.c:97).  So N ranks compress N contiguous ranges
cut at multiples of the codec's independence grain (gc_codec_grain: 8 MiB match-finder frames for zstd and FLZMA2; the brotli-mt chunk of `level` MiB) with no data-path collective, and the host concatenates
the compressed ranges in rank order.  This is the job split of the reference's own multi-threaded front ends (ZSTDMT jobs,
C/zstd/zstdmt_compress.c:1184-1247; brotli-mt chunks, C/zstdmt/brotli-mt_compress.c:209-333) with GPUs in place of worker
threads.  The in-process counterpart (several GPUs driven by one process, host threads) is gc_multi in csrc/gc_multi.hip.

torch.distributed is used for the gather of the compressed byte strings only (gloo on CPU tests, RCCL on GPUs).
"""
import numpy as np

GRAIN_ZSTD = 128 * 1024
FRAME_ZSTD = 64 * GRAIN_ZSTD      # 8 MiB frames (windowed match finder, csrc/gc_mf.h GC_MF_MAX_FRAME_BLOCKS)
FLZMA2_NO_END_MARK = 1            # include/gpucodec.h GC_FLZMA2_NO_END_MARK


def codec_of(encoder):
    """"zstd" / "flzma2" / "brotli" for one of the package's encoder objects."""
    name = type(encoder).__name__.lower()
    for k in ("zstd", "flzma2", "brotli"):
        if k in name:
            return k
    raise TypeError("not a gpucodec encoder: %r" % (encoder,))


def codec_grain(codec, level):
    """What gc_codec_grain returns (kept in Python too so that range planning needs no library handle)."""
    if codec == "brotli":
        return max(1, min(11, int(level))) * 8 * GRAIN_ZSTD
    if codec == "zstd" and int(level) >= 16:
        return 4 * FRAME_ZSTD                                     # overlapping finder frames inside 32 MiB zstd frames (csrc/gc_api.hip zstd_group_blocks)
    if codec == "flzma2" and int(level) >= 7:
        return 8 * FRAME_ZSTD                                     # ... inside groups of 64 MiB (flzma2_group_blocks)
    if codec == "flzma2" and int(level) >= 5:
        return 2 * FRAME_ZSTD                                     # ... of 16 MiB at levels 5-6 (round 5)
    return FRAME_ZSTD                                             # zstd and FLZMA2: the windowed finder at every level


def zstd_grain(level, n, world):
    """Grain used for a zstd input of n bytes on `world` ranks: a whole 8 MiB frame when every rank gets at
    least one; otherwise one block (a range that starts inside a frame only cuts that frame's window short, the stream stays
    valid)."""
    if n >= world * FRAME_ZSTD:
        return FRAME_ZSTD
    return GRAIN_ZSTD


def shard_ranges(n, world, grain=GRAIN_ZSTD):
    """Contiguous [start, end) per rank, every boundary a multiple of `grain`, sizes as even as the grain allows.
    The concatenation of the per-rank streams is a valid stream for the whole input (independent frames / chunks)."""
    units = (n + grain - 1) // grain
    base, extra = divmod(units, world)
    out, u = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        s, e = min(u * grain, n), min((u + cnt) * grain, n)
        out.append((s, e))
        u += cnt
    return out


def compress_sharded(encoder, data, rank, world, dist=None, grain=None):
    """Every rank compresses its own range of `data` (numpy uint8, identical on all ranks or at least valid on its own
    range); rank 0 returns the concatenated stream, the others return None.

    grain=None picks the codec's own grain when every rank gets at least one unit of it, else one 128 KiB block (brotli: always
    the brotli-mt chunk -- a frame must not be cut).  FLZMA2: every rank codes its range with GC_FLZMA2_NO_END_MARK and rank 0
    appends the single end marker after the last piece."""
    codec = codec_of(encoder)
    if grain is None:
        grain = codec_grain(codec, encoder.level)
        if codec != "brotli" and data.size < world * grain:
            grain = GRAIN_ZSTD
    s, e = shard_ranges(data.size, world, grain)[rank]
    if codec == "flzma2":
        mine = np.asarray(encoder.code(data[s:e], flags=FLZMA2_NO_END_MARK), dtype=np.uint8) if e > s else np.empty(0, dtype=np.uint8)
    elif e > s or (rank == 0 and data.size == 0):
        mine = np.asarray(encoder.code(data[s:e]), dtype=np.uint8)
    else:
        mine = np.empty(0, dtype=np.uint8)     # an empty range contributes nothing (not even an empty frame)
    tail = b"\x00" if codec == "flzma2" else b""       # LZMA2 end of stream, once
    if world == 1 or dist is None:
        return np.frombuffer(mine.tobytes() + tail, dtype=np.uint8) if tail else mine
    pieces = [None] * world if rank == 0 else None
    dist.gather_object(mine.tobytes(), pieces, dst=0)
    if rank != 0:
        return None
    return np.frombuffer(b"".join(pieces) + tail, dtype=np.uint8)
