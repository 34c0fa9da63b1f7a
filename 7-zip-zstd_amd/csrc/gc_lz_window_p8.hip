// gc_lz_window_p8.hip -- the FAST geometry of the windowed match finder (gc_mf.h: 256 partitions, 8 KiB tiles): the W1..W5 kernels of
// gc_lz_window.hip compiled once more with GC_MF_FAST, their names suffixed _p8.  Used by zstd levels 3-6 and brotli qualities 3-4.
#define GC_MF_FAST 1
#include "gc_lz_window.hip"
