"""7-zip-zstd_amd -- host-side Python mirror of the codec interface, over the C ABI (include/gpucodec.h).

The directory name is not an importable identifier; load it with

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("sevenzip_zstd_amd", "<repo>/7-zip-zstd_amd/__init__.py")
    mod = importlib.util.module_from_spec(spec); sys.modules[spec.name] = mod; spec.loader.exec_module(mod)

(`__graft_entry__.load_package()` does exactly that).  The class below mirrors the reference's encoder
object NCompress::NZSTD::CEncoder (CPP/7zip/Compress/ZstdEncoder.h:35-78): create, set the level
(SetCoderProperties kLevel, ZstdEncoder.cpp:51-70), Code() bytes -> zstd stream.  Everything runs through
csrc/libgpucodec.so built by hipcc for gfx950; there is no CPU path and construction raises if the library
or the GPU is missing.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libgpucodec.so")
HOOKS_LIB_PATH = os.path.join(HERE, "csrc", "libgpucodec_hooks.so")     # test build: GC_* environment hooks compiled in (tests / tools only)

GC_OK = 0
_ERR = {-1: "GC_ERR_NO_DEVICE", -2: "GC_ERR_HIP", -3: "GC_ERR_NOMEM", -4: "GC_ERR_DST_SMALL", -5: "GC_ERR_PARAM", -6: "GC_ERR_CORRUPT", -7: "GC_ERR_UNSUPPORTED"}

EXPORTS = ["gc_test_hooks_enabled", "gc_lzfind_get_matches_device", "gc_device_count", "gc_ctx_create", "gc_ctx_destroy", "gc_last_error_message", "gc_zstd_compress_bound",
           "gc_zstd_compress_device", "gc_zstd_finish", "gc_zstd_compress_host", "gc_zstd_last_timing", "gc_ctx_stream",
           "gc_zstd_set_phase_profile", "gc_zstd_phase_profile", "gc_mf_last_timing", "gc_mf_price_timing", "gc_mf_pass_timing", "gc_host_begin_pre", "gc_codec_compress_host_pre",
           "gc_flzma2_compress_bound", "gc_flzma2_dict_prop", "gc_flzma2_compress_device", "gc_flzma2_finish", "gc_flzma2_compress_host",
           "gc_flzma2_last_timing",
           "gc_brotli_compress_bound", "gc_brotli_compress_device", "gc_brotli_finish", "gc_brotli_compress_host", "gc_brotli_last_timing",
           "gc_ctx_set_option", "gc_crc32_device", "gc_codec_grain", "gc_codec_compress_bound", "gc_host_begin", "gc_host_size", "gc_host_fetch", "gc_codec_compress_host",
           "gc_host_alloc", "gc_host_free", "gc_multi_create", "gc_multi_destroy", "gc_multi_workers", "gc_multi_last_error",
           "gc_multi_piece_bytes", "gc_multi_compress_host",
           "gc_bra_convert_device", "gc_bra_x86_convert_device", "gc_delta_convert_device", "gc_zstd_scan_frames", "gc_zstd_scan_prefix", "gc_zstd_decompress_device", "gc_zstd_decompress_host", "gc_zstd_decompress_timing", "gc_zstd_decompress_kernel_timing", "gc_zstd_decompress_wide_rounds", "gc_zstd_decompress_selfcheck", "gc_filter_host",
           "gc_brotli_scan_prefix", "gc_brotli_dec_set_dictionary", "gc_brotli_dec_has_dictionary", "gc_brotli_decompress_device", "gc_brotli_decompress_host", "gc_brotli_decompress_timing"]

CODEC_ZSTD, CODEC_FLZMA2, CODEC_BROTLI = 0, 1, 2
CODEC_IDS = {"zstd": CODEC_ZSTD, "flzma2": CODEC_FLZMA2, "brotli": CODEC_BROTLI}


class GpuCodecError(RuntimeError):
    pass


def load_library(path=None):
    """dlopen libgpucodec.so and declare the prototypes of include/gpucodec.h."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise GpuCodecError("HIP extension %s is missing: run __graft_entry__.build()" % path)
    lib = C.CDLL(path)
    lib.gc_test_hooks_enabled.restype = C.c_int
    lib.gc_lzfind_get_matches_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint]
    lib.gc_lzfind_get_matches_device.restype = C.c_int
    lib.gc_device_count.restype = C.c_int
    lib.gc_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.gc_ctx_create.restype = C.c_int
    lib.gc_ctx_destroy.argtypes = [C.c_void_p]
    lib.gc_ctx_destroy.restype = None
    lib.gc_last_error_message.argtypes = [C.c_void_p]
    lib.gc_last_error_message.restype = C.c_char_p
    lib.gc_zstd_compress_bound.argtypes = [C.c_size_t]
    lib.gc_zstd_compress_bound.restype = C.c_size_t
    lib.gc_zstd_compress_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    lib.gc_zstd_compress_device.restype = C.c_int
    lib.gc_zstd_finish.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.gc_zstd_finish.restype = C.c_int
    lib.gc_zstd_compress_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]
    lib.gc_zstd_compress_host.restype = C.c_int
    lib.gc_zstd_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gc_zstd_last_timing.restype = C.c_int
    lib.gc_mf_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gc_mf_last_timing.restype = C.c_int
    lib.gc_mf_price_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gc_mf_price_timing.restype = C.c_int
    lib.gc_mf_pass_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gc_mf_pass_timing.restype = C.c_int
    lib.gc_zstd_set_phase_profile.argtypes = [C.c_void_p, C.c_int]
    lib.gc_zstd_set_phase_profile.restype = C.c_int
    lib.gc_zstd_phase_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.gc_zstd_phase_profile.restype = C.c_int
    lib.gc_flzma2_compress_bound.argtypes = [C.c_size_t]
    lib.gc_flzma2_compress_bound.restype = C.c_size_t
    lib.gc_flzma2_dict_prop.argtypes = [C.c_int]
    lib.gc_flzma2_dict_prop.restype = C.c_ubyte
    lib.gc_flzma2_compress_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint]
    lib.gc_flzma2_compress_device.restype = C.c_int
    lib.gc_flzma2_finish.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.gc_flzma2_finish.restype = C.c_int
    lib.gc_flzma2_compress_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.POINTER(C.c_size_t)]
    lib.gc_flzma2_compress_host.restype = C.c_int
    lib.gc_flzma2_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gc_flzma2_last_timing.restype = C.c_int
    lib.gc_brotli_compress_bound.argtypes = [C.c_size_t]
    lib.gc_brotli_compress_bound.restype = C.c_size_t
    lib.gc_brotli_compress_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    lib.gc_brotli_compress_device.restype = C.c_int
    lib.gc_brotli_finish.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.gc_brotli_finish.restype = C.c_int
    lib.gc_brotli_compress_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_size_t)]
    lib.gc_brotli_compress_host.restype = C.c_int
    lib.gc_brotli_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gc_brotli_last_timing.restype = C.c_int
    lib.gc_ctx_stream.argtypes = [C.c_void_p]
    lib.gc_ctx_stream.restype = C.c_void_p
    lib.gc_crc32_device.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
    lib.gc_crc32_device.restype = C.c_int
    lib.gc_ctx_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.gc_ctx_set_option.restype = C.c_int
    lib.gc_codec_grain.argtypes = [C.c_int, C.c_int]
    lib.gc_codec_grain.restype = C.c_size_t
    lib.gc_codec_compress_bound.argtypes = [C.c_int, C.c_size_t]
    lib.gc_codec_compress_bound.restype = C.c_size_t
    lib.gc_host_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_uint]
    lib.gc_host_begin.restype = C.c_int
    lib.gc_host_size.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.gc_host_size.restype = C.c_int
    lib.gc_host_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.gc_host_fetch.restype = C.c_int
    lib.gc_codec_compress_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.POINTER(C.c_size_t)]
    lib.gc_codec_compress_host.restype = C.c_int
    lib.gc_host_alloc.argtypes = [C.c_size_t]
    lib.gc_host_alloc.restype = C.c_void_p
    lib.gc_host_free.argtypes = [C.c_void_p]
    lib.gc_host_free.restype = None
    lib.gc_multi_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_int]
    lib.gc_multi_create.restype = C.c_int
    lib.gc_multi_destroy.argtypes = [C.c_void_p]
    lib.gc_multi_destroy.restype = None
    lib.gc_multi_workers.argtypes = [C.c_void_p]
    lib.gc_multi_workers.restype = C.c_int
    lib.gc_multi_last_error.argtypes = [C.c_void_p]
    lib.gc_multi_last_error.restype = C.c_char_p
    lib.gc_multi_piece_bytes.argtypes = [C.c_int, C.c_int]
    lib.gc_multi_piece_bytes.restype = C.c_size_t
    lib.gc_multi_compress_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.gc_multi_compress_host.restype = C.c_int
    lib.gc_zstd_scan_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
    lib.gc_zstd_scan_frames.restype = C.c_int
    lib.gc_zstd_decompress_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.gc_zstd_decompress_device.restype = C.c_int
    lib.gc_zstd_decompress_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.gc_zstd_decompress_host.restype = C.c_int
    lib.gc_zstd_decompress_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gc_zstd_decompress_timing.restype = C.c_int
    lib.gc_brotli_scan_prefix.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_size_t)]
    lib.gc_brotli_scan_prefix.restype = C.c_int
    lib.gc_brotli_dec_set_dictionary.argtypes = [C.c_void_p, C.c_size_t]
    lib.gc_brotli_dec_set_dictionary.restype = C.c_int
    lib.gc_brotli_dec_has_dictionary.argtypes = []
    lib.gc_brotli_dec_has_dictionary.restype = C.c_int
    lib.gc_brotli_decompress_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.gc_brotli_decompress_device.restype = C.c_int
    lib.gc_brotli_decompress_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.gc_brotli_decompress_host.restype = C.c_int
    lib.gc_brotli_decompress_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.gc_brotli_decompress_timing.restype = C.c_int
    return lib


BRA_KINDS = {"ARM64": 0, "ARM": 1, "ARMT": 2, "PPC": 3, "SPARC": 4, "IA64": 5, "RISCV": 6}


def bra_convert_device(kind, src_ptr, dst_ptr, n, pc=0, encoding=True, lib_path=None):
    """Branch converter (C/Bra.c) over n bytes at device pointers (under the emulator: host pointers); returns the processed byte count."""
    lib = load_library(lib_path)
    lib.gc_bra_convert_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_size_t)]
    done = C.c_size_t(0)
    rc = lib.gc_bra_convert_device(BRA_KINDS[kind], src_ptr, dst_ptr, n, pc & 0xFFFFFFFF, 1 if encoding else 0, C.byref(done))
    if rc != GC_OK:
        raise GpuCodecError("gc_bra_convert_device failed: %s" % _ERR.get(rc, rc))
    return done.value


def bra_x86_convert_device(src_ptr, dst_ptr, n, pc=0, encoding=True, state=0, lib_path=None):
    """X86 branch converter (C/Bra86.c) over n bytes at device pointers, out of place; returns (processed bytes, state out)."""
    lib = load_library(lib_path)
    lib.gc_bra_x86_convert_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_size_t)]
    done = C.c_size_t(0); st = C.c_uint32(state)
    rc = lib.gc_bra_x86_convert_device(src_ptr, dst_ptr, n, pc & 0xFFFFFFFF, 1 if encoding else 0, C.byref(st), C.byref(done))
    if rc != GC_OK:
        raise GpuCodecError("gc_bra_x86_convert_device failed: %s" % _ERR.get(rc, rc))
    return done.value, st.value


def delta_convert_device(src_ptr, dst_ptr, n, delta, encoding=True, state=None, lib_path=None):
    """Delta filter (C/Delta.c) over n bytes at device pointers; state = bytes-like of 256 (None: zeros); returns the state behind the buffer."""
    lib = load_library(lib_path)
    lib.gc_delta_convert_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.POINTER(C.c_ubyte)]
    st = (C.c_ubyte * 256)(*(bytes(state) if state is not None else bytes(256)))
    rc = lib.gc_delta_convert_device(src_ptr, dst_ptr, n, int(delta), 1 if encoding else 0, st)
    if rc != GC_OK:
        raise GpuCodecError("gc_delta_convert_device failed: %s" % _ERR.get(rc, rc))
    return bytes(st)


class ZstdFrame(C.Structure):
    """gc_zstd_frame of include/gpucodec.h"""
    _fields_ = [("src_off", C.c_uint64), ("src_size", C.c_uint64), ("dst_off", C.c_uint64), ("content_size", C.c_uint64),
                ("flags", C.c_uint32), ("header_size", C.c_uint32), ("n_blocks", C.c_uint32), ("reserved", C.c_uint32)]


class BrotliChunk(C.Structure):
    """gc_brotli_chunk of include/gpucodec.h"""
    _fields_ = [("src_off", C.c_uint64), ("src_size", C.c_uint32), ("capacity", C.c_uint32)]


def crc32_device(ptr, n, lib_path=None):
    """CRC-32 (as C/7zCrc.c) of n bytes at a device pointer (under the emulator: any host pointer)."""
    v = C.c_uint32(0)
    rc = load_library(lib_path).gc_crc32_device(ptr, n, C.byref(v))
    if rc != GC_OK:
        raise GpuCodecError("gc_crc32_device failed: %s" % _ERR.get(rc, rc))
    return v.value


def lzfind_get_matches_device(src_ptr, n, counts_ptr, pairs_ptr, stride, history=1 << 20, bt=False, cut=32, nice=64, lib_path=None):
    """IMatchFinder2::GetMatches of the reference's HC4 (bt=False) / BT4 (bt=True) for every position of n bytes at a device pointer (under the
    emulator: host pointers): counts[i] uint32 values at pairs[i * stride ...] (length, distance - 1, ...).  (C/LzFind.c:1362, :1219)"""
    rc = load_library(lib_path).gc_lzfind_get_matches_device(src_ptr, n, 1 if bt else 0, history, cut, nice, counts_ptr, pairs_ptr, stride)
    if rc != GC_OK:
        raise GpuCodecError("gc_lzfind_get_matches_device failed: %s" % _ERR.get(rc, rc))


def lzfind_matches(data, history=1 << 20, bt=False, cut=32, nice=64, device=None, lib_path=None):
    """Host convenience over lzfind_get_matches_device: (values per position, all values in order) -- the layout of the oracle's match lists.
    device=None: the pointers are host pointers (emulator library); else the buffers go through torch on that device."""
    import numpy as np
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
    stride = 2 * (min(cut, nice) + 2)
    if device is None:
        counts = np.zeros(max(a.size, 1), dtype=np.uint32); pairs = np.zeros(max(a.size, 1) * stride, dtype=np.uint32)
        lzfind_get_matches_device(a.ctypes.data, a.size, counts.ctypes.data, pairs.ctypes.data, stride, history, bt, cut, nice, lib_path)
    else:
        import torch
        dev = "cuda:%d" % device
        d = torch.from_numpy(a).to(dev) if a.size else torch.zeros(1, dtype=torch.uint8, device=dev)
        dc = torch.zeros(max(a.size, 1), dtype=torch.int32, device=dev); dp = torch.zeros(max(a.size, 1) * stride, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        lzfind_get_matches_device(d.data_ptr(), a.size, dc.data_ptr(), dp.data_ptr(), stride, history, bt, cut, nice, lib_path)
        counts = dc.cpu().numpy().view(np.uint32); pairs = dp.cpu().numpy().view(np.uint32)
    counts = counts[:a.size]
    p2 = pairs.reshape(-1, stride)[:a.size]
    mask = np.arange(stride, dtype=np.uint32)[None, :] < counts[:, None]
    return counts, p2[mask]


def codec_grain(codec, level, lib_path=None):
    """Independence grain in bytes of a codec ("zstd" / "flzma2" / "brotli") at a level (gc_codec_grain)."""
    return load_library(lib_path).gc_codec_grain(CODEC_IDS[codec], int(level))


class MultiEncoder:
    """The host scheduler gc_multi (csrc/gc_multi.hip): one host buffer range-split over the contexts of the node's GPUs,
    compressed pieces concatenated in order -- the job front end of ZSTDMT / brotli-mt (C/zstd/zstdmt_compress.c:1184-1247,
    C/zstdmt/brotli-mt_compress.c:209-333) with GPU contexts in place of worker threads."""

    def __init__(self, codec="zstd", level=None, devices=None, ctx_per_device=2, lib_path=None):
        self._lib = load_library(lib_path)
        self.codec = CODEC_IDS[codec]
        self.level = {"zstd": 3, "flzma2": 5, "brotli": 6}[codec] if level is None else int(level)
        self._m = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices) if devices else None
        rc = self._lib.gc_multi_create(C.byref(self._m), arr, len(devices) if devices else 0, ctx_per_device)
        if rc != GC_OK:
            self._m = C.c_void_p()
            raise GpuCodecError("gc_multi_create failed: %s (no GPU fallback exists)" % _ERR.get(rc, rc))

    def close(self):
        if getattr(self, "_m", None) and self._m.value:
            self._lib.gc_multi_destroy(self._m)
            self._m = C.c_void_p()

    __del__ = close

    def workers(self):
        return self._lib.gc_multi_workers(self._m)

    def code(self, data, flags=0, piece_bytes=0):
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        cap = self._lib.gc_codec_compress_bound(self.codec, a.size) + 1
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self._lib.gc_multi_compress_host(self._m, self.codec, a.ctypes.data, a.size, out.ctypes.data, cap, self.level, flags, piece_bytes, C.byref(n))
        if rc != GC_OK:
            msg = self._lib.gc_multi_last_error(self._m)
            raise GpuCodecError("gc_multi_compress_host failed: %s (%s)" % (_ERR.get(rc, rc), msg.decode() if msg else ""))
        return out[:n.value]


class _EncoderBase:
    def __init__(self, device=0, level=3, lib_path=None):
        self._lib = load_library(lib_path)
        self._ctx = C.c_void_p()
        rc = self._lib.gc_ctx_create(C.byref(self._ctx), device)
        if rc != GC_OK:
            self._ctx = C.c_void_p()
            raise GpuCodecError("gc_ctx_create(device=%d) failed: %s (no GPU fallback exists)" % (device, _ERR.get(rc, rc)))
        self.level = level

    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.gc_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p() if C is not None else None      # (C is gone when this runs at interpreter shutdown)

    __del__ = close

    def _check(self, rc, what):
        if rc != GC_OK:
            msg = self._lib.gc_last_error_message(self._ctx)
            raise GpuCodecError("%s failed: %s (%s)" % (what, _ERR.get(rc, rc), msg.decode() if msg else ""))

    def set_level(self, level):          # SetCoderProperties(kLevel)
        self.level = int(level)

    OPT_ZSTD_SEEK_TABLE, OPT_BROTLI_PLAIN = 1, 2

    def set_option(self, option, value=1):
        self._check(self._lib.gc_ctx_set_option(self._ctx, int(option), int(value)), "gc_ctx_set_option")

    def stream(self):
        return self._lib.gc_ctx_stream(self._ctx)

    CODEC = None                      # GC_CODEC_* of the subclass (0 zstd, 1 flzma2, 2 brotli)
    FILTER_X86, FILTER_DELTA = 100, 101

    def code_pre(self, data, filter=0, pc=0, delta=1, want_crc=True, flags=0):
        """gc_codec_compress_host_pre: one host buffer -> (compressed stream of the FILTERED bytes, CRC-32 of the raw bytes, bytes the converter converted).
        What a 7z folder `filter -> coder` holds, with the reader's CRC, from one transfer (include/gpucodec.h gc_pre)."""
        import numpy as np

        class GcPre(C.Structure):
            _fields_ = [("filter", C.c_int), ("pc", C.c_uint32), ("delta", C.c_uint), ("want_crc", C.c_int), ("state", C.c_ubyte * 256), ("crc", C.c_uint32), ("processed", C.c_uint64)]
        x = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
        pre = GcPre(); pre.filter = int(filter); pre.pc = int(pc); pre.delta = int(delta); pre.want_crc = 1 if want_crc else 0
        self._lib.gc_codec_compress_bound.restype = C.c_size_t; self._lib.gc_codec_compress_bound.argtypes = [C.c_int, C.c_size_t]
        cap = int(self._lib.gc_codec_compress_bound(self.CODEC, x.size))
        out = np.empty(cap, dtype=np.uint8); n = C.c_size_t(0)
        self._lib.gc_codec_compress_host_pre.restype = C.c_int
        self._lib.gc_codec_compress_host_pre.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_uint, C.c_void_p, C.POINTER(C.c_size_t)]
        self._check(self._lib.gc_codec_compress_host_pre(self._ctx, self.CODEC, x.ctypes.data_as(C.c_void_p), x.size, out.ctypes.data_as(C.c_void_p), cap, int(self.level), int(flags),
                                                         C.byref(pre), C.byref(n)), "gc_codec_compress_host_pre")
        return out[: n.value].copy(), int(pre.crc), int(pre.processed)

    MF_KERNELS = ("mf.count", "mf.scan", "mf.scatter", "mf.link", "mf.verify", "mf.parse")

    def mf_timing_ms(self):
        """Stage durations of the windowed match finder in the last call, or None if the block-local finder ran."""
        ms = (C.c_float * 6)()
        if self._lib.gc_mf_last_timing(self._ctx, ms) != GC_OK:
            return None
        d = dict(zip(self.MF_KERNELS, [float(x) for x in ms]))
        pm = (C.c_float * 4)()
        if self._lib.gc_mf_price_timing(self._ctx, pm) == GC_OK:      # price-based parse ran: split "parse" into its kernels
            d["mf.parse"] = float(pm[0]) + float(pm[3])               # W6 twice (greedy + statistics, then following W7's records)
            d["mf.short"] = float(pm[1])
            d["mf.dp"] = float(pm[2])
        ps = (C.c_float * 4)()
        if self._lib.gc_mf_pass_timing(self._ctx, ps) == GC_OK and max(ps[1], ps[2], ps[3]) > 0.02:   # extra passes ran: split "verify"
            d["mf.verify"] = float(ps[0])           # (a part that did not run reads the few microseconds between two event records)
            for k, v in (("mf.far", ps[1]), ("mf.deepen", ps[2]), ("mf.shortpass", ps[3])):
                if v > 0.02:
                    d[k] = float(v)
        return d


class Flzma2Encoder(_EncoderBase):
    CODEC = 1
    L2_PHASES = ("l2.header_cycles", "l2.generate_cycles", "l2.apply_cycles", "l2.apply_steps", "l2.rounds", "l2.segments")

    def set_phase_profile(self, on=True):
        self._check(self._lib.gc_zstd_set_phase_profile(self._ctx, 1 if on else 0), "gc_zstd_set_phase_profile")

    def phase_profile(self):
        """Sums over all model segments of the last call: shader cycles per L2 phase, 64-event application steps, rounds played."""
        v = (C.c_double * 16)()
        self._check(self._lib.gc_zstd_phase_profile(self._ctx, v), "gc_zstd_phase_profile")
        return dict(zip(self.L2_PHASES, [float(x) for x in v[:6]]))

    """Mirror of NCompress::NLzma2::CFastEncoder (CPP/7zip/Compress/Lzma2Encoder.h:60-100; Code() at Lzma2Encoder.cpp:260-350):
    bytes -> LZMA2 chunk stream; `coder_props()` is what WriteCoderProperties emits (1 byte dictionary size, :353-364)."""

    KERNELS = ("lz", "prep", "model", "rc", "plan", "emit", "total")
    NO_END_MARK = 1

    def __init__(self, device=0, level=5, lib_path=None):
        super().__init__(device, level, lib_path)

    def compress_bound(self, n):
        return self._lib.gc_flzma2_compress_bound(n)

    def coder_props(self):
        return bytes([self._lib.gc_flzma2_dict_prop(self.level)])

    def code(self, data, flags=0):
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        cap = self.compress_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self._lib.gc_flzma2_compress_host(self._ctx, a.ctypes.data, a.size, out.ctypes.data, cap, self.level, flags, C.byref(n))
        self._check(rc, "gc_flzma2_compress_host")
        return out[:n.value]

    def code_device(self, d_src_ptr, n, d_dst_ptr, dst_cap, flags=0):
        self._check(self._lib.gc_flzma2_compress_device(self._ctx, d_src_ptr, n, d_dst_ptr, dst_cap, self.level, flags), "gc_flzma2_compress_device")

    def finish(self):
        n = C.c_size_t(0)
        self._check(self._lib.gc_flzma2_finish(self._ctx, C.byref(n)), "gc_flzma2_finish")
        return n.value

    def last_timing_ms(self):
        ms = (C.c_float * 7)()
        self._check(self._lib.gc_flzma2_last_timing(self._ctx, ms), "gc_flzma2_last_timing")
        return dict(zip(self.KERNELS, [float(x) for x in ms]))


class BrotliEncoder(_EncoderBase):
    CODEC = 2
    """Mirror of NCompress::NBROTLI::CEncoder (CPP/7zip/Compress/BrotliEncoder.h:35-70; Code() at BrotliEncoder.cpp:118-164):
    bytes -> brotli-mt framed chunks; `coder_props()` is the 3-byte blob {BROTLI_VERSION_MAJOR 1, BROTLI_VERSION_MINOR 2, level} of BrotliEncoder.h:18-32."""

    KERNELS = ("lz", "block", "plan", "emit", "total")

    def __init__(self, device=0, level=6, lib_path=None):
        super().__init__(device, level, lib_path)

    def compress_bound(self, n):
        return self._lib.gc_brotli_compress_bound(n)

    def coder_props(self):
        return bytes([1, 2, self.level])

    def code(self, data):
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        cap = self.compress_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self._lib.gc_brotli_compress_host(self._ctx, a.ctypes.data, a.size, out.ctypes.data, cap, self.level, C.byref(n))
        self._check(rc, "gc_brotli_compress_host")
        return out[:n.value]

    def code_device(self, d_src_ptr, n, d_dst_ptr, dst_cap):
        self._check(self._lib.gc_brotli_compress_device(self._ctx, d_src_ptr, n, d_dst_ptr, dst_cap, self.level), "gc_brotli_compress_device")

    def finish(self):
        n = C.c_size_t(0)
        self._check(self._lib.gc_brotli_finish(self._ctx, C.byref(n)), "gc_brotli_finish")
        return n.value

    def last_timing_ms(self):
        ms = (C.c_float * 5)()
        self._check(self._lib.gc_brotli_last_timing(self._ctx, ms), "gc_brotli_last_timing")
        return dict(zip(self.KERNELS, [float(x) for x in ms]))


class BrotliDecoder(_EncoderBase):
    """Mirror of NCompress::NBROTLI::CDecoder (CPP/7zip/Compress/BrotliDecoder.cpp:124) for whole brotli-mt streams: one wave per chunk on the GPU.  A stream that refers to
    the static dictionary of RFC 7932 needs set_dictionary() first (the library does not carry the 122 784 bytes; a host with a brotli of its own has them)."""

    def set_dictionary(self, data):
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        rc = self._lib.gc_brotli_dec_set_dictionary(a.ctypes.data if a.size else None, a.size)
        if rc != GC_OK:
            raise GpuCodecError("gc_brotli_dec_set_dictionary: %s (not the dictionary of RFC 7932 Appendix A)" % _ERR.get(rc, rc))

    def has_dictionary(self):
        return bool(self._lib.gc_brotli_dec_has_dictionary())

    def scan(self, data):
        """-> (array of BrotliChunk, their number, the sum of their capacities, bytes of whole frames)"""
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        n = C.c_size_t(0); used = C.c_size_t(0); cap = C.c_uint64(0)
        rc = self._lib.gc_brotli_scan_prefix(a.ctypes.data, a.size, None, 0, C.byref(n), None, None)
        if rc != GC_OK:
            raise GpuCodecError("gc_brotli_scan_prefix failed: %s" % _ERR.get(rc, rc))
        chunks = (BrotliChunk * max(1, n.value))()
        rc = self._lib.gc_brotli_scan_prefix(a.ctypes.data, a.size, chunks, n.value, C.byref(n), C.byref(cap), C.byref(used))
        if rc != GC_OK:
            raise GpuCodecError("gc_brotli_scan_prefix failed: %s" % _ERR.get(rc, rc))
        return chunks, n.value, cap.value, used.value

    def code(self, data, capacity=None):
        """brotli-mt stream (or a bare RFC 7932 stream, then capacity is needed) -> numpy uint8 content (host buffers; includes PCIe copies)"""
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        if capacity is None:
            framed = a.size >= 4 and int.from_bytes(a[:4].tobytes(), "little") == 0x184D2A50
            capacity = self.scan(a)[2] if framed else max(1 << 20, 64 * a.size)
        out = np.empty(max(1, capacity), dtype=np.uint8)
        n = C.c_size_t(0)
        self._check(self._lib.gc_brotli_decompress_host(self._ctx, a.ctypes.data, a.size, out.ctypes.data, capacity, C.byref(n)), "gc_brotli_decompress_host")
        return out[:n.value]

    def code_device(self, d_src_ptr, n, d_dst_ptr, dst_cap, chunks, n_chunks):
        size = C.c_size_t(0)
        self._check(self._lib.gc_brotli_decompress_device(self._ctx, d_src_ptr, n, d_dst_ptr, dst_cap, chunks, n_chunks, C.byref(size)), "gc_brotli_decompress_device")
        return size.value

    def last_timing_ms(self):
        ms = C.c_float(0)
        self._check(self._lib.gc_brotli_decompress_timing(self._ctx, C.byref(ms)), "gc_brotli_decompress_timing")
        return float(ms.value)


class ZstdDecoder(_EncoderBase):
    """Mirror of NCompress::NZSTD::CDecoder (CPP/7zip/Compress/ZstdDecoder.cpp) for whole streams: frames decode concurrently on the GPU."""
    UNKNOWN = (1 << 64) - 1

    def scan(self, data):
        """-> (array of ZstdFrame, total content size or None if a frame does not state its size)"""
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self._lib.gc_zstd_scan_frames(a.ctypes.data, a.size, None, 0, C.byref(n), None)
        if rc != GC_OK:
            raise GpuCodecError("gc_zstd_scan_frames failed: %s" % _ERR.get(rc, rc))
        frames = (ZstdFrame * max(1, n.value))()
        total = C.c_uint64(0)
        rc = self._lib.gc_zstd_scan_frames(a.ctypes.data, a.size, frames, n.value, C.byref(n), C.byref(total))
        if rc != GC_OK:
            raise GpuCodecError("gc_zstd_scan_frames failed: %s" % _ERR.get(rc, rc))
        return frames, n.value, (None if total.value == self.UNKNOWN else total.value)

    def code(self, data, capacity=None):
        """compressed bytes-like -> numpy uint8 content (host buffers; includes PCIe copies).  capacity is only needed for
        frames that do not state their content size."""
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        _, _, total = self.scan(a)
        cap = total if total is not None else (capacity if capacity is not None else max(1 << 20, 64 * a.size))
        out = np.empty(max(1, cap), dtype=np.uint8)
        n = C.c_size_t(0)
        self._check(self._lib.gc_zstd_decompress_host(self._ctx, a.ctypes.data, a.size, out.ctypes.data, cap, C.byref(n)), "gc_zstd_decompress_host")
        return out[:n.value]

    def code_device(self, d_src_ptr, n, d_dst_ptr, dst_cap, frames, n_frames):
        size = C.c_size_t(0)
        self._check(self._lib.gc_zstd_decompress_device(self._ctx, d_src_ptr, n, d_dst_ptr, dst_cap, frames, n_frames, C.byref(size)), "gc_zstd_decompress_device")
        return size.value

    def last_timing_ms(self):
        ms = C.c_float(0)
        self._check(self._lib.gc_zstd_decompress_timing(self._ctx, C.byref(ms)), "gc_zstd_decompress_timing")
        return float(ms.value)

    KERNELS = ("index", "literals", "sequences", "execution")

    def kernel_timing_ms(self):
        ms = (C.c_float * 4)()
        self._lib.gc_zstd_decompress_kernel_timing.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        self._check(self._lib.gc_zstd_decompress_kernel_timing(self._ctx, ms), "gc_zstd_decompress_kernel_timing")
        return dict(zip(self.KERNELS, [float(v) for v in ms]))

    def selfcheck(self):
        """1 if this context's six-blocks-per-wave sequences kernel decoded the built-in frame of the reference's encoder correctly, -1 if not (the
        context then uses the one-block-per-wave kernel).  Runs the check if it has not run yet."""
        st = C.c_int(0)
        self._lib.gc_zstd_decompress_selfcheck.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._check(self._lib.gc_zstd_decompress_selfcheck(self._ctx, C.byref(st)), "gc_zstd_decompress_selfcheck")
        return int(st.value)

    def wide_rounds(self):
        """Pointer-jumping rounds of the last call if it took the wide execution path (0: the frame-per-workgroup kernel ran)."""
        r = C.c_uint(0)
        self._lib.gc_zstd_decompress_wide_rounds.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
        self._check(self._lib.gc_zstd_decompress_wide_rounds(self._ctx, C.byref(r)), "gc_zstd_decompress_wide_rounds")
        return int(r.value)


class ZstdEncoder(_EncoderBase):
    CODEC = 0
    """Mirror of NCompress::NZSTD::CEncoder for the compression hot path (one object per GPU)."""

    KERNELS = ("lz", "huf", "seq", "plan", "emit", "total")

    def compress_bound(self, n):
        return self._lib.gc_zstd_compress_bound(n)

    def code(self, data):
        """bytes-like / numpy uint8 -> compressed bytes (host buffers; includes PCIe copies)."""
        import numpy as np
        a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        cap = self.compress_bound(a.size)
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        rc = self._lib.gc_zstd_compress_host(self._ctx, a.ctypes.data, a.size, out.ctypes.data, cap, self.level, C.byref(n))
        self._check(rc, "gc_zstd_compress_host")
        return out[:n.value]

    def code_device(self, d_src_ptr, n, d_dst_ptr, dst_cap):
        """Enqueue compression of device memory (raw pointers, e.g. torch.Tensor.data_ptr())."""
        rc = self._lib.gc_zstd_compress_device(self._ctx, d_src_ptr, n, d_dst_ptr, dst_cap, self.level)
        self._check(rc, "gc_zstd_compress_device")

    def finish(self):
        n = C.c_size_t(0)
        self._check(self._lib.gc_zstd_finish(self._ctx, C.byref(n)), "gc_zstd_finish")
        return n.value

    def last_timing_ms(self):
        ms = (C.c_float * 6)()
        self._check(self._lib.gc_zstd_last_timing(self._ctx, ms), "gc_zstd_last_timing")
        return dict(zip(self.KERNELS, [float(x) for x in ms]))

    PHASES = ("lz.probe", "lz.insert", "lz.verify", "lz.double", "lz.chain", "lz.walk", "lz.emit",
              "seq.merge", "seq.codes", "seq.tables", "seq.chains", "seq.pack",
              "seq.chains.stage", "seq.chains.warm", "seq.chains.walk", "seq.chains.out")

    def set_phase_profile(self, on=True):
        self._check(self._lib.gc_zstd_set_phase_profile(self._ctx, 1 if on else 0), "gc_zstd_set_phase_profile")

    def phase_profile(self):
        """Average shader cycles per block and phase of the last call (thread 0's view, barrier waits included)."""
        v = (C.c_double * 16)()
        self._check(self._lib.gc_zstd_phase_profile(self._ctx, v), "gc_zstd_phase_profile")
        return dict(zip(self.PHASES, [float(x) for x in v]))
