"""ctypes binding of include/fqtk_match.h.  Fails loudly when the HIP library is missing."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libfqtk_match.so")

FQTK_OK, FQTK_EINVAL, FQTK_ELEN, FQTK_EHIP, FQTK_ENOMEM, FQTK_ENODEV, FQTK_ENCCL = 0, 1, 2, 3, 4, 5, 6
FQTK_NO_MATCH = 0xFFFF
FQTK_MAX_SLOTS = 8


class fqtk_match_t(C.Structure):
    _fields_ = [("idx", C.c_uint16), ("best", C.c_uint8), ("next", C.c_uint8)]


class fqtk_bgzf_block(C.Structure):   # include/fqtk_bgzf.h
    _fields_ = [("in_", C.c_void_p), ("out", C.c_void_p), ("n_in", C.c_uint32), ("reserved", C.c_uint32)]


FQTK_BGZF_MAX_IN, FQTK_BGZF_OUT_STRIDE, FQTK_BGZF_SLOTS = 65280, 65536, 4


class fqtk_inflate_member(C.Structure):   # include/fqtk_inflate.h
    _fields_ = [("payload_off", C.c_uint64), ("out_off", C.c_uint64), ("payload_len", C.c_uint32), ("isize", C.c_uint32),
                ("crc", C.c_uint32), ("reserved", C.c_uint32)]


FQTK_INFLATE_SLOTS, FQTK_INFLATE_MAX_ISIZE, FQTK_INFLATE_ERR_CRC = 4, 65536, 10


class fqtk_stream_chunk(C.Structure):   # include/fqtk_demux.h
    _fields_ = [("start_bit", C.c_uint64), ("stop_bit", C.c_uint64)]


class fqtk_stream_end(C.Structure):
    _fields_ = [("status", C.c_uint32), ("final_block", C.c_uint32), ("n_bytes", C.c_uint64), ("end_bit", C.c_uint64), ("start_bit", C.c_uint64),
                ("n_blocks", C.c_uint32), ("flags", C.c_uint32)]


# include/fqtk_demux.h
class fqtk_fed_window(C.Structure):   # a cut of an input's fed text (fqtk_demuxer_fed_cut -> fqtk_demuxer_submit_windows)
    _fields_ = [("home", C.c_void_p), ("input", C.c_uint32), ("lead", C.c_uint32), ("first_line", C.c_uint32), ("n_templates", C.c_uint32),
                ("base", C.c_void_p), ("len", C.c_uint64), ("pos", C.c_uint64), ("arena", C.c_uint32), ("reserved", C.c_uint32)]


class fqtk_demux_segment(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("length", C.c_int32), ("kind", C.c_char)]


class fqtk_demux_config(C.Structure):
    _fields_ = [("n_inputs", C.c_uint32), ("n_segments", C.POINTER(C.c_uint32)),
                ("segments", C.POINTER(fqtk_demux_segment)), ("want", C.c_uint8 * 4),
                ("skip_too_few_bases", C.c_int), ("max_chunk_templates", C.c_uint32), ("carry_blocks", C.c_int),
                ("compression_level", C.c_int)]


class fqtk_demux_result(C.Structure):
    _fields_ = [("bytes", C.c_void_p), ("file_off", C.POINTER(C.c_uint64)), ("n_files", C.c_uint64),
                ("n_blocks", C.c_uint64), ("n_templates", C.c_uint32), ("n_skipped", C.c_uint32), ("error", C.c_int),
                ("error_input", C.c_uint32), ("error_template", C.c_uint32), ("error_detail", C.c_uint32),
                ("text_end", C.POINTER(C.c_uint64))]


FQTK_DEMUX_SLOTS, FQTK_DEMUX_STAGES = 3, 8


_lib = None
_hip_preloaded = False


def preload_hip_runtime() -> None:
    """One HIP runtime per process.  libfqtk_match.so needs `libamdhip64.so.7`; PyTorch-ROCm wheels
    bundle their own copy with that soname but link it as plain `libamdhip64.so`, so whichever side
    loads second would otherwise pull in a SECOND runtime (which then sees no GPU).  If a PyTorch-ROCm
    install is present, load its copy first (RTLD_GLOBAL) so both sides share it; otherwise the
    system ROCm runtime is found through the library's RUNPATH.  torch itself is NOT imported."""
    global _hip_preloaded
    if _hip_preloaded:
        return
    _hip_preloaded = True
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass

# (name, restype, argtypes) for EVERY symbol include/fqtk_match.h declares
SIGNATURES = [
    ("fqtk_last_error", C.c_char_p, []),
    ("fqtk_abi_version", C.c_int, []),
    ("fqtk_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("fqtk_matcher_create", C.c_int, [C.POINTER(C.c_char_p), C.c_uint32, C.c_uint32, C.c_uint8,
                                      C.c_uint8, C.c_int, C.POINTER(C.c_void_p)]),
    ("fqtk_matcher_destroy", None, [C.c_void_p]),
    ("fqtk_matcher_set_sample_ids", C.c_int, [C.c_void_p, C.POINTER(C.c_char_p)]),
    ("fqtk_matcher_n_samples", C.c_uint32, [C.c_void_p]),
    ("fqtk_matcher_barcode_len", C.c_uint32, [C.c_void_p]),
    ("fqtk_matcher_max_ns_in_barcodes", C.c_uint32, [C.c_void_p]),
    ("fqtk_matcher_device", C.c_int, [C.c_void_p]),
    ("fqtk_matcher_set_use_cache", C.c_int, [C.c_void_p, C.c_int]),
    ("fqtk_matcher_memo_entries", C.c_uint64, [C.c_void_p]),
    ("fqtk_matcher_memo_candidates", C.c_uint64, [C.c_void_p]),
    ("fqtk_matcher_memo_kind", C.c_int, [C.c_void_p]),
    ("fqtk_matcher_set_memo_kind", C.c_int, [C.c_void_p, C.c_int]),
    ("fqtk_matcher_memo_direct_bytes", C.c_int, [C.c_void_p]),
    ("fqtk_matcher_assign_batch", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                            C.c_uint64, C.c_void_p, C.c_void_p]),
    ("fqtk_matcher_assign_batch_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p,
                                                   C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("fqtk_matcher_poll_error", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    ("fqtk_matcher_assign1", C.c_int, [C.c_void_p, C.c_char_p, C.c_uint32, C.POINTER(fqtk_match_t)]),
    ("fqtk_pinned_alloc", C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    ("fqtk_pinned_free", C.c_int, [C.c_void_p]),
    ("fqtk_matcher_enqueue", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p,
                                       C.c_uint64, C.c_void_p]),
    ("fqtk_packed_stride", C.c_uint32, [C.c_uint32]),
    ("fqtk_pack_barcodes", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("fqtk_matcher_enqueue_packed", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p,
                                              C.c_void_p, C.c_uint64, C.c_void_p]),
    ("fqtk_matcher_wait", C.c_int, [C.c_void_p, C.c_int]),
    ("fqtk_matcher_counts", C.c_int, [C.c_void_p, C.c_void_p]),
    ("fqtk_matchers_allreduce_counts", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    # include/fqtk_bgzf.h
    ("fqtk_bgzf_last_error", C.c_char_p, []),
    ("fqtk_bgzf_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("fqtk_bgzf_destroy", None, [C.c_void_p]),
    ("fqtk_bgzf_deflate_enqueue", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("fqtk_bgzf_wait", C.c_int, [C.c_void_p, C.c_int]),
    # include/fqtk_inflate.h
    ("fqtk_inflate_last_error", C.c_char_p, []),
    ("fqtk_inflate_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("fqtk_inflate_destroy", None, [C.c_void_p]),
    ("fqtk_inflate_enqueue", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("fqtk_inflate_wait", C.c_int, [C.c_void_p, C.c_int]),
    # include/fqtk_demux.h
    ("fqtk_demuxer_create", C.c_int, [C.c_void_p, C.POINTER(fqtk_demux_config), C.POINTER(C.c_void_p)]),
    ("fqtk_demuxer_destroy", None, [C.c_void_p]),
    ("fqtk_demuxer_files_per_sample", C.c_uint32, [C.c_void_p]),
    ("fqtk_demuxer_submit", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_uint32]),
    ("fqtk_demuxer_collect", C.c_int, [C.c_void_p, C.c_int, C.POINTER(fqtk_demux_result)]),
    ("fqtk_demuxer_collect_begin", C.c_int, [C.c_void_p, C.c_int]),
    ("fqtk_demuxer_text_done", C.c_int, [C.c_void_p, C.c_int]),
    ("fqtk_demuxer_record_text", C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    ("fqtk_demuxer_flush", C.c_int, [C.c_void_p, C.POINTER(fqtk_demux_result)]),
    ("fqtk_demuxer_counts", C.c_int, [C.c_void_p, C.c_void_p]),
    ("fqtk_demuxer_stage_seconds", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("fqtk_demuxer_stage_name", C.c_char_p, [C.c_int]),
    ("fqtk_demuxer_feed", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]),
    ("fqtk_demuxer_submit_fed", C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    ("fqtk_demuxer_fed_cut", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(fqtk_fed_window)]),
    ("fqtk_demuxer_submit_windows", C.c_int, [C.c_void_p, C.c_int, C.POINTER(fqtk_fed_window), C.c_uint32]),
    ("fqtk_demuxer_fed_tail", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    ("fqtk_demuxer_inflate_seconds", C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    ("fqtk_demuxer_stream_decode", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("fqtk_demuxer_stream_scan", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]),
    ("fqtk_demuxer_stream_reserve", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]),
    ("fqtk_demuxer_stream_window", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    ("fqtk_demuxer_stream_commit_text", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    ("fqtk_demuxer_stream_commit", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
]


def load() -> C.CDLL:
    """Loads libfqtk_match.so (the hand-written HIP product library).  No fallback of any kind."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m fqtk_amd.build` (hipcc, gfx950). "
            "fqtk_amd has no CPU fallback.")
    preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().fqtk_last_error().decode(errors="replace")
