"""ctypes binding of libgoleft_depth.so (include/goleft_depth.h).

Fails loudly when the HIP library is missing: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libgoleft_depth.so")


class GdParams(C.Structure):
    _fields_ = [("window_size", C.c_int32), ("min_mapq", C.c_int32), ("min_cov", C.c_int32),
                ("max_mean_depth", C.c_int32), ("flag_mask", C.c_uint32),
                ("max_span_hint", C.c_int32), ("step", C.c_int64)]


class GdBatch(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p),
                ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
                ("reads_cap", C.c_size_t), ("ops_cap", C.c_size_t), ("slot", C.c_int32)]


class GdRun(C.Structure):
    _fields_ = [("start", C.c_int32), ("end", C.c_int32), ("cls", C.c_int32)]


class GdStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_ops", C.c_uint64), ("n_ref_bases", C.c_uint64),
                ("n_windows", C.c_uint64), ("n_tiles", C.c_uint64), ("n_runs", C.c_uint64),
                ("tile_positions", C.c_int32), ("lookback", C.c_int32),
                ("max_span_seen", C.c_int32), ("reruns", C.c_int32),
                ("path", C.c_int32), ("n_slow_tiles", C.c_int32), ("n_canonical_ops", C.c_uint64),
                ("tile_kernel", C.c_int32), ("reserved_", C.c_int32), ("n_deletions", C.c_uint64)]


# every symbol include/goleft_depth.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "gd_strerror": (C.c_char_p, [C.c_int]),
    "gd_abi_version": (C.c_int, []),
    "gd_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gd_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gd_destroy": (None, [_P]),
    "gd_last_error": (C.c_char_p, [_P]),
    "gd_set_stream": (C.c_int, [_P, _P]),
    "gd_set_params": (C.c_int, [_P, C.POINTER(GdParams)]),
    "gd_default_params": (C.c_int, [C.POINTER(GdParams)]),
    "gd_set_path": (C.c_int, [_P, C.c_int]),
    "gd_set_outputs": (C.c_int, [_P, C.c_uint]),
    "gd_set_contigs": (C.c_int, [_P, C.c_int, _P]),
    "gd_select_contigs": (C.c_int, [_P, C.c_int, _P]),
    "gd_acquire": (C.c_int, [_P, C.c_size_t, C.c_size_t, C.POINTER(GdBatch)]),
    "gd_check_commits": (C.c_int, [_P]),
    "gd_get_option": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    "gd_commit": (C.c_int, [_P, C.POINTER(GdBatch), C.c_int32, C.c_size_t, C.c_size_t]),
    "gd_reserve": (C.c_int, [_P, C.c_int32, C.c_size_t, C.c_size_t]),
    "gd_push": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, C.c_size_t, C.c_size_t]),
    "gd_adopt_device": (C.c_int, [_P, C.c_int32, C.POINTER(GdBatch), C.c_size_t, C.c_size_t]),
    "gd_reset": (C.c_int, [_P]),
    "gd_compute": (C.c_int, [_P]),
    "gd_compute_launch": (C.c_int, [_P]),
    "gd_compute_finish": (C.c_int, [_P]),
    "gd_perbase": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, _P]),
    "gd_windows": (C.c_int, [_P, C.c_int32, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gd_callable": (C.c_int, [_P, C.c_int32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gd_region_windows": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, _P, _P, C.c_size_t,
                                    C.POINTER(C.c_size_t)]),
    "gd_region_callable": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int64, _P, C.c_size_t,
                                     C.POINTER(C.c_size_t)]),
    "gd_regions": (C.c_int, [_P, C.c_size_t, _P, _P, _P, _P, _P, C.c_size_t, _P, _P, C.c_size_t, _P]),
    "gd_depthwed": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64, _P, _P, _P, _P, C.c_size_t,
                              C.POINTER(C.c_size_t)]),
    "gd_depthwed_device": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int64, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "gd_seq_load": (C.c_int, [_P, _P, C.c_int64]),
    "gd_seq_stats": (C.c_int, [_P, C.c_size_t, _P, _P, _P, _P, _P]),
    "gd_seq_stats_ex": (C.c_int, [_P, C.c_size_t, _P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "gd_md_flags": (C.c_int, [_P, C.c_int, _P, C.c_int32, C.c_int32, _P, _P, C.c_size_t]),
    "gd_md_begin": (C.c_int, [_P, C.c_int64]),
    "gd_md_accumulate": (C.c_int, [_P, C.c_int, _P, C.c_int32]),
    "gd_md_finish": (C.c_int, [_P, C.c_int32, _P, _P, C.c_size_t]),
    "gd_md_load_flags": (C.c_int, [_P, _P, _P, C.c_int64]),
    "gd_md_blocks": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gd_md_sums_group": (C.c_int, [_P, C.c_int, _P, C.c_size_t, _P, _P, _P]),
    "gd_md_sums": (C.c_int, [_P, C.c_size_t, _P, _P, _P]),
    "gd_inflate_bgzf": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "gd_ingest_bgzf": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_size_t, C.c_uint64, _P, C.c_size_t,
                                 C.POINTER(C.c_uint64)]),
    "gd_bgzf_members": (C.c_int, [_P, C.c_size_t, C.c_size_t, _P, _P, _P, _P, _P, C.POINTER(C.c_size_t)]),
    "gd_ingest_begin": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_size_t, _P, _P, _P, _P, _P]),
    "gd_ingest_feed": (C.c_int, [_P, _P, C.c_size_t]),
    "gd_ingest_feed_fd": (C.c_int, [_P, C.c_int, C.c_uint64, C.c_size_t]),
    "gd_ingest_timing": (C.c_int, [_P, _P, C.c_size_t]),
    "gd_ingest_finish": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_size_t, C.POINTER(C.c_uint64)]),
    "gd_ingest_decode": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_size_t, C.POINTER(C.c_uint64)]),
    "gd_ingest_decode_part": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_size_t, C.c_uint64, C.c_uint, C.c_double,
                              C.POINTER(C.c_uint64)]),
    "gd_ingest_release": (C.c_int, [_P]),
    "gd_ingest_abort": (C.c_int, [_P]),
    "gd_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "gd_host_free": (C.c_int, [_P, _P]),
    "gd_device_perbase": (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "gd_device_windows": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "gd_window_offset": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "gd_device_runs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "gd_set_option": (C.c_int, [_P, C.c_int, C.c_int64]),
    "gd_drop_derived": (C.c_int, [_P]),
    "gd_rebuild_derived": (C.c_int, [_P]),
    "gd_set_export": (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    "gd_wait_event": (C.c_int, [_P, _P]),
    "gd_get_stats": (C.c_int, [_P, C.POINTER(GdStats)]),
    "gd_set_profiling": (C.c_int, [_P, C.c_int]),
    "gd_kernel_ms": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    "gd_comm_unique_id": (C.c_int, [_P, C.c_size_t]),
    "gd_comm_library": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "gd_comm_init": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t]),
    "gd_comm_destroy": (C.c_int, [_P]),
    "gd_gather_export": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int]),
    "gd_gather_wait": (C.c_int, [_P]),
    "gd_device_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "gd_device_free": (C.c_int, [_P, _P]),
    "gd_device_read": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "gd_compute_timing": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int]),
}

_LIB = None


def build(force: bool = False) -> str:
    """Compile everything in-tree for gfx950 (hipcc cross-compiles without a GPU):
    libgoleft_depth.so (HIP engine), libgoleft_host.so (C++ host), goleft-depth (CLI)."""
    src_dir = os.path.join(HERE, "csrc")
    subprocess.check_call(["make", "-s", "-C", src_dir, "all"] + (["-B"] if force else []))
    return SO_PATH


def load():
    """Load the library; raises (never falls back) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.environ.get("GOLEFT_DEPTH_SO") or SO_PATH      # (measurement: a kernel variant built beside the product library)
    if not os.path.exists(so):
        raise ImportError(
            "goleft_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % so)
    lib = C.CDLL(so)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib
