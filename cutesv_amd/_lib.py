"""Loader of libcutesv_hip.so (the HIP kernels + C ABI).  There is no CPU fallback: if the
extension is missing or does not load, importing this module's `lib()` raises."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# CUTESV_AMD_LIB: load another build of the same library (A/B timing of kernel variants); never a CPU fallback
LIB_PATH = os.environ.get("CUTESV_AMD_LIB") or os.path.join(_HERE, "libcutesv_hip.so")
_LIB = None

# every symbol include/cutesv_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("csv_abi_version", C.c_int, []),
    ("csv_struct_size", C.c_int, [C.c_int]),
    ("csv_measure_copy_bandwidth", C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_double)]),
    ("csv_cache_flush", C.c_int, [C.c_void_p, C.c_int64]),
    ("csv_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("csv_device_info", C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]),
    ("csv_ctx_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("csv_ctx_destroy", None, [C.c_void_p]),
    ("csv_last_error", C.c_char_p, [C.c_void_p]),
    ("csv_stage_name", C.c_char_p, [C.c_int]),
    ("csv_cluster_batch", C.c_int, [C.c_void_p, C.POINTER(_abi.BatchIn), C.POINTER(_abi.BatchOut)]),
    ("csv_batch_upload", C.c_int, [C.c_void_p, C.POINTER(_abi.BatchIn)]),
    ("csv_batch_run", C.c_int, [C.c_void_p, C.POINTER(_abi.RunStats)]),
    ("csv_batch_download", C.c_int, [C.c_void_p, C.POINTER(_abi.BatchOut)]),
    ("csv_ctx_sync", C.c_int, [C.c_void_p]),
    ("csv_batch_publish_async", C.c_int, [C.c_void_p, C.POINTER(_abi.BatchOut)]),
    ("csv_batch_publish_wait", C.c_int, [C.c_void_p, C.POINTER(C.POINTER(_abi.BatchOut))]),
    ("csv_batch_reads_mode", C.c_int, [C.c_void_p]),
    ("csv_batch_validate", C.c_int, [C.c_void_p]),
    ("csv_batch_info", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    ("csv_batch_option", C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    ("csv_gl_index", C.c_int32, [C.c_int64, C.c_int64]),
    ("csv_host_alloc", C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    ("csv_host_free", None, [C.c_void_p]),
    ("csv_host_register", C.c_int, [C.c_void_p, C.c_int64]),
    ("csv_host_unregister", C.c_int, [C.c_void_p]),
    ("csv_rows_emit", C.c_int, None),          # prototype set in cutesv_amd/rows.py (needs its struct)
    ("csv_cigar_signatures", C.c_int, None),   # prototype set in cutesv_amd/extract.py
    ("csv_split_signatures", C.c_int, None),   # prototype set in cutesv_amd/extract.py
    ("csv_rebuild_signatures", C.c_int, None),  # prototype set in cutesv_amd/rebuild.py
    ("csv_pool_reset", C.c_int, [C.c_void_p]),
    ("csv_pool_rows", C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    ("csv_pool_append", C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csv_vcf_emit", C.c_int, None),           # prototype set in cutesv_amd/vcf.py (needs its struct)
    ("csv_fasta_index", C.c_int64, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
]


class ExtensionMissing(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ExtensionMissing(
                "%s not found: build it with `make -C cutesv_amd/csrc` (or __graft_entry__.build()). "
                "cutesv_amd has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)          # AttributeError here == ABI mismatch, fail loudly
            fn.restype = res
            if args is not None:
                fn.argtypes = args
        if L.csv_abi_version() != _abi.ABI_VERSION:
            raise ExtensionMissing("libcutesv_hip.so ABI %d != python side %d" % (L.csv_abi_version(), _abi.ABI_VERSION))
        _LIB = L
    return _LIB
