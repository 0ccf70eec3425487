"""ctypes driver of oracle/liboracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from cutesv_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.csvo_cluster_batch.restype = C.c_int
        _LIB.csvo_cluster_batch.argtypes = [C.POINTER(_abi.BatchIn), C.POINTER(_abi.BatchOut)]
        _LIB.csvo_gl_index.restype = C.c_int32
        _LIB.csvo_gl_index.argtypes = [C.c_int64, C.c_int64]
        _LIB.csvo_np_sum_f64.restype = C.c_double
        _LIB.csvo_np_sum_f64.argtypes = [C.c_void_p, C.c_int64]
        _LIB.csvo_np_std_i64.restype = C.c_double
        _LIB.csvo_np_std_i64.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        _LIB.csvo_cipos.restype = C.c_int32
        _LIB.csvo_cipos.argtypes = [C.c_double, C.c_int64]
        _LIB.csvo_cover_count.restype = C.c_int
        _LIB.csvo_cover_count.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    return _LIB


def cluster_batch(batch, per_sig=True, cap_calls=None, cap_support=None):
    """Run the C restatement on a HostBatch; returns a HostResult (retries once on capacity)."""
    batch = batch.widened()                    # (the oracle reads int64 columns only)
    n = batch.n_sig
    cap_calls = cap_calls or max(64, n // 8 + 16)
    cap_support = cap_support or max(64, n + 16)
    for _ in range(2):
        res = _abi.HostResult(n, cap_calls, cap_support, per_sig=per_sig, n_seg=len(batch.segments))
        rc = lib().csvo_cluster_batch(C.byref(batch.c), C.byref(res.c))
        if rc == _abi.E_CAPACITY:
            cap_calls, cap_support = res.n_calls + 1, res.n_support + 1
            continue
        if rc != _abi.OK:
            raise RuntimeError("oracle: %s" % _abi.ERR_NAME.get(rc, rc))
        return res
    raise RuntimeError("oracle: capacity retry failed")


def cluster_tasks_mt(store, tasks, params, threads):
    """The C restatement over the (chr, type) tasks of a workload, one task per call like the reference's pool
    (main script :1116-1189), on `threads` threads (ctypes drops the GIL inside the C call); largest tasks first.
    Returns (wall seconds of the parallel region, total calls).  Batches and result arrays are built outside the clock."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    lib()
    jobs = []
    for t in tasks:
        hb = store.host_batch([t], params)
        if hb.r_start is not None and t[0] != "TRA":
            # a task reads its own chromosome's block of the reads table and nothing else (INDEL:52-58)
            ci = store.chroms.index(t[1])
            lo, hi = int(store.reads_off[ci]), int(store.reads_off[ci + 1])
            off = np.zeros(len(store.chroms) + 1, np.int64)
            off[ci + 1:] = hi - lo
            hb = _abi.HostBatch(hb.segments, hb.a, hb.b, hb.read_id, hb.aux, n_chrom=len(store.chroms), reads_off=off,
                                r_start=hb.r_start[lo:hi], r_end=hb.r_end[lo:hi], r_primary=hb.r_primary[lo:hi],
                                r_id=hb.r_id[lo:hi])
        hb = hb.widened()
        n = hb.n_sig
        res = _abi.HostResult(n, max(64, n // 4 + 16), max(64, n + 16), per_sig=False, n_seg=1)
        jobs.append((hb, res))
    jobs.sort(key=lambda j: -j[0].n_sig)

    def run(j):
        rc = _LIB.csvo_cluster_batch(C.byref(j[0].c), C.byref(j[1].c))
        if rc != _abi.OK:
            raise RuntimeError("oracle: %s" % _abi.ERR_NAME.get(rc, rc))
        return j[1].n_calls
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda i: time.sleep(0.002), range(threads)))       # (the threads exist before the clock starts)
        t0 = time.perf_counter()
        calls = sum(ex.map(run, jobs))
        dt = time.perf_counter() - t0
    return dt, calls


def np_sum(x):
    x = np.ascontiguousarray(x, np.float64)
    return lib().csvo_np_sum_f64(x.ctypes.data, len(x))


def np_std(v):
    v = np.ascontiguousarray(v, np.int64)
    s = np.empty(len(v), np.float64)
    return lib().csvo_np_std_i64(v.ctypes.data, len(v), s.ctypes.data)


def cover_count(r_start, r_end, r_primary, r_id, L2, R2):
    r_start = np.ascontiguousarray(r_start, np.int64); r_end = np.ascontiguousarray(r_end, np.int64)
    r_primary = np.ascontiguousarray(r_primary, np.uint8); r_id = np.ascontiguousarray(r_id, np.int32)
    L2 = np.ascontiguousarray(L2, np.int64); R2 = np.ascontiguousarray(R2, np.int64)
    out = np.zeros(len(L2), np.int32)
    rc = lib().csvo_cover_count(r_start.ctypes.data, r_end.ctypes.data, r_primary.ctypes.data, r_id.ctypes.data,
                                len(r_start), L2.ctypes.data, R2.ctypes.data, len(L2), out.ctypes.data)
    assert rc == 0
    return out


def cigar_signatures(cig_off, cigar, ref_start, use=None, min_siglength=10, merge_ins_threshold=100, merge_del_threshold=0):
    """the C restatement of the CIGAR scan (csvo_cigar_signatures); same result dict as cutesv_amd.extract.cigar_signatures"""
    from cutesv_amd import extract
    L = lib()
    L.csvo_cigar_signatures.restype = C.c_int
    L.csvo_cigar_signatures.argtypes = [C.POINTER(extract.CigarIn), C.POINTER(extract.CigarOut)]

    def check(rc):
        if rc != _abi.OK:
            raise RuntimeError("oracle: %s" % _abi.ERR_NAME.get(rc, rc))
    return extract._run(L.csvo_cigar_signatures, None, cig_off, cigar, ref_start, use, min_siglength, merge_ins_threshold, merge_del_threshold, check)


def split_signatures(enc, sv_size=30, min_mapq=20, max_split_parts=7, max_size=100000):
    """the C restatement of the split-read analysis (csvo_split_signatures); same result dict as cutesv_amd.extract.split_signatures"""
    from cutesv_amd import extract
    L = lib()
    L.csvo_split_signatures.restype = C.c_int
    L.csvo_split_signatures.argtypes = [C.POINTER(extract.SplitIn), C.POINTER(extract.SplitOut)]

    def check(rc):
        if rc != _abi.OK:
            raise RuntimeError("oracle: %s" % _abi.ERR_NAME.get(rc, rc))
    return extract._run_split(L.csvo_split_signatures, None, enc, sv_size, min_mapq, max_split_parts, max_size, check)
