"""Thin Python face of the C ABI: one `Context` per process and GPU (created after fork, like the
reference's pool workers), batches in, `HostResult`s out."""
import ctypes as C

from . import _abi
from ._lib import lib


class CsvError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("%s: %s" % (_abi.ERR_NAME.get(code, code), text))
        self.code = code


def device_count():
    n = C.c_int(0)
    lib().csv_device_count(C.byref(n))
    return n.value


def device_info(device=0):
    """(PCI bus id, compute units) of a device: csv_device_info"""
    buf = C.create_string_buffer(64)
    ncu = C.c_int(0)
    rc = lib().csv_device_info(int(device), buf, 64, C.byref(ncu))
    if rc != _abi.OK:
        raise CsvError(rc, "csv_device_info(%d) failed" % device)
    return buf.value.decode(), ncu.value


class Context:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().csv_ctx_create(int(device), C.byref(self._h))
        if rc != _abi.OK:
            raise CsvError(rc, "csv_ctx_create(device=%d) failed (is a GPU visible?)" % device)
        self.device = device
        self._batch = None
        self._res_cache = None

    def close(self):
        if self._h:
            lib().csv_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != _abi.OK:
            raise CsvError(rc, (lib().csv_last_error(self._h) or b"").decode())

    # ---- resident mode
    def upload(self, batch, per_sig=None):
        """per_sig: also produce cluster_id / allele_id (CSV_IN_PER_SIG); None keeps the batch's own flag"""
        if per_sig is not None:
            batch.c.flags = (batch.c.flags & ~_abi.IN_PER_SIG) | (_abi.IN_PER_SIG if per_sig else 0)
        self._check(lib().csv_batch_upload(self._h, C.byref(batch.c)))
        self._batch = batch
        self._inflight = []                       # (an upload drains the deliveries nobody waited for)

    def run(self, stats=False):
        if stats:
            st = _abi.RunStats()
            self._check(lib().csv_batch_run(self._h, C.byref(st)))
            return st
        self._check(lib().csv_batch_run(self._h, None))
        return None

    def validate(self):
        """check the uploaded batch against the reference's rebuild order (raises CsvError E_UNSORTED)"""
        self._check(lib().csv_batch_validate(self._h))

    def sync(self):
        self._check(lib().csv_ctx_sync(self._h))

    def copy_bandwidth(self, nbytes=512 << 20, reps=10):
        """device-to-device copy ceiling of this GPU in GB/s (read + write bytes); a measurement aid for bench.py"""
        out = C.c_double(0.0)
        self._check(lib().csv_measure_copy_bandwidth(self._h, int(nbytes), int(reps), C.byref(out)))
        return float(out.value)

    def option(self, option, value):
        """context options (include/cutesv_hip.h CSV_OPT_*): 1 = keep the ordered reads table across runs of one upload"""
        self._check(lib().csv_batch_option(self._h, int(option), int(value)))

    def last_reads_mode(self):
        """0 promised sorted, 1 run-level reorder on the device, 2 general radix sort, -1 no reads table"""
        return int(lib().csv_batch_reads_mode(self._h))

    def cache_flush(self, nbytes=1 << 30):
        """evict L2 / Infinity Cache (a measurement aid: the next run reads its columns from HBM)"""
        self._check(lib().csv_cache_flush(self._h, int(nbytes)))

    def result_buffers(self, per_sig=False, cap_calls=None, cap_support=None, pinned=True, no_support=False, coord32=False, fields=None, block=True):
        """caller-owned result arrays for the uploaded batch, to be handed to download(into=...) again and again; page-locked by
        default: the device then writes the calls straight into them (k_publish) and a download is one synchronisation.
        no_support / coord32 / fields: the slim forms of ABI v7 (_abi.HostResult).  block: the per-call arrays and the support list
        back to back in ONE page-locked block (what csv_batch_publish_async's copy-engine delivery wants)"""
        n = self._batch.n_sig
        return _abi.HostResult(n, cap_calls or max(64, n // 16 + 16), cap_support or max(64, n + 16), per_sig=per_sig,
                               n_seg=len(self._batch.segments), alloc=pinned_empty if pinned else None, narrow_support=True,
                               no_support=no_support, coord32=coord32, fields=fields, block=pinned_block if (pinned and block and not per_sig) else None)

    def lazy_info(self):
        """(gate-first?, signature-column bytes the bulk copy of the last upload did not send): csv_batch_info"""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(lib().csv_batch_info(self._h, 0, C.byref(a)))
        self._check(lib().csv_batch_info(self._h, 1, C.byref(b)))
        return bool(a.value), int(b.value)

    def reads_delta_info(self):
        """bit 0: the last upload's reads starts crossed as 16-bit gaps, bit 1: its ends as 16-bit lengths (CSV_IN_READS_DELTA16)"""
        a = C.c_int64(0)
        self._check(lib().csv_batch_info(self._h, 3, C.byref(a)))
        return int(a.value)

    def delta16_info(self):
        """did the last upload send its position column as 16-bit gaps (CSV_IN_SIG_DELTA16)?  csv_batch_info(2)"""
        a = C.c_int64(0)
        self._check(lib().csv_batch_info(self._h, 2, C.byref(a)))
        return bool(a.value)

    def download(self, per_sig=False, cap_calls=None, cap_support=None, into=None):
        if into is not None:
            # the library checks cap_calls / cap_support (E_CAPACITY); the arrays it fills WITHOUT a capacity of their own are
            # sized by the batch: seg_status (one word per segment), cluster_id / allele_id (one per signature)
            if into.n_seg < len(self._batch.segments):
                raise ValueError("download(into=...): the result has room for %d segments, the uploaded batch has %d"
                                 % (into.n_seg, len(self._batch.segments)))
            if into.per_sig and into.n_sig != self._batch.n_sig:
                raise ValueError("download(into=...): per-signature arrays of %d rows, the uploaded batch has %d" % (into.n_sig, self._batch.n_sig))
            self._check(lib().csv_batch_download(self._h, C.byref(into.c)))       # (E_CAPACITY: the caller sized `into`; it is raised)
            into.n_seg_used = len(self._batch.segments)
            return into
        n = self._batch.n_sig
        cap_calls = cap_calls or max(64, n // 16 + 16)
        cap_support = cap_support or max(64, n + 16)
        for _ in range(2):
            res = _abi.HostResult(n, cap_calls, cap_support, per_sig=per_sig, n_seg=len(self._batch.segments))
            rc = lib().csv_batch_download(self._h, C.byref(res.c))
            if rc == _abi.E_CAPACITY:            # required sizes were filled in: re-allocate and retry
                cap_calls, cap_support = res.n_calls + 1, res.n_support + 1
                continue
            self._check(rc)
            return res
        raise CsvError(_abi.E_CAPACITY, "capacity retry failed")

    # ---- pipelined delivery (csv_batch_publish_async / _wait): run k's result crosses PCIe while run k + 1 computes
    def publish_async(self, into):
        """start delivering the LAST run's result into `into` (page-locked HostResult, e.g. result_buffers()); returns at once"""
        if into.n_seg < len(self._batch.segments):
            raise ValueError("publish_async: the result has room for %d segments, the uploaded batch has %d" % (into.n_seg, len(self._batch.segments)))
        self._check(lib().csv_batch_publish_async(self._h, C.byref(into.c)))
        self._inflight = getattr(self, "_inflight", [])
        self._inflight.append(into)

    def publish_wait(self):
        """wait for the oldest delivery in flight; returns its HostResult (raises on its error status, e.g. E_CAPACITY)"""
        done = C.POINTER(_abi.BatchOut)()
        rc = lib().csv_batch_publish_wait(self._h, C.byref(done))
        res = self._inflight.pop(0) if getattr(self, "_inflight", None) else None
        self._check(rc)
        if res is not None:
            res.n_seg_used = len(self._batch.segments)
        return res

    # ---- one shot: csv_cluster_batch (H2D, kernels and D2H overlap inside the one call)
    # ---- a reuse=True result that its consumer keeps (rows.RowsBacking reads the arrays in place instead of copying them)
    def lend(self, res):
        """take `res` out of recycling until give_back(res); False if it is not this context's recycled result"""
        if res is not self._res_cache:
            return False
        self._res_cache = None
        return True

    def give_back(self, res):
        if self._res_cache is None:
            self._res_cache = res                # (otherwise it is simply dropped: a newer one took the slot)

    def cluster_batch(self, batch, per_sig=False, cap_calls=None, cap_support=None, reuse=False, no_support=False, coord32=False, fields=None):
        """reuse=True hands the C call the result arrays of this context's previous reuse=True call when they are large
        enough (a caller that consumes a result before asking for the next one, like resolve.run_batch, saves the
        page faults of ~20 MB of fresh arrays per call); the previous result is overwritten.
        no_support / coord32 / fields: the slim result forms of ABI v7 (what does not cross PCIe: _abi.HostResult)."""
        n = batch.n_sig
        cap_calls = cap_calls or max(64, n // 16 + 16)
        cap_support = cap_support or max(64, n + 16)         # (a signature supports at most one call)
        self._batch = batch
        self._inflight = []
        for _ in range(2):
            res = self._res_cache if reuse else None
            key = (bool(per_sig), bool(no_support), bool(coord32), None if fields is None else frozenset(fields), bool(reuse))
            if (res is None or res.cap_calls < cap_calls or res.cap_support < cap_support or res.n_seg < len(batch.segments)
                    or res.shape_key() != key or (per_sig and res.n_sig != n)):
                # (recycled arrays are worth page-locking: the result copies then land in them by DMA)
                res = _abi.HostResult(n, cap_calls, cap_support, per_sig=per_sig, n_seg=len(batch.segments),
                                      alloc=pinned_empty if reuse else None, narrow_support=bool(reuse),
                                      no_support=no_support, coord32=coord32, fields=fields)
                if reuse:
                    self._res_cache = res
            rc = lib().csv_cluster_batch(self._h, C.byref(batch.c), C.byref(res.c))
            if rc == _abi.E_CAPACITY:            # required sizes were filled in: re-allocate (never smaller) and retry
                cap_calls, cap_support = max(cap_calls, res.n_calls + 1), max(cap_support, res.n_support + 1)
                continue
            self._check(rc)
            res.n_seg_used = len(batch.segments)
            return res
        raise CsvError(_abi.E_CAPACITY, "capacity retry failed")


def stage_names():
    return [lib().csv_stage_name(i).decode() for i in range(_abi.N_STAGES)]


# ---- page-locked host memory (csv_host_alloc / csv_host_register): columns that live in it reach the GPU by DMA
class _PinnedBlock:
    def __init__(self, nbytes):
        p = C.c_void_p()
        rc = lib().csv_host_alloc(int(nbytes), C.byref(p))
        if rc != _abi.OK:
            raise CsvError(rc, "csv_host_alloc(%d) failed" % nbytes)
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                lib().csv_host_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in page-locked host memory (freed when the array and its views are gone)"""
    import numpy as np
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) if not isinstance(shape, int) else int(shape)
    blk = _PinnedBlock(max(1, n * dtype.itemsize))
    buf = (C.c_char * blk.nbytes).from_address(blk.ptr)
    buf._csv_block = blk                      # the array's base is `buf`; the block lives exactly as long
    return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def pinned_block(nbytes):
    """a writable page-locked buffer of nbytes (freed with the last array carved out of it)"""
    blk = _PinnedBlock(max(1, int(nbytes)))
    buf = (C.c_char * blk.nbytes).from_address(blk.ptr)
    buf._csv_block = blk
    return buf


def pinned_copy(arr):
    """copy of `arr` in page-locked host memory"""
    import numpy as np
    arr = np.ascontiguousarray(arr)
    out = pinned_empty(arr.shape, arr.dtype)
    out[...] = arr
    return out


class Registered:
    """An existing contiguous numpy array page-locked in place (csv_host_register).  Keeps the array alive and
    unregisters it when this object is released (or on close()): the library never holds a mapping of memory that has
    been freed."""

    def __init__(self, arr):
        self.arr = arr
        self.ok = lib().csv_host_register(C.c_void_p(arr.ctypes.data), int(arr.nbytes)) == _abi.OK

    def close(self):
        if self.ok:
            self.ok = False
            lib().csv_host_unregister(C.c_void_p(self.arr.ctypes.data))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __bool__(self):
        return self.ok


_REGISTERED = {}                   # address -> Registered: registrations made through host_register live until host_unregister


def host_register(arr):
    """page-lock an existing contiguous numpy array in place (csv_host_register).  The registration (and the array) stays alive
    until `host_unregister(arr)` - the contract of rounds 1-2: `if host_register(a): ... host_unregister(a)` - whatever
    the caller does with the returned handle (truthy on success).  For a registration that ends with a scope use
    `host_register_scoped`."""
    h = Registered(arr)
    if h.ok:
        _REGISTERED[int(arr.ctypes.data)] = h
    return h


def host_register_scoped(arr):
    """like host_register, but the returned `Registered` handle OWNS the registration: dropping it (or .close()) unregisters"""
    return Registered(arr)


def host_unregister(arr):
    """`arr`: the array that was registered, or the handle host_register / host_register_scoped returned"""
    if isinstance(arr, Registered):
        _REGISTERED.pop(int(arr.arr.ctypes.data), None)
        ok = arr.ok
        arr.close()
        return ok
    h = _REGISTERED.pop(int(arr.ctypes.data), None)
    if h is not None:
        ok = h.ok
        h.close()
        return ok
    return lib().csv_host_unregister(C.c_void_p(arr.ctypes.data)) == _abi.OK
