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


class Context:
    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().csv_ctx_create(int(device), C.byref(self._h))
        if rc != _abi.OK:
            raise CsvError(rc, "csv_ctx_create(device=%d) failed (is a GPU visible?)" % device)
        self.device = device
        self._batch = None

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
    def upload(self, batch):
        self._check(lib().csv_batch_upload(self._h, C.byref(batch.c)))
        self._batch = batch

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

    def download(self, per_sig=False, cap_calls=None, cap_support=None):
        n = self._batch.n_sig
        cap_calls = cap_calls or max(64, n // 16 + 16)
        cap_support = cap_support or max(64, n // 2 + 16)
        for _ in range(2):
            res = _abi.HostResult(n, cap_calls, cap_support, per_sig=per_sig)
            rc = lib().csv_batch_download(self._h, C.byref(res.c))
            if rc == _abi.E_CAPACITY:            # required sizes were filled in: re-allocate and retry
                cap_calls, cap_support = res.n_calls + 1, res.n_support + 1
                continue
            self._check(rc)
            return res
        raise CsvError(_abi.E_CAPACITY, "capacity retry failed")

    # ---- one shot
    def cluster_batch(self, batch, per_sig=False):
        self.upload(batch)
        self.run()
        return self.download(per_sig=per_sig)


def stage_names():
    return [lib().csv_stage_name(i).decode() for i in range(_abi.N_STAGES)]
