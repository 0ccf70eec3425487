"""TEST INFRASTRUCTURE ONLY: cutesv_amd.broker's serving loop with the C oracle behind it instead of libcutesv_hip.so.

The CPU suite (no GPU in the build container) uses it to exercise everything of the pool path that is not a kernel: the
socket protocol, the shared-memory regions and pointer rebasing, the merging of waiting requests into one batch and the
slicing of its result, the workers' side under a forked `multiprocessing.Pool`.  The product never starts it:
`broker.spawn` runs `python -m cutesv_amd.broker`, whose engine is the HIP library and nothing else.

    python tests/broker_oracle.py --name <socket name> --device 0 --watch-pid <pid>
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cutesv_amd import _abi, broker          # noqa: E402


def _view(addr, n, dtype):
    return broker._view(addr, n, dtype)


class OracleEngine:
    """csv_cluster_batch's contract (optional result fields, int32 coordinates, the narrow support list, capacity negotiation)
    on top of csvo_cluster_batch, which fills the full int64 structure of arrays"""

    def __init__(self, device):
        from oracle import oracle
        self.oracle = oracle
        oracle.lib()
        self.device = device
        self.err = ""
        self.n_calls = 0
        self.batch_sizes = []

    def call(self, cin, cout):
        self.n_calls += 1
        k, n, nr = int(cin.n_seg), int(cin.n_sig), int(cin.n_reads)
        self.batch_sizes.append(k)
        if cin.flags & _abi.IN_DEVICE_COLUMNS:
            self.err = "device columns are not the oracle's"
            return _abi.E_INVALID
        sdt = np.int32 if cin.flags & _abi.IN_SIG_I32 else np.int64
        rdt = np.int32 if cin.flags & _abi.IN_READS_I32 else np.int64
        kw = {}
        if cin.reads_off:
            kw = dict(reads_off=_view(cin.reads_off, int(cin.n_chrom) + 1, np.int64), r_start=_view(cin.r_start, nr, rdt).astype(np.int64),
                      r_end=_view(cin.r_end, nr, rdt).astype(np.int64), r_primary=_view(cin.r_primary, nr, np.uint8), r_id=_view(cin.r_id, nr, np.int32))
        if cin.contig_len:
            kw["contig_len"] = _view(cin.contig_len, int(cin.n_chrom), np.int64)
        hb = _abi.HostBatch(_view(cin.seg, k, _abi.SEGMENT_DTYPE), _view(cin.a, n, sdt).astype(np.int64), _view(cin.b, n, sdt).astype(np.int64),
                            _view(cin.read_id, n, np.int32), _view(cin.aux, n, np.int32), n_chrom=int(cin.n_chrom),
                            reads_sorted=bool(cin.flags & _abi.IN_READS_SORTED), **kw)
        per_sig = bool(cout.cluster_id or cout.allele_id)
        try:
            res = self.oracle.cluster_batch(hb, per_sig=per_sig)
        except RuntimeError as e:
            self.err = str(e)
            return _abi.E_INVALID
        t = res.trimmed()
        nc, ns = res.n_calls, res.n_support
        no_sup = bool(cout.flags & _abi.OUT_NO_SUPPORT_LIST)
        cout.n_calls, cout.n_support, cout.n_clusters = nc, (0 if no_sup else ns), res.n_clusters
        if nc > cout.cap_calls or (not no_sup and ns > cout.cap_support):
            cout.n_support = ns
            return _abi.E_CAPACITY
        for name, dt, cap in _abi._OUT_ARRAYS:
            dst = getattr(cout, name)
            if not dst or cap not in ("calls", "sig"):
                continue
            if name in _abi.COORD_FIELDS and cout.flags & _abi.OUT_COORD_I32:
                dt = np.int32
            _view(dst, nc if cap == "calls" else n, dt)[:] = t[name]
        if not no_sup:
            if cout.support_off:
                _view(cout.support_off, nc + 1, np.int64)[:] = t["support_off"]
            if cout.support_sig32:
                _view(cout.support_sig32, ns, np.int32)[:] = t["support_sig"]
            elif cout.support_sig:
                _view(cout.support_sig, ns, np.int64)[:] = t["support_sig"]
        if cout.seg_status:
            _view(cout.seg_status, k, np.int32)[:] = t["seg_status"][:k]
        return _abi.OK

    def last_error(self):
        return self.err

    def alloc(self, shape, dtype):
        return np.empty(shape, dtype)

    def register(self, addr, size):
        return False

    def unregister(self, addr):
        pass

    def describe(self):
        return dict(bus="oracle", compute_units=0, engine="oracle/liboracle.so (tests only)", engine_calls=self.n_calls,
                    engine_batch_sizes=self.batch_sizes[-64:])

    def close(self):
        pass


if __name__ == "__main__":
    sys.exit(broker.main(engine_factory=OracleEngine))
