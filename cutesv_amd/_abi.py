"""ctypes mirror of include/cutesv_hip.h (structs, constants) and numpy-side batch buffers.

Nothing here computes anything: it only lays out memory for the C ABI.  The same struct
definitions are used by the tests to drive oracle/liboracle.so, which consumes the same
csv_batch_in / csv_batch_out layout.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 8
DEL, INS, DUP, INV, TRA = 0, 1, 2, 3, 4
SVTYPE_CODE = {"DEL": DEL, "INS": INS, "DUP": DUP, "INV": INV, "TRA": TRA}
SVTYPE_NAME = {v: k for k, v in SVTYPE_CODE.items()}

OK, E_INVALID, E_CAPACITY, E_HIP, E_NOMEM, E_UNSORTED, E_STATE = range(7)
ERR_NAME = {OK: "CSV_OK", E_INVALID: "CSV_E_INVALID", E_CAPACITY: "CSV_E_CAPACITY", E_HIP: "CSV_E_HIP",
            E_NOMEM: "CSV_E_NOMEM", E_UNSORTED: "CSV_E_UNSORTED", E_STATE: "CSV_E_STATE"}
N_STAGES = 24
GL_TABLE_SIZE = 101 * 101 + 2
IN_PER_SIG, IN_READS_SORTED, IN_SIG_I32, IN_READS_I32, IN_DEVICE_COLUMNS, IN_SIG_DELTA16, IN_READS_DELTA16 = 1, 2, 4, 8, 16, 32, 64     # csv_batch_in.flags
RB_KEEP_ON_DEVICE = 1
RB_FROM_POOL = 2                        # ... the rows are the context's device-resident signature pool
CG_TO_POOL = 1                         # csv_cigar_in.flags: the signatures also become pool rows                         # csv_rebuild_in.flags
SEG_KEY_RANGE = 1                             # csv_batch_out.seg_status bits
OUT_NO_SUPPORT_LIST, OUT_COORD_I32 = 1, 2     # csv_batch_out.flags (ABI v7)
OPTIONAL_CALL_FIELDS = ("call_cluster", "call_aux", "cipos", "cilen", "search_pos", "seq_pick", "dr", "dv", "gl_idx")
COORD_FIELDS = ("bp1", "bp2", "search_pos", "seq_pick")

# numpy dtype with exactly the C layout of `csv_segment` (all members naturally aligned)
SEGMENT_DTYPE = np.dtype([
    ("svtype", "<i4"), ("chrom", "<i4"),
    ("sig_begin", "<i8"), ("sig_end", "<i8"),
    ("max_cluster_bias", "<i8"),
    ("diff_ratio", "<f8"), ("remain_reads_ratio", "<f8"),
    ("sv_size", "<i8"), ("max_size", "<i8"), ("gt_bias", "<i8"),
    ("read_count", "<i4"), ("min_support_reads", "<i4"),
    ("genotype", "<i4"), ("gt_round", "<i4"),
], align=True)
assert SEGMENT_DTYPE.itemsize == 88


class BatchIn(C.Structure):
    _fields_ = [
        ("n_seg", C.c_int32), ("n_chrom", C.c_int32),
        ("seg", C.c_void_p),
        ("n_sig", C.c_int64),
        ("a", C.c_void_p), ("b", C.c_void_p), ("read_id", C.c_void_p), ("aux", C.c_void_p),
        ("reads_off", C.c_void_p),
        ("n_reads", C.c_int64),
        ("r_start", C.c_void_p), ("r_end", C.c_void_p), ("r_primary", C.c_void_p), ("r_id", C.c_void_p),
        ("contig_len", C.c_void_p),
        ("flags", C.c_int32), ("reserved", C.c_int32),
        ("a_delta", C.c_void_p), ("n_esc", C.c_int64), ("a_esc_row", C.c_void_p), ("a_esc_val", C.c_void_p),      # ABI v8: CSV_IN_SIG_DELTA16
        ("rows8", C.c_void_p),                                                                                      # ABI v8: {b, read_id} interleaved
        ("r_delta", C.c_void_p), ("n_r_esc", C.c_int64), ("r_esc_row", C.c_void_p), ("r_esc_val", C.c_void_p),      # ABI v8: CSV_IN_READS_DELTA16
        ("r_len16", C.c_void_p), ("n_l_esc", C.c_int64), ("l_esc_row", C.c_void_p), ("l_esc_val", C.c_void_p),
        ("r_idp", C.c_void_p),                                                                                      # ABI v8: r_id | r_primary << 31
    ]


_OUT_ARRAYS = [  # (name, dtype, which capacity)
    ("call_seg", np.int32, "calls"), ("call_cluster", np.int32, "calls"), ("call_aux", np.int32, "calls"),
    ("bp1", np.int64, "calls"), ("bp2", np.int64, "calls"),
    ("support", np.int32, "calls"), ("cipos", np.int32, "calls"), ("cilen", np.int32, "calls"),
    ("search_pos", np.int64, "calls"), ("seq_pick", np.int64, "calls"),
    ("dr", np.int32, "calls"), ("dv", np.int32, "calls"), ("gl_idx", np.int32, "calls"),
    ("support_off", np.int64, "calls+1"), ("support_sig", np.int64, "support"),
    ("cluster_id", np.int32, "sig"), ("allele_id", np.int32, "sig"), ("seg_status", np.int32, "seg"),
]


class BatchOut(C.Structure):
    _fields_ = ([("cap_calls", C.c_int64), ("cap_support", C.c_int64),
                 ("n_calls", C.c_int64), ("n_support", C.c_int64), ("n_clusters", C.c_int64)]
                + [(name, C.c_void_p) for name, _, _ in _OUT_ARRAYS] + [("support_sig32", C.c_void_p), ("flags", C.c_int32), ("reserved", C.c_int32)])


class RunStats(C.Structure):
    _fields_ = [("ms_total", C.c_float), ("ms_stage", C.c_float * N_STAGES),
                ("n_clusters", C.c_int64), ("n_work_wave", C.c_int64), ("n_work_block", C.c_int64),
                ("n_calls", C.c_int64), ("n_support", C.c_int64)]


def _ptr(arr):
    return None if arr is None else arr.ctypes.data


def _col(x, dtype):
    return np.ascontiguousarray(x, dtype=dtype)


class HostBatch:
    """Host-side buffers of one csv_batch_in.  Keeps the numpy arrays alive for the C call."""

    def __init__(self, segments, a, b, read_id, aux, n_chrom=0, reads_off=None,
                 r_start=None, r_end=None, r_primary=None, r_id=None, contig_len=None, per_sig=False, reads_sorted=False, a_delta=None, rows8=None, r_delta=None, r_len16=None, r_idp=None):
        """a_delta: (delta uint16[n_sig], escape rows int64[], escape values int32[]) of `a` - delta16_of(a) - when the position
        column may cross the link as 16-bit gaps (CSV_IN_SIG_DELTA16; int32 columns only)"""
        self.segments = np.ascontiguousarray(segments, dtype=SEGMENT_DTYPE)
        # positions / lengths may come as int32 columns (CSV_IN_SIG_I32: a third less data on the link); both alike
        sig32 = getattr(a, "dtype", None) == np.int32 and getattr(b, "dtype", None) == np.int32
        self.a = _col(a, np.int32 if sig32 else np.int64)
        self.b = _col(b, np.int32 if sig32 else np.int64)
        self.read_id = _col(read_id, np.int32)
        self.aux = _col(aux, np.int32)
        n = self.a.shape[0]
        if not (self.b.shape[0] == n and self.read_id.shape[0] == n and self.aux.shape[0] == n):
            raise ValueError("signature columns differ in length")
        self.n_chrom = int(n_chrom)
        if reads_off is not None:
            self.reads_off = _col(reads_off, np.int64)
            rd32 = getattr(r_start, "dtype", None) == np.int32 and getattr(r_end, "dtype", None) == np.int32
            self.r_start = _col(r_start, np.int32 if rd32 else np.int64)
            self.r_end = _col(r_end, np.int32 if rd32 else np.int64)
            self.r_primary = _col(r_primary, np.uint8)
            self.r_id = _col(r_id, np.int32)
            if self.reads_off.shape[0] != self.n_chrom + 1:
                raise ValueError("reads_off must have n_chrom + 1 entries")
        else:
            self.reads_off = self.r_start = self.r_end = self.r_primary = self.r_id = None
        self.contig_len = None if contig_len is None else _col(contig_len, np.int64)
        if self.contig_len is not None and self.contig_len.shape[0] != self.n_chrom:
            raise ValueError("contig_len must have n_chrom entries")
        # rows8: the (n, 2) int32 array {b, read_id} per signature (page-locked: SigStore.pinned()) for the gate-first fetch
        self.rows8 = None
        if rows8 is not None and self.a.dtype == np.int32:
            self.rows8 = _col(rows8, np.int32)
            if self.rows8.shape != (n, 2):
                raise ValueError("rows8: one {b, read_id} pair per signature is expected")
        self.a_delta = None
        if a_delta is not None and self.a.dtype == np.int32:
            d, er, ev = a_delta
            self.a_delta = (_col(d, np.uint16), _col(er, np.int64), _col(ev, np.int32))
            if self.a_delta[0].shape[0] != n or self.a_delta[1].shape[0] != self.a_delta[2].shape[0]:
                raise ValueError("a_delta: one gap per signature and one value per escape row are expected")
        # r_delta / r_len16: (uint16[n_reads], escape rows int64[], escape values int32[]) - delta16_of(r_start), len16_of(r_start, r_end)
        self.r_delta = self.r_len16 = None
        if self.r_start is not None and self.r_start.dtype == np.int32:
            nr = self.r_start.shape[0]
            for name, v in (("r_delta", r_delta), ("r_len16", r_len16)):
                if v is not None:
                    t3 = (_col(v[0], np.uint16), _col(v[1], np.int64), _col(v[2], np.int32))
                    if t3[0].shape[0] != nr or t3[1].shape[0] != t3[2].shape[0]:
                        raise ValueError("%s: one entry per read and one value per escape row are expected" % name)
                    setattr(self, name, t3)
        rdx = {}
        self.r_idp = None
        if r_idp is not None and self.r_id is not None:
            self.r_idp = _col(r_idp, np.uint32)
            if self.r_idp.shape[0] != self.r_id.shape[0]:
                raise ValueError("r_idp: one word per read is expected")
            rdx["r_idp"] = _ptr(self.r_idp)
        if self.r_delta is not None:
            rdx.update(r_delta=_ptr(self.r_delta[0]), n_r_esc=self.r_delta[1].shape[0], r_esc_row=_ptr(self.r_delta[1]), r_esc_val=_ptr(self.r_delta[2]))
        if self.r_len16 is not None:
            rdx.update(r_len16=_ptr(self.r_len16[0]), n_l_esc=self.r_len16[1].shape[0], l_esc_row=_ptr(self.r_len16[1]), l_esc_val=_ptr(self.r_len16[2]))
        self.c = BatchIn(
            n_seg=len(self.segments), n_chrom=self.n_chrom, seg=_ptr(self.segments),
            n_sig=n, a=_ptr(self.a), b=_ptr(self.b), read_id=_ptr(self.read_id), aux=_ptr(self.aux),
            reads_off=_ptr(self.reads_off), **rdx,
            n_reads=0 if self.r_start is None else self.r_start.shape[0],
            r_start=_ptr(self.r_start), r_end=_ptr(self.r_end), r_primary=_ptr(self.r_primary), r_id=_ptr(self.r_id),
            contig_len=_ptr(self.contig_len),
            flags=(IN_PER_SIG if per_sig else 0) | (IN_READS_SORTED if reads_sorted else 0) | (IN_SIG_I32 if self.a.dtype == np.int32 else 0)
            | (IN_READS_I32 if self.r_start is not None and self.r_start.dtype == np.int32 else 0) | (IN_SIG_DELTA16 if self.a_delta is not None else 0)
            | (IN_READS_DELTA16 if (self.r_delta is not None or self.r_len16 is not None) else 0),
            a_delta=None if self.a_delta is None else _ptr(self.a_delta[0]), n_esc=0 if self.a_delta is None else self.a_delta[1].shape[0],
            a_esc_row=None if self.a_delta is None else _ptr(self.a_delta[1]), a_esc_val=None if self.a_delta is None else _ptr(self.a_delta[2]),
            rows8=_ptr(self.rows8))

    @classmethod
    def on_device(cls, segments, dev, n_sig, n_chrom=0, keep=None, **reads):
        """A batch whose signature columns already live in device memory (CSV_IN_DEVICE_COLUMNS): `dev` = dict(a=, b=, read_id=,
        aux=) of device addresses (int64 a / b, int32 read_id / aux: what csv_rebuild_signatures leaves with
        CSV_RB_KEEP_ON_DEVICE); `keep`: objects that own that memory.  The reads table, if any, comes from the host as usual."""
        self = cls.__new__(cls)
        self.segments = np.ascontiguousarray(segments, dtype=SEGMENT_DTYPE)
        self.a = self.b = self.read_id = self.aux = None
        self._n_sig, self._keep = int(n_sig), keep
        self.n_chrom = int(n_chrom)
        self.reads_off = self.r_start = self.r_end = self.r_primary = self.r_id = self.contig_len = None
        if reads.get("reads_off") is not None:
            self.reads_off = _col(reads["reads_off"], np.int64)
            rd32 = getattr(reads["r_start"], "dtype", None) == np.int32 and getattr(reads["r_end"], "dtype", None) == np.int32
            self.r_start = _col(reads["r_start"], np.int32 if rd32 else np.int64); self.r_end = _col(reads["r_end"], np.int32 if rd32 else np.int64)
            self.r_primary = _col(reads["r_primary"], np.uint8); self.r_id = _col(reads["r_id"], np.int32)
        if reads.get("contig_len") is not None:
            self.contig_len = _col(reads["contig_len"], np.int64)
        self.c = BatchIn(
            n_seg=len(self.segments), n_chrom=self.n_chrom, seg=_ptr(self.segments), n_sig=self._n_sig,
            a=dev["a"], b=dev["b"], read_id=dev["read_id"], aux=dev["aux"], reads_off=_ptr(self.reads_off),
            n_reads=0 if self.r_start is None else self.r_start.shape[0],
            r_start=_ptr(self.r_start), r_end=_ptr(self.r_end), r_primary=_ptr(self.r_primary), r_id=_ptr(self.r_id),
            contig_len=_ptr(self.contig_len),
            flags=IN_DEVICE_COLUMNS | (IN_READS_I32 if self.r_start is not None and self.r_start.dtype == np.int32 else 0))
        return self

    def widened(self):
        """the same batch with int64 position columns (what the oracle takes)"""
        if self.a.dtype != np.int32 and (self.r_start is None or self.r_start.dtype != np.int32):
            return self
        kw = {}
        if self.reads_off is not None:
            kw = dict(reads_off=self.reads_off, r_start=self.r_start.astype(np.int64), r_end=self.r_end.astype(np.int64),
                      r_primary=self.r_primary, r_id=self.r_id)
        flags = self.c.flags
        return HostBatch(self.segments, self.a.astype(np.int64), self.b.astype(np.int64), self.read_id, self.aux, n_chrom=self.n_chrom,
                         contig_len=self.contig_len, per_sig=bool(flags & IN_PER_SIG), reads_sorted=bool(flags & IN_READS_SORTED), **kw)

    @property
    def n_sig(self):
        return self.a.shape[0] if self.a is not None else self._n_sig


class HostResult:
    """Caller-allocated csv_batch_out plus numpy views on it."""

    def __init__(self, n_sig, cap_calls, cap_support, per_sig=False, n_seg=0, alloc=None, narrow_support=False,
                 no_support=False, coord32=False, fields=None, seg_alloc=None, block=None):
        """alloc(shape, dtype) -> array: where the result arrays live (default numpy; engine.pinned_empty puts them in
        page-locked memory, so that the device-to-host copies land in them by DMA).  narrow_support: the support list as int32
        (csv_batch_out.support_sig32): `arrays["support_sig"]` is then an int32 array - every consumer indexes with it.
        ABI v7: no_support - CSV_OUT_NO_SUPPORT_LIST (support_off / support_sig are None); coord32 - CSV_OUT_COORD_I32 (bp1, bp2,
        search_pos, seq_pick are int32 arrays; needs int32 input columns); fields - the OPTIONAL_CALL_FIELDS to carry (None: all),
        the others stay None and are not written.  seg_alloc: where `seg_status` lives (default: an ordinary numpy array even
        under `alloc` - one word per segment is not worth a page-locked block of its own; broker.Client puts it in its shared region).
        block(nbytes) -> writable buffer: the per-call arrays and the support list are carved back to back out of ONE buffer (widest
        elements first, no gaps) - csv_batch_publish_async then moves them with the copy engine in one piece (engine.pinned_block)"""
        empty = alloc or (lambda n, dt: np.empty(n, dtype=dt))
        if block is not None:
            empty = self._carver(block, cap_calls, cap_support, no_support, coord32, fields, narrow_support)
        self.cap_calls = int(cap_calls)
        self.cap_support = int(cap_support)
        self.n_sig, self.n_seg, self.per_sig = int(n_sig), int(n_seg), bool(per_sig)
        self.no_support, self.coord32 = bool(no_support), bool(coord32)
        self.fields = None if fields is None else frozenset(fields)
        self.arrays = {}
        kw = {}
        for name, dt, cap in _OUT_ARRAYS:
            if name in COORD_FIELDS and coord32:
                dt = np.int32
            if self.fields is not None and name in OPTIONAL_CALL_FIELDS and name not in self.fields:
                arr = None
            elif cap == "sig":
                arr = empty(n_sig, dt) if per_sig else None
            elif cap == "seg":
                arr = np.zeros(max(1, n_seg), dtype=dt) if seg_alloc is None else seg_alloc(max(1, n_seg), dt)
                arr[:] = 0
            elif cap == "calls":
                arr = empty(self.cap_calls, dt)
            elif no_support:
                arr = None
            elif cap == "calls+1":
                arr = empty(self.cap_calls + 1, dt)
                arr[:] = 0
            else:
                arr = empty(self.cap_support, np.int32 if narrow_support else dt)
            self.arrays[name] = arr
            kw[name] = _ptr(arr)
        if narrow_support:
            kw["support_sig32"], kw["support_sig"] = kw["support_sig"], None
        self.narrow_support = bool(narrow_support)
        self.n_seg_used = self.n_seg                 # segments of the batch the arrays were last filled for (a recycled result may be larger)
        self.c = BatchOut(cap_calls=self.cap_calls, cap_support=self.cap_support,
                          flags=(OUT_NO_SUPPORT_LIST if no_support else 0) | (OUT_COORD_I32 if coord32 else 0), **kw)

    @staticmethod
    def _carver(block, cap_calls, cap_support, no_support, coord32, fields, narrow_support):
        """alloc(n, dtype) over one buffer: the arrays __init__ is about to ask for, in its order, each at a fixed place"""
        want = []
        for name, dt, cap in _OUT_ARRAYS:
            if name in COORD_FIELDS and coord32:
                dt = np.int32
            if fields is not None and name in OPTIONAL_CALL_FIELDS and name not in frozenset(fields):
                continue
            if cap == "calls":
                want.append((name, np.dtype(dt), int(cap_calls)))
            elif cap == "calls+1" and not no_support:
                want.append((name, np.dtype(dt), int(cap_calls) + 1))
            elif cap == "support" and not no_support:
                want.append((name, np.dtype(np.int32 if narrow_support else dt), int(cap_support)))
        order = sorted(range(len(want)), key=lambda i: (-want[i][1].itemsize, want[i][0] == "support_sig", i))
        off, place = 0, {}
        for i in order:
            place[i] = off
            off += want[i][1].itemsize * want[i][2]
        buf = block(max(8, off))
        seq = iter(range(len(want)))

        def empty(n, dt):
            i = next(seq)
            assert want[i][2] == int(n) and want[i][1] == np.dtype(dt), (want[i], n, dt)
            return np.frombuffer(buf, dtype=dt, count=int(n), offset=place[i])
        return empty

    def snapshot(self):
        """a private copy cut to the produced sizes (ordinary memory): what outlives the recycled, page-locked arrays of a
        reuse=True call (rows.LazyRows keeps one)"""
        out = HostResult(self.n_sig, max(1, self.n_calls), max(1, self.n_support), per_sig=False, n_seg=max(1, self.n_seg_used),
                         narrow_support=self.narrow_support, no_support=self.no_support, coord32=self.coord32, fields=self.fields)
        t = self.trimmed()
        for name, _, cap in _OUT_ARRAYS:
            if cap != "sig" and out.arrays.get(name) is not None and t.get(name) is not None:
                out.arrays[name][:len(t[name])] = t[name]
        out.c.n_calls, out.c.n_support, out.c.n_clusters = self.n_calls, self.n_support, self.n_clusters
        out.n_seg_used = self.n_seg_used
        return out

    def shape_key(self):
        """what a recycled result must agree on with a request besides its capacities"""
        return (self.per_sig, self.no_support, self.coord32, self.fields, self.narrow_support)

    @property
    def n_calls(self):
        return int(self.c.n_calls)

    @property
    def n_support(self):
        return int(self.c.n_support)

    @property
    def n_clusters(self):
        return int(self.c.n_clusters)

    def trimmed(self):
        """dict of arrays cut to the produced sizes (views)."""
        nc, ns = self.n_calls, self.n_support
        out = {}
        for name, _, cap in _OUT_ARRAYS:
            arr = self.arrays[name]
            if arr is None:
                out[name] = None
            elif cap == "calls":
                out[name] = arr[:nc]
            elif cap == "calls+1":
                out[name] = arr[:nc + 1]
            elif cap == "support":
                out[name] = arr[:ns]
            elif cap == "seg":
                out[name] = arr[:max(1, self.n_seg_used)]
            else:
                out[name] = arr
        out["n_clusters"] = self.n_clusters
        return out


def delta16_of(a, alloc=None):
    """The position column as CSV_IN_SIG_DELTA16 takes it: (gaps uint16[n], escape rows int64[], escape values int32[]).
    gap[i] = a[i] - a[i - 1] where that lies in [0, 0xFFFF) (i > 0), else 0xFFFF with (i, a[i]) in the escape list.
    alloc(shape, dtype): where the gap array lives (engine.pinned_empty for a page-locked one)."""
    a = np.ascontiguousarray(a, np.int32)
    n = a.shape[0]
    d = np.empty(n, np.int64)
    if n:
        d[0] = -1
        np.subtract(a[1:], a[:-1], out=d[1:], dtype=np.int64)
    esc = (d < 0) | (d >= 0xFFFF)
    out = (alloc or (lambda s_, dt: np.empty(s_, dt)))(n, np.uint16)
    np.copyto(out, np.where(esc, 0xFFFF, d), casting="unsafe")
    rows = np.flatnonzero(esc).astype(np.int64)
    return out, rows, a[rows].astype(np.int32)


def len16_of(start, end, alloc=None):
    """The reads table's end column as CSV_IN_READS_DELTA16 takes it: (lengths uint16[n], escape rows int64[], escape values int32[]);
    length[i] = end[i] - start[i] where that lies in [0, 0xFFFF), else 0xFFFF with (i, end[i]) in the escape list."""
    start, end = np.ascontiguousarray(start, np.int32), np.ascontiguousarray(end, np.int32)
    d = end.astype(np.int64) - start
    esc = (d < 0) | (d >= 0xFFFF)
    out = (alloc or (lambda s_, dt: np.empty(s_, dt)))(len(d), np.uint16)
    np.copyto(out, np.where(esc, 0xFFFF, d), casting="unsafe")
    rows = np.flatnonzero(esc).astype(np.int64)
    return out, rows, end[rows].astype(np.int32)


def make_segment(svtype, chrom, sig_begin, sig_end, max_cluster_bias, read_count, diff_ratio=0.0,
                 remain_reads_ratio=1.0, sv_size=0, max_size=-1, gt_bias=0, min_support_reads=None, genotype=False,
                 gt_round=0):
    """One csv_segment record (numpy void) from the reference's run_* scalars."""
    s = np.zeros((), dtype=SEGMENT_DTYPE)
    s["svtype"] = SVTYPE_CODE[svtype] if isinstance(svtype, str) else svtype
    s["chrom"] = chrom
    s["sig_begin"], s["sig_end"] = sig_begin, sig_end
    s["max_cluster_bias"] = max_cluster_bias
    s["diff_ratio"] = diff_ratio
    s["remain_reads_ratio"] = remain_reads_ratio
    s["sv_size"], s["max_size"], s["gt_bias"] = sv_size, max_size, gt_bias
    s["read_count"] = read_count
    s["min_support_reads"] = min(read_count, 5) if min_support_reads is None else min_support_reads
    s["genotype"] = 1 if genotype else 0
    s["gt_round"] = gt_round
    return s
