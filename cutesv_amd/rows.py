"""Output side of the hot path: the structure-of-arrays result -> the reference's row lists.

Row layouts are the output contract of the stage (SURVEY.md §8b "Return rows"), consumed
unchanged by the reference's VCF emitter (cuteSV_genotype.py:263-458):
    DEL 13 fields / INS 14   cuteSV_resolveINDEL.py:207-219, 419-432 (no genotype), :464-478 (genotype)
    DUP 11                   cuteSV_resolveDUP.py:121-131, 170-180
    INV 12                   cuteSV_resolveINV.py:145-156, 240-251
    TRA 12                   cuteSV_resolveTRA.py:171-182 (genotype fields from call_gt, :258-309)
All numeric fields are `str`; read names are joined by ','.

`materialise` is the product path: the CPython extension `_rows_native` (cutesv_amd/csrc/rows_py.cpp over rows_layout.h)
formats the calls on worker threads and creates the list / str objects on the calling thread - ~10 ms for the 25 k calls
of a 30x genome; `csv_rows_emit` (cutesv_amd/csrc/rows_emit.cpp, C ABI) writes the same rows as one text blob for hosts
that are not CPython.  `materialise_py` is the per-call Python loop it
replaced (round 1: ~90 ms); it is kept as the readable statement of the layouts and the tests compare the two.
"""
import ctypes as C
import weakref

import numpy as np

from . import _abi
from ._lib import lib
from .genotype import gl_fields, gl_table_blob

_TRA_ALT = ("N[%s[", "N]%s]", "[%s[N", "]%s]N")      # cuteSV_resolveTRA.py:142-153


class RowsIn(C.Structure):
    _fields_ = [
        ("res", C.POINTER(_abi.BatchOut)), ("seg", C.c_void_p), ("n_seg", C.c_int32), ("n_chrom", C.c_int32),
        ("chrom_name", C.POINTER(C.c_char_p)),
        ("read_id", C.c_void_p), ("aux", C.c_void_p),
        ("name_blob", C.c_char_p), ("name_off", C.c_void_p), ("n_names", C.c_int64),
        ("name_prefix", C.c_char_p), ("name_width", C.c_int32), ("n_strand", C.c_int32),
        ("ins_blob", C.c_char_p), ("ins_off", C.c_void_p),
        ("strand_name", C.POINTER(C.c_char_p)),
        ("gl_blob", C.c_char_p), ("gl_off", C.c_void_p),
    ]


def _native():
    from . import _rows_native as m       # built by `make -C cutesv_amd/csrc`; no pure-Python stand-in on the product path
    return m


def _rows_in(store, segments, res):
    """-> (csv_rows_in, objects to keep alive while it is used)"""
    segs = np.ascontiguousarray(segments, dtype=_abi.SEGMENT_DTYPE)
    nch = len(store.chroms)
    names = (C.c_char_p * max(1, nch))(*[c.encode() for c in store.chroms])
    strands = (C.c_char_p * max(1, len(store.strands)))(*[s.encode() for s in store.strands])
    n = res.n_calls
    # (an explicit name table / explicit sequences: only the entries these calls mention are encoded, unless the store already
    # holds its full blobs - see SigStore.names_blob)
    nb = store.names_blob(picks=store.read_id[res.arrays["support_sig"][:res.n_support]] if store.names.names is not None else None)
    ib = store.ins_blob(picks=res.arrays["seq_pick"][:n])
    gl = res.arrays["gl_idx"][:n]
    glb, glo = gl_table_blob(np.unique(gl[gl >= 0]) if n else ())
    rid = np.ascontiguousarray(store.read_id, np.int32)
    aux = np.ascontiguousarray(store.aux, np.int32)
    rin = RowsIn(res=C.pointer(res.c), seg=segs.ctypes.data, n_seg=len(segs), n_chrom=nch, chrom_name=names,
                 read_id=rid.ctypes.data, aux=aux.ctypes.data,
                 name_blob=nb[0], name_off=None if nb[1] is None else nb[1].ctypes.data, n_names=nb[2],
                 name_prefix=nb[3], name_width=nb[4], n_strand=len(store.strands),
                 ins_blob=ib[0], ins_off=None if ib[1] is None else ib[1].ctypes.data,
                 strand_name=strands, gl_blob=glb, gl_off=glo.ctypes.data)
    return rin, (segs, names, strands, nb, ib, glb, glo, rid, aux, res)


def rows_blob(store, segments, res):
    """The C ABI's form of the rows (csv_rows_emit): -> (bytes, number of rows); fields '\\t', rows '\\n'"""
    L = lib()
    L.csv_rows_emit.restype = C.c_int
    L.csv_rows_emit.argtypes = [C.POINTER(RowsIn), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    rin, keep = _rows_in(store, segments, res)
    need = C.c_int64(0)
    rc = L.csv_rows_emit(C.byref(rin), None, 0, C.byref(need))       # first pass only counts
    if rc not in (_abi.OK, _abi.E_CAPACITY):
        raise RuntimeError("csv_rows_emit: %s" % _abi.ERR_NAME.get(rc, rc))
    buf = bytearray(need.value)
    cbuf = (C.c_char * max(1, need.value)).from_buffer(buf) if need.value else None
    if need.value:
        rc = L.csv_rows_emit(C.byref(rin), cbuf, need.value, C.byref(need))
        if rc != _abi.OK:
            raise RuntimeError("csv_rows_emit: %s" % _abi.ERR_NAME.get(rc, rc))
    del cbuf
    return bytes(buf), res.n_calls


def materialise(store, segments, res):
    """res: _abi.HostResult of a batch -> (rows, call_seg): `rows` = one list of str per call, in call order (the
    reference's emission order), `call_seg[c]` = index of the segment (task) call c belongs to (non-decreasing)."""
    n = res.n_calls
    if n == 0:
        return [], np.zeros(0, np.int32)
    rin, keep = _rows_in(store, segments, res)
    rows = _native().build(C.addressof(rin))
    del keep
    return rows, res.arrays["call_seg"][:n]


def rows_by_segment(store, segments, res):
    """-> list (per segment) of row lists"""
    rows, call_seg = materialise(store, segments, res)
    n_seg = len(segments)
    cut = np.searchsorted(call_seg, np.arange(n_seg + 1)).tolist()
    return [rows[cut[k]:cut[k + 1]] for k in range(n_seg)]


# ------------------------------------------------------------------------------------------------ lazy rows
class RowsBacking:
    """What a LazyRows sequence is made from: the store, the batch's segments and a private copy of the result's structure of
    arrays.  The csv_rows_in block (name / sequence tables) is built on the first row anyone asks for."""

    def __init__(self, store, segments, res, ctx=None):
        # a recycled (reuse=True) result is overwritten by the context's next call: either the context lends it for the life of
        # this object (no copy: the 0.6 ms of a 30x genome's stage), or a private copy is cut to the produced sizes
        if ctx is not None and getattr(ctx, "lend", None) is not None and ctx.lend(res):
            weakref.finalize(self, ctx.give_back, res)
        else:
            res = res.snapshot()
        self.store, self.segments, self.res = store, np.ascontiguousarray(segments, dtype=_abi.SEGMENT_DTYPE), res
        self._rin = None
        n = self.res.n_calls
        self.key = np.asarray(self.res.arrays["bp1"][:n], dtype=np.int64)        # int(row[2]) of every call: what generate_output sorts by

    def rin(self):
        if self._rin is None:
            self._rin = _rows_in(self.store, self.segments, self.res)
        return self._rin[0]

    def build(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        if len(idx) == 0:
            return []
        return _native().build_some(C.addressof(self.rin()), idx)


class LazyRows:
    """The rows of some calls as the list the reference's consumers expect (main script :1191-1197 `results[chr].extend(rows)`,
    cuteSV_genotype.py:242-252 `semi_result.sort(key=lambda x: int(x[2]))` and iteration) WITHOUT the strings: a row's 11-14
    str objects are created when it is indexed or iterated (in blocks of 1024), `sort` by `int(row[2])` is answered from the
    bp1 column.  A consumer that reads the structure of arrays itself (vcf.emit_records: `backing()`) never creates one.
    Parts are (RowsBacking, call indices) or plain lists of rows (e.g. tra_bam.genotype_rows' output)."""
    __slots__ = ("_parts",)
    BLOCK = 1024

    def __init__(self, backing=None, idx=None):
        self._parts = [] if backing is None else [(backing, np.ascontiguousarray(idx, dtype=np.int64))]

    # ---- list protocol
    def __len__(self):
        return sum(len(p) if isinstance(p, list) else len(p[1]) for p in self._parts)

    def __iter__(self):
        for p in self._parts:
            if isinstance(p, list):
                yield from p
            else:
                b, idx = p
                for lo in range(0, len(idx), self.BLOCK):
                    yield from b.build(idx[lo:lo + self.BLOCK])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.materialise()[i]
        n = len(self)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("row index out of range")
        for p in self._parts:
            m = len(p) if isinstance(p, list) else len(p[1])
            if i < m:
                return p[i] if isinstance(p, list) else p[0].build(p[1][i:i + 1])[0]
            i -= m
        raise IndexError("row index out of range")

    def extend(self, other):
        if isinstance(other, LazyRows):
            for p in other._parts:
                if (self._parts and not isinstance(p, list) and not isinstance(self._parts[-1], list) and self._parts[-1][0] is p[0]):
                    self._parts[-1] = (p[0], np.concatenate([self._parts[-1][1], p[1]]))
                else:
                    self._parts.append(p if not isinstance(p, list) else list(p))
        else:
            rows = list(other)
            if rows:
                if self._parts and isinstance(self._parts[-1], list):
                    self._parts[-1].extend(rows)
                else:
                    self._parts.append(rows)

    def __iadd__(self, other):
        self.extend(other)
        return self

    # ---- the rest of the list protocol (advisor, r05): anything that changes single rows turns the sequence into plain rows
    def __reduce__(self):
        """pickles as a plain list (main_ctrl hands results[chrom] to Pool.starmap_async(generate_output, ...), which pickles
        it; the backing holds ctypes pointers that cannot travel)"""
        return (list, (self.materialise(),))

    def _plain(self):
        if not (len(self._parts) == 1 and isinstance(self._parts[0], list)):
            self._parts = [self.materialise()]
        return self._parts[0]

    def append(self, row):
        if self._parts and isinstance(self._parts[-1], list):
            self._parts[-1].append(row)
        else:
            self._parts.append([row])

    def insert(self, i, row):
        self._plain().insert(i, row)

    def pop(self, i=-1):
        return self._plain().pop(i)

    def remove(self, row):
        self._plain().remove(row)

    def reverse(self):
        self._plain().reverse()

    def clear(self):
        self._parts = []

    def copy(self):
        out = LazyRows()
        out._parts = [list(p) if isinstance(p, list) else (p[0], p[1].copy()) for p in self._parts]
        return out

    def index(self, row, *a):
        return self.materialise().index(row, *a)

    def count(self, row):
        return self.materialise().count(row)

    def __contains__(self, row):
        return row in self.materialise()

    def __setitem__(self, i, row):
        self._plain()[i] = row

    def __delitem__(self, i):
        del self._plain()[i]

    def __add__(self, other):
        out = self.copy()
        out.extend(other)
        return out

    def __radd__(self, other):
        out = LazyRows()
        out.extend(other)
        out.extend(self)
        return out

    def __bool__(self):
        return len(self) > 0

    def __eq__(self, other):
        return self.materialise() == (other.materialise() if isinstance(other, LazyRows) else other)

    def __repr__(self):
        return "LazyRows(%d rows, %d part(s))" % (len(self), len(self._parts))

    def materialise(self):
        """the plain list of rows"""
        out = []
        for p in self._parts:
            out.extend(p if isinstance(p, list) else p[0].build(p[1]))
        return out

    def backing(self):
        """the (store, segments, result) block behind the sequence when it has exactly one (else None)"""
        bs = {id(p[0]): p[0] for p in self._parts if not isinstance(p, list)}
        return next(iter(bs.values())) if len(bs) == 1 and not any(isinstance(p, list) for p in self._parts) else None

    def call_indices(self):
        """call indices of the rows, in order (single-backing sequences)"""
        return np.concatenate([p[1] for p in self._parts]) if self._parts else np.zeros(0, np.int64)

    def _keys(self):
        ks = [np.array([int(r[2]) for r in p], np.int64) if isinstance(p, list) else p[0].key[p[1]] for p in self._parts]
        return np.concatenate(ks) if ks else np.zeros(0, np.int64)

    def sort(self, key=None, reverse=False):
        """list.sort.  `key=rows.BY_POS` (or sort_by_pos()) is answered from the bp1 column: a stable argsort of integers, no
        row is built.  Any other key that is `int(row[2])` on a sample of rows - the reference's own lambda (GT:252) - takes the
        same path AFTER the key has been checked against the column on every row it can be evaluated on cheaply: the sample
        first, and, when the rows exist anyway (plain parts), all of them; a key that disagrees anywhere sorts the materialised
        rows with list.sort (advisor, r05: a sampled probe alone could silently replace a key that differs elsewhere)."""
        n = len(self)
        if n < 2:
            return
        vector = key is BY_POS
        if not vector and key is not None:
            vector = True
            step = max(1, n // 64)
            try:
                for i in range(0, n, step):
                    r = self[i]
                    kv = key(r)
                    if type(kv) is not int or kv != int(r[2]):
                        vector = False
                        break
                if vector:                    # the key as a function of the row's fields: a row the sample did not hold cannot
                    probe = list(self[0])     # differ unless the key reads other fields - try it on a row whose other fields moved
                    for f in range(len(probe)):
                        if f != 2:
                            probe[f] = "0"
                    vector = key(probe) == int(probe[2])
            except Exception:                 # noqa: BLE001  (a key this probe cannot evaluate: let list.sort raise what it raises)
                vector = False
        if not vector:
            rows = self.materialise()
            rows.sort(key=key, reverse=reverse)
            self._parts = [rows]
            return
        k = self._keys()
        order = np.argsort(-k if reverse else k, kind="stable")
        # the permutation, regrouped into runs of the same part
        part_of = np.concatenate([np.full(len(p) if isinstance(p, list) else len(p[1]), j, np.int64) for j, p in enumerate(self._parts)])
        local = np.concatenate([np.arange(len(p) if isinstance(p, list) else len(p[1]), dtype=np.int64) for p in self._parts])
        po, lo = part_of[order], local[order]
        cuts = np.flatnonzero(np.r_[True, po[1:] != po[:-1], True])
        new = []
        for a0, a1 in zip(cuts[:-1].tolist(), cuts[1:].tolist()):
            p = self._parts[int(po[a0])]
            sel = lo[a0:a1]
            new.append([p[int(q)] for q in sel] if isinstance(p, list) else (p[0], p[1][sel]))
        self._parts = new


    def sort_by_pos(self, reverse=False):
        """the reference's `semi_result.sort(key=lambda x: int(x[2]))` (GT:252), explicitly"""
        self.sort(key=BY_POS, reverse=reverse)


def BY_POS(row):
    """sort key of the reference's VCF writer: int(row[2]) (cuteSV_genotype.py:252); LazyRows.sort recognises it by identity"""
    return int(row[2])


def lazy_rows_by_segment(store, segments, res, ctx=None):
    """-> (RowsBacking, list (per segment) of call-index ranges [lo, hi)): the lazy counterpart of rows_by_segment.
    ctx: the engine.Context `res` came from with reuse=True (its arrays are then kept, not copied)."""
    b = RowsBacking(store, segments, res, ctx)
    n = b.res.n_calls
    cut = np.searchsorted(b.res.arrays["call_seg"][:n], np.arange(len(b.segments) + 1)).tolist()
    return b, [(cut[k], cut[k + 1]) for k in range(len(b.segments))]


# ------------------------------------------------------------------------------------------------ the plain statement
def _ci(v):
    return "-%d,%d" % (v, v)                          # cal_CIPOS, cuteSV_genotype.py:60


def materialise_py(store, segments, res):
    """res: dict of trimmed result arrays (HostResult.trimmed()) -> list of (segment index, row), one Python loop."""
    n = len(res["bp1"])
    if n == 0:
        return []
    sup_off = res["support_off"]
    sup_sig = res["support_sig"]
    # one vectorised gather for all read names
    names = store.names.take(store.read_id[sup_sig]) if len(sup_sig) else []
    call_seg = res["call_seg"].tolist()
    bp1 = res["bp1"].tolist(); bp2 = res["bp2"].tolist(); support = res["support"].tolist()
    cipos = res["cipos"].tolist(); cilen = res["cilen"].tolist()
    seq_pick = res["seq_pick"].tolist(); aux = res["call_aux"].tolist()
    dr = res["dr"].tolist(); gl = res["gl_idx"].tolist()
    off = sup_off.tolist()
    seg_type = segments["svtype"].tolist()
    seg_chrom = segments["chrom"].tolist()
    seg_gt = segments["genotype"].tolist()
    out = []
    for c in range(n):
        k = call_seg[c]
        t = seg_type[k]
        chrom = store.chroms[seg_chrom[k]]
        reads = ",".join(names[off[c]:off[c + 1]])
        if seg_gt[k] and gl[c] >= 0:          # TRA: count_coverage may give up -> '.' fields (cuteSV_resolveTRA.py:276-281)
            gt, pl, gq, qual = gl_fields(gl[c])
            g = (str(dr[c]), gt, pl, gq, qual)
        else:
            g = (".", "./.", ".,.,.", ".", ".")
        if t == _abi.DEL:
            row = [chrom, "DEL", str(bp1[c]), str(-bp2[c]), str(support[c]), _ci(cipos[c]), _ci(cilen[c]),
                   g[0], g[1], g[2], g[3], g[4], reads]
        elif t == _abi.INS:
            row = [chrom, "INS", str(bp1[c]), str(bp2[c]), str(support[c]), _ci(cipos[c]), _ci(cilen[c]),
                   g[0], g[1], g[2], g[3], g[4], reads, store.sequence(seq_pick[c])[:bp2[c]]]
        elif t == _abi.DUP:
            row = [chrom, "DUP", str(bp1[c]), str(bp2[c] - bp1[c]), str(support[c]), g[0], g[1], g[2], g[3], g[4], reads]
        elif t == _abi.INV:
            row = [chrom, "INV", str(bp1[c]), str(bp2[c] - bp1[c]), str(support[c]), g[0], g[1],
                   store.strands[aux[c]], g[2], g[3], g[4], reads]
        else:
            code = aux[c] & 7
            chr2 = store.chroms[aux[c] >> 3]
            mate = bp2[c] + (1 if code in (0, 2) else 0)              # types A/C, cuteSV_resolveTRA.py:140
            row = [chrom, _TRA_ALT[code] % ("%s:%s" % (chr2, mate)), str(bp1[c]), chr2, str(bp2[c]), str(support[c]),
                   g[0], g[1], g[2], g[3], g[4], reads]
        out.append((k, row))
    return out
