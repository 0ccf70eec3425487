"""Flat columnar signature store — the input side of the hot path.

The reference keeps signatures as pickled lists of Python tuples, one list per (SV type,
chromosome), in `<work_dir>/<TYPE>.pickle` at the byte offsets of `sigindex.pickle`
(cuteSV main script :817-857, 1092-1093).  This module holds the same information as flat
little-endian columns, already in the layout of `csv_batch_in` (include/cutesv_hip.h), so a
whole genome goes to the GPU in one H2D copy:

    a:int64[N]  b:int64[N]  read_id:int32[N]  aux:int32[N]     + segment table (type, chrom, begin, end)
    reads: r_start:int64[R] r_end:int64[R] r_primary:u8[R] r_id:int32[R] + reads_off[n_chrom+1]

Order contract: inside a segment, rows are in the order the reference's
rebuild step leaves them (main script :764-802 sort keys, :958-969 adjacent de-duplication);
read ids are interned so that id order == Python string order of the names, which makes the
reference's name tie-break an integer comparison.  The reads block of a chromosome is sorted
by start here (the reference sorts its sweep-line events itself, cuteSV_genotype.py:109).
"""
import json
import os
import pickle
from dataclasses import dataclass, field, asdict

import numpy as np

from . import _abi

TYPES = ("DEL", "INS", "INV", "DUP", "TRA")      # the reference's submission order (main script :1116-1189)
BND_CODE = {"A": 0, "B": 1, "C": 2, "D": 3}
BND_NAME = "ABCD"


@dataclass
class Params:
    """The hot-path flags of cuteSV_Description.py:53-263 with their defaults."""
    min_support: int = 10
    min_size: int = 30
    max_size: int = 100000
    genotype: bool = False
    genotype_tra: bool = False        # with genotype: TRA calls are genotyped from the reads table
    gt_round: int = 500
    max_cluster_bias_INS: int = 100
    diff_ratio_merging_INS: float = 0.3
    max_cluster_bias_DEL: int = 200
    diff_ratio_merging_DEL: float = 0.5
    max_cluster_bias_INV: int = 500
    max_cluster_bias_DUP: int = 500
    max_cluster_bias_TRA: int = 50
    diff_ratio_filtering_TRA: float = 0.6
    remain_reads_ratio: float = 1.0

    @classmethod
    def ont(cls, **kw):          # README preset: INS 100/0.3, DEL 100/0.3
        return cls(max_cluster_bias_INS=100, diff_ratio_merging_INS=0.3,
                   max_cluster_bias_DEL=100, diff_ratio_merging_DEL=0.3, **kw)

    @classmethod
    def hifi(cls, **kw):         # INS 1000/0.9, DEL 1000/0.5
        return cls(max_cluster_bias_INS=1000, diff_ratio_merging_INS=0.9,
                   max_cluster_bias_DEL=1000, diff_ratio_merging_DEL=0.5, **kw)

    @classmethod
    def clr(cls, **kw):          # INS 100/0.3, DEL 200/0.5 (== defaults)
        return cls(max_cluster_bias_INS=100, diff_ratio_merging_INS=0.3,
                   max_cluster_bias_DEL=200, diff_ratio_merging_DEL=0.5, **kw)


class SpanList:
    """A list of strings kept as (offset, length) spans of UTF-8 bytes inside one buffer - a task's pickle file, mapped, or the
    blob `_cols_native.span_intern` returns - and decoded only when somebody asks for an element: a task's rows mention a fifth
    of its read names and a handful of its inserted sequences (SigStore.from_task_pickles)."""

    def __init__(self, buf, off, ln):
        self.buf = buf
        self.off = np.ascontiguousarray(off, np.int64)
        self.len = np.ascontiguousarray(ln, np.int32)
        self._mv = memoryview(buf)

    def __len__(self):
        return int(self.off.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        i = int(i)
        if i < 0:
            i += len(self)
        o = int(self.off[i])
        return bytes(self._mv[o:o + int(self.len[i])]).decode("utf-8", "surrogatepass")      # (as pickle decodes its own strings)

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def join(self, picks, clips=None):
        """(b"".join(self[p][:clip].encode() ...), bytes taken per pick) in C"""
        from . import _cols_native as cn
        picks = np.ascontiguousarray(picks, np.int64)
        took = np.empty(len(picks), np.int64)
        blob = cn.span_join(self.buf, self.off, self.len, picks, None if clips is None else np.ascontiguousarray(clips, np.int64), took)
        return blob, took


class NameTable:
    """read_id -> read name.  Either an explicit list (ids index it; a SpanList stays one) or a fixed-width
    synthetic scheme whose string order equals id order."""

    def __init__(self, names=None, fmt="r%09d"):
        self.names = None if names is None else (names if isinstance(names, SpanList) else list(names))
        self.fmt = fmt

    def __getitem__(self, i):
        return self.names[int(i)] if self.names is not None else self.fmt % int(i)

    def take(self, ids):
        if self.names is not None:
            n = self.names
            return [n[int(i)] for i in ids]
        f = self.fmt
        return [f % int(i) for i in ids]

    def __len__(self):
        return len(self.names) if self.names is not None else 0


def intern_names(*name_lists):
    """Rank-preserving interning: returns (sorted unique names, [id arrays...])."""
    uniq = sorted(set().union(*[set(x) for x in name_lists]))
    rank = {n: i for i, n in enumerate(uniq)}
    return uniq, [np.fromiter((rank[n] for n in lst), dtype=np.int32, count=len(lst)) for lst in name_lists]


def _sparse_blob(table, picks, n=None):
    """(blob, offsets[n + 1]) holding only table[i] for i in picks (a list, or a dict keyed by index); every other entry empty"""
    n = len(table) if n is None else n
    picks = np.asarray(picks, np.int64)
    picks = np.unique(picks[picks >= 0])
    if isinstance(table, SpanList):
        blob, took = table.join(picks)
        lens = np.zeros(n, np.int64)
        lens[picks] = took
        off = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        return blob, off
    if isinstance(table, dict):
        picks = picks[np.fromiter((int(i) in table for i in picks.tolist()), bool, len(picks))] if len(picks) else picks
    from . import _cols_native as cn                     # (built by the same make as the library; no Python fallback)
    took = np.empty(len(picks), np.int64)
    blob = cn.clip_join(table, np.ascontiguousarray(picks), np.full(len(picks), np.iinfo(np.int64).max, np.int64), took)
    lens = np.zeros(n, np.int64)
    lens[picks] = took
    off = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    return blob, off


class WalkedReads:
    """A chromosome's reads block (main script :733: (start, end, is_primary, read, chr) per row) as `pickle_table` leaves it:
    three integer columns and the (offset, length) spans of the two strings inside the mapped `reads.pickle`.  Flat arrays only,
    so that one worker's walk can be handed to the others through shared memory (`write_into` / `from_buffer`).  What is handed
    over is no wider than it has to be: the primary flag as a byte (every consumer casts it to one), and the chromosome's span
    ONCE when every row carries the same one - the block of a chromosome, as the main script writes it: 29 bytes a row, not 48."""
    FIELDS = (("start", np.int64), ("end", np.int64), ("primary", np.uint8), ("name_off", np.int64), ("name_len", np.int32),
              ("chr_off", np.int64), ("chr_len", np.int32))
    PER_BLOCK = ("chr_off", "chr_len")                        # one row instead of n when `one_chr`

    def __init__(self, n, one_chr=False, **cols):
        self.n = int(n)
        self.one_chr = bool(one_chr)
        for k, _ in self.FIELDS:
            setattr(self, k, cols[k])

    def rows_of(self, k):
        return min(self.n, 1) if (self.one_chr and k in self.PER_BLOCK) else self.n

    @classmethod
    def from_table(cls, rt):
        ints = [np.frombuffer(x, np.int64) for x in rt[2]]
        sp = [(np.frombuffer(o, np.int64), np.frombuffer(l, np.int32)) for o, l in rt[3]]
        n = int(rt[0])
        co, cl = sp[1]
        one = n > 0 and bool((co == co[0]).all()) and bool((cl == cl[0]).all())       # (a memo reference per row: the same bytes)
        if one:
            co, cl = co[:1], cl[:1]
        return cls(n, one, start=ints[0], end=ints[1], primary=ints[2].astype(np.uint8), name_off=sp[0][0], name_len=sp[0][1], chr_off=co, chr_len=cl)

    def chr_spans(self, sel=None):
        """(offsets, lengths) of the chromosome field, a span per row (of the rows `sel`)"""
        if self.one_chr:
            m = self.n if sel is None else len(sel)
            return np.repeat(self.chr_off, m), np.repeat(self.chr_len, m)
        return (self.chr_off, self.chr_len) if sel is None else (self.chr_off.take(sel), self.chr_len.take(sel))

    def nbytes(self):
        return 64 + sum((self.rows_of(k) * np.dtype(dt).itemsize + 63) // 64 * 64 for k, dt in self.FIELDS)

    def write_into(self, buf):
        """lay the columns out in `buf` (a writable buffer of nbytes()): [n, one_chr, 0...] then the columns, 64-byte aligned"""
        np.frombuffer(buf, np.int64, 8)[:] = [self.n, int(self.one_chr), 0, 0, 0, 0, 0, 0]
        off = 64
        for k, dt in self.FIELDS:
            m = self.rows_of(k)
            np.frombuffer(buf, dt, m, off)[:] = getattr(self, k)
            off += (m * np.dtype(dt).itemsize + 63) // 64 * 64

    def write_fd(self, fd):
        """the same layout written into a file of nbytes() (a fresh memfd: the gaps between the columns stay holes, i.e. zeros) with
        one pwrite per column: the kernel copies straight into new pages, where a store through a mapping first takes a fault per
        4 KB page and zeroes it (8 blocks of a HiFi genome, 89 MB: 39 ms against 90)"""
        import os
        os.pwrite(fd, np.array([self.n, int(self.one_chr), 0, 0, 0, 0, 0, 0], np.int64), 0)
        off = 64
        for k, dt in self.FIELDS:
            m = self.rows_of(k)
            if m:
                os.pwrite(fd, memoryview(np.ascontiguousarray(getattr(self, k), dt)).cast("B"), off)
            off += (m * np.dtype(dt).itemsize + 63) // 64 * 64

    @classmethod
    def from_buffer(cls, buf):
        n, one = (int(x) for x in np.frombuffer(buf, np.int64, 2))
        self = cls.__new__(cls)
        self.n, self.one_chr = n, bool(one)
        off = 64
        for k, dt in cls.FIELDS:
            m = self.rows_of(k)
            setattr(self, k, np.frombuffer(buf, dt, m, off))
            off += (m * np.dtype(dt).itemsize + 63) // 64 * 64
        return self


def _reads_near(pos1, pos2, r_start, r_end, margin, shift=10):
    """Which reads of a chromosome's block can matter to ONE task's genotyping?  A call's window is [p - h, p + h] around a
    point p that lies inside the span of its cluster's positions (a member's position for DEL / INS - INDEL:177, 399-403, 450-451 -
    a mean of member positions for DUP / INV, for either coordinate - DUP:99-109, 146-151; INV:129-130, 218-221) with h <=
    `margin`; neighbouring members of a cluster are at most max_cluster_bias <= margin apart (the chain and sub-cluster rules),
    so every window lies inside the union of [x - margin, x + margin] over the task's signature coordinates x.  A read that COVERS
    a window (start <= L and end >= R, GT:95-159) contains it, hence intersects that union.  The union is kept as a flag per
    1024-bp bin; a read stays iff a flagged bin lies in [start, end].  Everything else in the block - 90 % and more of a genome's
    reads for a 30x call set - is never interned, never uploaded and never sorted.  -> bool mask, or None (keep all).
    One pass over the signatures and one over the block in C (`_cols_native.reads_near`; the numpy statement of the same rule is
    kept by tests/test_host_logic.py)."""
    from . import _cols_native as cn
    i64 = lambda x: np.ascontiguousarray(x, np.int64)          # noqa: E731
    m = cn.reads_near(i64(pos1), None if pos2 is None else i64(pos2), i64(r_start), i64(r_end), int(margin), int(shift))
    return None if m is None else np.frombuffer(bytearray(m), np.bool_)


@dataclass
class SigStore:
    chroms: list                                  # chromosome names; index = chrom id (also the chr2 rank for TRA)
    a: np.ndarray
    b: np.ndarray
    read_id: np.ndarray
    aux: np.ndarray
    seg_index: dict                               # (type name, chrom name) -> (begin, end)
    names: NameTable = field(default_factory=NameTable)
    ins_seq: object = None                        # None (synthetic: "ACGT" repeated to aux length) or dict sig index -> str
    strands: tuple = ("++", "--")                 # INV aux code -> strand string
    reads_off: np.ndarray = None
    r_start: np.ndarray = None
    r_end: np.ndarray = None
    r_primary: np.ndarray = None
    r_id: np.ndarray = None
    contig_len: np.ndarray = None                 # reference lengths per chromosome (TRA genotyping windows)
    narrow: dict = None                           # pinned(): int32 twins of a / b / r_start / r_end (what travels to the GPU)

    # ------------------------------------------------------------------ basic access
    @property
    def n_sig(self):
        return int(self.a.shape[0])

    @property
    def n_reads(self):
        return 0 if self.r_start is None else int(self.r_start.shape[0])

    def has_reads(self, chrom):
        if self.reads_off is None:
            return False
        c = self.chroms.index(chrom)
        return bool(self.reads_off[c + 1] > self.reads_off[c])

    def sequence(self, sig):
        """Inserted sequence of INS signature `sig` (global index)."""
        if self.ins_seq is None:
            if self.names.names is not None:
                raise KeyError("this store holds real read names but no inserted sequences: pass them to the builder "
                               "(SigStore.from_tuple_lists / rebuild.store_from_unsorted(per_type['INS']['seq']))")
            n = int(self.aux[sig])                           # synthetic workloads: 'ACGT' repeated to the aux length
            return ("ACGT" * (n // 4 + 1))[:n]
        return self.ins_seq[int(sig)]

    # ------------------------------------------------------------------ page-locked columns
    def pinned(self):
        """A store whose columns live in page-locked host memory (csv_host_alloc): csv_cluster_batch then moves them to
        the GPU by DMA straight from these pages (PCIe rate) instead of through the runtime's staging copies.  The
        natural home of the columns in a worker process: load the flat `.cols` files into it once."""
        import dataclasses
        from . import engine
        cols = {}
        for k in ("read_id", "aux", "reads_off", "r_primary", "r_id", "contig_len"):
            v = getattr(self, k)
            cols[k] = None if v is None else engine.pinned_copy(v)
        # positions and lengths: int32 twins when they fit (a genome's coordinates do) - a third less data on the link
        # (CSV_IN_SIG_I32 / CSV_IN_READS_I32); the int64 columns stay what every host-side consumer reads
        narrow = {}
        for pair in (("a", "b"), ("r_start", "r_end")):
            vs = [getattr(self, k) for k in pair]
            if vs[0] is None:
                continue
            fits = all(len(v) == 0 or (int(v.min()) >= -(1 << 31) and int(v.max()) < (1 << 31)) for v in vs)
            for k, v in zip(pair, vs):
                if fits:
                    narrow[k] = engine.pinned_copy(v.astype(np.int32))
                else:
                    cols[k] = engine.pinned_copy(v)
        if "b" in narrow and self.n_sig:
            # ... the lengths and read ids once more, interleaved {b, read_id}: what a gate-first call's device-side fetch reads
            r8 = engine.pinned_empty((self.n_sig, 2), np.int32)
            r8[:, 0] = narrow["b"]
            r8[:, 1] = self.read_id
            narrow["rows8"] = r8
        if self.r_id is not None and self.n_reads and int(self.r_id.min()) >= 0:
            # ... the read id and the primary flag in one word (the form the device keeps them in): 5 -> 4 bytes per read on the link
            idp = engine.pinned_empty(self.n_reads, np.uint32)
            np.copyto(idp, self.r_id.astype(np.uint32) | (self.r_primary.astype(np.uint32) << np.uint32(31)), casting="unsafe")
            narrow["r_idp"] = idp
        if "r_start" in narrow and self.n_reads:
            # ... the reads table's starts as 16-bit gaps and its ends as 16-bit lengths (CSV_IN_READS_DELTA16), each where the
            # column is not mostly escapes (a shuffled block; ultra-long reads)
            keep_all = bool(os.environ.get("CSV_DELTA16_ESC"))
            rd = _abi.delta16_of(narrow["r_start"], alloc=engine.pinned_empty)
            if keep_all or len(rd[1]) * 64 <= len(rd[0]):
                narrow["r_delta"] = rd
            rl = _abi.len16_of(narrow["r_start"], narrow["r_end"], alloc=engine.pinned_empty)
            if keep_all or len(rl[1]) * 64 <= len(rl[0]):
                narrow["r_len16"] = rl
        if "a" in narrow:
            # ... and the position column once more as 16-bit gaps (CSV_IN_SIG_DELTA16): half of the largest transfer of a call
            ad = _abi.delta16_of(narrow["a"], alloc=engine.pinned_empty)
            if len(ad[1]) * 64 <= len(ad[0]) or os.environ.get("CSV_DELTA16_ESC"):      # (a sparse column is mostly escapes: the column itself travels)
                narrow["a_delta"] = ad
        return dataclasses.replace(self, narrow=narrow or None, **cols)

    # ------------------------------------------------------------------ string tables for the native row / VCF emitters
    def names_blob(self, picks=None):
        """read names as csv_rows_emit takes them: (blob, offsets, n_names, prefix, width); an explicit table, or
        (None, None, 0, prefix, width) for the synthetic '<prefix>%0<width>d' scheme.  The full table is cached on the store.
        picks (read ids, any order, duplicates allowed): only THESE names are put into the blob - every other id gets an empty
        string - and nothing is cached: a task's rows name ~20 k of its 110 k reads, and encoding all of them cost more than
        the rows themselves (resolve.run_* builds one store per task)."""
        nb = getattr(self, "_names_blob", None)
        if nb is not None:
            return nb
        if self.names.names is None:
            import re
            m = re.fullmatch(r"([^%]*)%0(\d+)d", self.names.fmt)
            if not m:
                raise ValueError("synthetic read-name format %r is not '<prefix>%%0<width>d'" % self.names.fmt)
            nb = self._names_blob = (None, None, 0, m.group(1).encode(), int(m.group(2)))
            return nb
        names = self.names.names
        if picks is not None:
            return _sparse_blob(names, picks) + (len(names), None, 0)
        enc = [x.encode() for x in names]
        off = np.zeros(len(enc) + 1, np.int64)
        if enc:
            np.cumsum([len(x) for x in enc], out=off[1:])
        nb = self._names_blob = (b"".join(enc), off, len(enc), None, 0)
        return nb

    def ins_blob(self, picks=None):
        """inserted sequences by global signature index as (blob, offsets[n_sig + 1]); (None, None) when the store is
        synthetic ('ACGT' repeated to the aux length).  The full blob is cached on the store; picks (signature indices, -1
        entries ignored): only those sequences, nothing cached (see names_blob)."""
        ib = getattr(self, "_ins_blob", None)
        if ib is not None:
            return ib
        if self.ins_seq is None:
            ib = self._ins_blob = (None, None)
            return ib
        seqs = self.ins_seq
        if picks is not None:
            return _sparse_blob(seqs, picks, n=self.n_sig)
        if isinstance(seqs, dict):
            keys = sorted(seqs)
            vals = [seqs[k] for k in keys]
        else:                                                  # a list by signature index (from_task_lists: the task's own objects)
            keys, vals = list(range(len(seqs))), seqs
        lens = np.zeros(self.n_sig, np.int64)
        if keys:
            lens[np.array(keys, np.int64)] = [len(v) for v in vals]
        off = np.zeros(self.n_sig + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        ib = self._ins_blob = ("".join(vals).encode(), off)
        return ib

    # ------------------------------------------------------------------ segments / batches
    def segment(self, svtype, chrom, p: Params):
        """csv_segment record for one reference task, scalars as main script :1117-1188 passes them."""
        beg, end = self.seg_index[(svtype, chrom)]
        c = self.chroms.index(chrom)
        if svtype in ("DEL", "INS"):
            bias = p.max_cluster_bias_DEL if svtype == "DEL" else p.max_cluster_bias_INS
            ratio = p.diff_ratio_merging_DEL if svtype == "DEL" else p.diff_ratio_merging_INS
            return _abi.make_segment(svtype, c, beg, end, bias, p.min_support, diff_ratio=ratio,
                                     remain_reads_ratio=p.remain_reads_ratio,
                                     gt_bias=bias if svtype == "DEL" else 1000,
                                     min_support_reads=min(p.min_support, 5), genotype=p.genotype)
        if svtype == "INV":
            return _abi.make_segment(svtype, c, beg, end, p.max_cluster_bias_INV, p.min_support,
                                     sv_size=p.min_size, max_size=p.max_size, gt_bias=p.max_cluster_bias_INV,
                                     genotype=p.genotype)
        if svtype == "DUP":
            return _abi.make_segment(svtype, c, beg, end, p.max_cluster_bias_DUP, p.min_support,
                                     sv_size=p.min_size, max_size=p.max_size, gt_bias=p.max_cluster_bias_DUP,
                                     genotype=p.genotype)
        if svtype == "TRA":
            # TRA genotyping (cuteSV_resolveTRA.py:258-309) runs over the reads table when asked for; the
            # reference re-fetches the BAM there, see include/cutesv_hip.h
            gt = bool(p.genotype and p.genotype_tra)
            if gt and self.contig_len is None:
                raise ValueError("TRA genotyping needs the reference lengths (SigStore.contig_len)")
            return _abi.make_segment(svtype, c, beg, end, p.max_cluster_bias_TRA, p.min_support,
                                     diff_ratio=p.diff_ratio_filtering_TRA, gt_bias=p.max_cluster_bias_TRA,
                                     genotype=gt, gt_round=p.gt_round)
        raise ValueError(svtype)

    def tasks(self, types=TYPES, chroms=None):
        """(type, chrom) pairs in the reference's submission order (main script :1116-1189)."""
        out = []
        for t in types:
            for (tt, ch) in self.seg_index:
                if tt == t and (chroms is None or ch in chroms):
                    out.append((t, ch))
        return out

    def host_batch(self, tasks, p: Params):
        segs = np.array([self.segment(t, ch, p) for t, ch in tasks], dtype=_abi.SEGMENT_DTYPE)
        need_reads = bool(segs["genotype"].any()) if len(segs) else False
        kw = {}
        nw = self.narrow or {}
        if need_reads and self.reads_off is not None:
            kw = dict(reads_off=self.reads_off, r_start=nw.get("r_start", self.r_start), r_end=nw.get("r_end", self.r_end),
                      r_primary=self.r_primary, r_id=self.r_id)
            if os.environ.get("CUTESV_AMD_NO_DELTA16") is None:
                kw.update(r_delta=nw.get("r_delta"), r_len16=nw.get("r_len16"), r_idp=nw.get("r_idp"))
            if bool(((segs["svtype"] == _abi.TRA) & (segs["genotype"] != 0)).any()):
                kw["contig_len"] = self.contig_len
        return _abi.HostBatch(segs, nw.get("a", self.a), nw.get("b", self.b), self.read_id, self.aux, n_chrom=len(self.chroms),
                              a_delta=nw.get("a_delta") if os.environ.get("CUTESV_AMD_NO_DELTA16") is None else None,
                              rows8=nw.get("rows8") if os.environ.get("CUTESV_AMD_NO_ROWS8") is None else None, **kw)

    # ------------------------------------------------------------------ persistence (flat .cols directory)
    def save(self, path):
        os.makedirs(path, exist_ok=True)
        cols = dict(a=self.a, b=self.b, read_id=self.read_id, aux=self.aux)
        if self.reads_off is not None:
            cols.update(reads_off=self.reads_off, r_start=self.r_start, r_end=self.r_end,
                        r_primary=self.r_primary, r_id=self.r_id)
        if self.contig_len is not None:
            cols.update(contig_len=self.contig_len)
        for k, v in cols.items():
            np.save(os.path.join(path, k + ".npy"), v)
        # int32 twins of the positions / lengths when they fit (a genome's do): what travels to the GPU (CSV_IN_SIG_I32 /
        # CSV_IN_READS_I32) - a worker that maps the directory then neither scans the columns for their range nor narrows them
        for pair in (("a", "b"), ("r_start", "r_end")):
            vs = [cols.get(k) for k in pair]
            if vs[0] is None:
                continue
            if all(len(v) == 0 or (int(v.min()) >= -(1 << 31) and int(v.max()) < (1 << 31)) for v in vs):
                for k, v in zip(pair, vs):
                    np.save(os.path.join(path, k + "32.npy"), np.ascontiguousarray(v, np.int32))
        meta = dict(chroms=self.chroms, strands=list(self.strands),
                    seg_index=[[t, c, int(b), int(e)] for (t, c), (b, e) in self.seg_index.items()],
                    names=None if self.names.names is None else list(self.names.names), name_fmt=self.names.fmt,
                    ins_seq=None if self.ins_seq is None else {str(k): v for k, v in (self.ins_seq.items() if isinstance(self.ins_seq, dict) else enumerate(self.ins_seq))})
        with open(os.path.join(path, "sigindex.json"), "w") as f:
            json.dump(meta, f)

    @classmethod
    def load(cls, path, mmap=True):
        with open(os.path.join(path, "sigindex.json")) as f:
            meta = json.load(f)
        ld = lambda k: np.load(os.path.join(path, k + ".npy"), mmap_mode="r" if mmap else None)
        kw = {}
        if os.path.exists(os.path.join(path, "reads_off.npy")):
            kw = {k: ld(k) for k in ("reads_off", "r_start", "r_end", "r_primary", "r_id")}
        if os.path.exists(os.path.join(path, "contig_len.npy")):
            kw["contig_len"] = ld("contig_len")
        narrow = {k: ld(k + "32") for k in ("a", "b", "r_start", "r_end") if os.path.exists(os.path.join(path, k + "32.npy"))}
        if ("a" in narrow) != ("b" in narrow) or ("r_start" in narrow) != ("r_end" in narrow):
            narrow = {}
        return cls(chroms=meta["chroms"], a=ld("a"), b=ld("b"), read_id=ld("read_id"), aux=ld("aux"),
                   seg_index={(t, c): (b, e) for t, c, b, e in meta["seg_index"]},
                   names=NameTable(meta["names"], meta["name_fmt"]), strands=tuple(meta["strands"]),
                   ins_seq=None if meta["ins_seq"] is None else {int(k): v for k, v in meta["ins_seq"].items()},
                   narrow=narrow or None, **kw)

    # ------------------------------------------------------------------ conversion from the reference's layout
    @classmethod
    def from_tuple_lists(cls, per_type, reads=None, chroms=None, contig_len=None):
        """Build the flat store from the reference's in-memory representation.

        per_type: {"DEL": [(pos, len, read, "DEL", chr), ...], "INS": [(pos, len, read, seq, "INS", chr)],
                   "DUP": [(p1, p2, read, "DUP", chr)], "INV": [(strand, p1, p2, read, "INV", chr)],
                   "TRA": [(type, p1, chr2, p2, read, "TRA", chr1)]}   (main script :228-257, 520-575)
        reads:    [(start, end, is_primary, read, chr)]                  (main script :733)
        The lists are sorted and adjacent-deduplicated here exactly as the rebuild step does
        (main script :764-802, 958-969), so unsorted extraction output may be passed in.
        contig_len: {chromosome: reference length} (the BAM header), only needed to genotype TRA calls.
        """
        keys = {
            "DEL": lambda x: (x[-1], int(x[0]), x[1], x[2]),
            "INS": lambda x: (x[-1], int(x[0]), x[1], x[2], x[3]),
            "DUP": lambda x: (x[-1], int(x[0]), int(x[1]), x[2]),
            "INV": lambda x: (x[-1], x[0], int(x[1]), x[2], x[3]),
            "TRA": lambda x: (x[-1], x[2], x[0], int(x[1]), x[3], x[4], x[5]),
        }
        lists = {}
        for t in TYPES:
            lst = sorted(per_type.get(t, []), key=keys[t])
            ded = []
            for x in lst:                      # adjacent exact duplicates only (main script :958-969)
                if not ded or ded[-1] != x:
                    ded.append(x)
            lists[t] = ded
        reads = list(reads or [])
        name_pos = {"DEL": 2, "INS": 2, "DUP": 2, "INV": 3, "TRA": 4}
        all_names = set(r[3] for r in reads)
        for t in TYPES:
            all_names.update(x[name_pos[t]] for x in lists[t])
        uniq = sorted(all_names)
        rank = {n: i for i, n in enumerate(uniq)}
        if chroms is None:
            cs = set(r[4] for r in reads)
            for t in TYPES:
                cs.update(x[-1] for x in lists[t])
            cs.update(x[2] for x in lists["TRA"])
            chroms = sorted(cs)
        chrom_rank = {c: i for i, c in enumerate(chroms)}
        strands = sorted(set(x[0] for x in lists["INV"])) or ["++", "--"]
        strand_code = {s: i for i, s in enumerate(strands)}

        a, b, rid, aux, seg_index, ins_seq = [], [], [], [], {}, {}
        n = 0
        for t in TYPES:
            cur, beg = None, n
            for x in lists[t]:
                ch = x[-1]
                if ch != cur:
                    if cur is not None:
                        seg_index[(t, cur)] = (beg, n)
                    cur, beg = ch, n
                if t == "DEL":
                    a.append(int(x[0])); b.append(int(x[1])); rid.append(rank[x[2]]); aux.append(0)
                elif t == "INS":
                    a.append(int(x[0])); b.append(int(x[1])); rid.append(rank[x[2]]); aux.append(len(x[3]))
                    ins_seq[n] = x[3]
                elif t == "DUP":
                    a.append(int(x[0])); b.append(int(x[1])); rid.append(rank[x[2]]); aux.append(0)
                elif t == "INV":
                    a.append(int(x[1])); b.append(int(x[2])); rid.append(rank[x[3]]); aux.append(strand_code[x[0]])
                else:
                    code = BND_CODE.get(x[0], 4)
                    a.append(int(x[1])); b.append(int(x[3])); rid.append(rank[x[4]])
                    aux.append(chrom_rank[x[2]] * 8 + code)
                n += 1
            if cur is not None:
                seg_index[(t, cur)] = (beg, n)

        kw = {}
        if reads:
            per = {c: [] for c in chroms}
            for r in reads:
                per[r[4]].append(r)
            off, rs, re_, rp, ri = [0], [], [], [], []
            for c in chroms:
                blk = sorted(per[c], key=lambda r: r[0])     # start-sorted block (stable)
                rs += [int(r[0]) for r in blk]; re_ += [int(r[1]) for r in blk]
                rp += [int(r[2]) for r in blk]; ri += [rank[r[3]] for r in blk]
                off.append(len(rs))
            kw = dict(reads_off=np.array(off, np.int64), r_start=np.array(rs, np.int64), r_end=np.array(re_, np.int64),
                      r_primary=np.array(rp, np.uint8), r_id=np.array(ri, np.int32))
        if contig_len is not None:
            kw["contig_len"] = np.array([int(contig_len[c]) for c in chroms], np.int64)
        return cls(chroms=list(chroms), a=np.array(a, np.int64), b=np.array(b, np.int64),
                   read_id=np.array(rid, np.int32), aux=np.array(aux, np.int32), seg_index=seg_index,
                   names=NameTable(uniq), ins_seq=ins_seq, strands=tuple(strands), **kw)

    @classmethod
    def from_task_lists(cls, svtype, chrom, sigs, reads=None, chroms=None):
        """One reference task - the list `pickle.load` returns at sigs_index[svtype][chrom] (already in the rebuild order
        and de-duplicated: main script :764-802, :958-969 wrote it) and, when the task genotypes, its chromosome's reads list
        - as a flat store.  What a pool worker pays per task in the drop-in (resolve.run_*): ONE pass over the tuples in C
        (`_cols_native.walk`, csrc/cols_py.cpp: integer fields - int() of an x.5 float included, main script :228 - straight
        into the column buffers, names numbered by first appearance, len() of the inserted sequences).  (r03 did it with list
        comprehensions and numpy.fromiter: ~50 ms for the 110 862 signatures of INS chr2.)"""
        from . import _cols_native as cn                     # (built by the same make as the library; no Python fallback)
        n = len(sigs)
        reads = reads or []
        nr = len(reads)
        # The list is already in the rebuild's order, so the ids only have to tell reads apart: numbered by first appearance
        # (`from_tuple_lists`, which sorts rows BY name, interns rank-preservingly).
        a, b = np.empty(n, np.int64), np.empty(n, np.int64)
        rid, aux = np.empty(n, np.int32), np.zeros(n, np.int32)
        ins_seq, strands = [], ("++", "--")
        name_col = {"DEL": 2, "INS": 2, "DUP": 2, "INV": 3, "TRA": 4}[svtype]
        if svtype in ("DEL", "DUP"):
            cn.walk(sigs, ((0, a), (1, b)), (), ())
        elif svtype == "INS":
            cn.walk(sigs, ((0, a), (1, b)), (), ((3, aux),))
            ins_seq = cn.column(sigs, 3)                      # (the objects of the task list, shared: SigStore.sequence indexes it)
        elif svtype == "INV":
            sd = {}
            cn.walk(sigs, ((1, a), (2, b)), ((0, aux, sd),), ())
            strands = tuple(sorted(sd)) or ("++", "--")
            if n:
                aux = np.array([strands.index(s_) for s_ in sd], np.int32)[aux]     # first-appearance id -> rank of the strand string
        else:                                                 # TRA: (bnd type, pos1, chr2, pos2, read, "TRA", chr1)
            td, cd = {}, {}
            t_id, c_id = np.empty(n, np.int32), np.empty(n, np.int32)
            cn.walk(sigs, ((1, a), (3, b)), ((0, t_id, td), (2, c_id, cd)), ())
        rd = {}
        r_chr = np.empty(nr, np.int32)
        r_start, r_end = np.empty(nr, np.int64), np.empty(nr, np.int64)
        r_primary, r_id = np.empty(nr, np.uint8), np.empty(nr, np.int32)
        if nr:                                                # (start, end, is_primary, read, chr), main script :733
            cn.walk(reads, ((0, r_start), (1, r_end), (2, r_primary)), ((4, r_chr, rd),), ())
        # read names of the signatures, then of the reads table, in ONE id space (first appearance)
        uniq = cn.intern(((sigs, name_col, rid), (reads, 3, r_id)) if nr else ((sigs, name_col, rid),))
        if chroms is None:
            cs = {chrom}
            if svtype == "TRA":
                cs.update(cd)
            cs.update(rd)
            chroms = sorted(cs)
        crank = {c: i for i, c in enumerate(chroms)}
        if svtype == "TRA" and n:
            clut = np.array([crank[c] for c in cd], np.int32)
            tlut = np.array([BND_CODE.get(t, 4) for t in td], np.int32)
            aux = clut[c_id] * 8 + tlut[t_id]
        kw = {}
        if nr:
            rc = np.array([crank[c] for c in rd], np.int64)[r_chr]
            o = np.argsort(rc, kind="stable")                # blocks by chromosome; the device orders every block by start
            kw = dict(reads_off=np.searchsorted(rc[o], np.arange(len(chroms) + 1)).astype(np.int64),
                      r_start=r_start[o], r_end=r_end[o], r_primary=r_primary[o], r_id=r_id[o])
        return cls(chroms=list(chroms), a=a, b=b, read_id=rid, aux=aux, seg_index={(svtype, chrom): (0, n)} if n else {},
                   names=NameTable(uniq), ins_seq=ins_seq if svtype == "INS" else {}, strands=strands, **kw)

    @classmethod
    def from_task_pickles(cls, svtype, chrom, sig_buf, sig_off, reads_buf=None, reads_off=None, gt_margin=None, reads_cache=None, reads_key=None,
                          sig_end=None, reads_end=None):
        """from_task_lists without the lists: the task's pickle (and its chromosome's reads pickle) walked in C straight out of
        the mapped files (`_cols_native.pickle_table`): integer fields into the columns, strings as spans of the file - the
        read names interned by their bytes, the inserted sequences never touched unless a call picks one.  The 110 862
        signatures of INS chr2: ~6 ms instead of 21-24 ms of pickle.load + 5 ms over its objects.  Returns None when the
        stream holds anything pickle_table does not know (the caller unpickles then).
        gt_margin (the task's genotyping half-window: max_cluster_bias, 1000 for INS - INDEL:450-451, DUP:146-151, INV:218-221):
        only the reads that can cover a window of THIS task are kept - see _reads_near.
        reads_cache / reads_key: the walked form of a chromosome's reads block is shared between the tasks of the chromosome - DEL,
        INS, INV, DUP each walk the same block in the reference (INDEL:445-448) - through anything that offers
        `reads_get(key) -> WalkedReads | None` and `reads_put(key, WalkedReads)` (broker.Client: the GPU's broker keeps the blocks
        in shared memory for the pool's workers).
        sig_end / reads_end: where the block ends in its file, if the caller knows (the next offset of `sigindex.pickle`): the walker
        then sizes its columns in a few steps instead of doubling them (a hint - a wrong one costs memory or time, never a row)."""
        from . import _cols_native as cn                     # (built by the same make as the library; no Python fallback)
        ints, strs, width = {"DEL": ((0, 1), (2,), 5), "DUP": ((0, 1), (2,), 5), "INS": ((0, 1), (2, 3), 6),
                             "INV": ((1, 2), (0, 3), 6), "TRA": ((1, 3), (0, 2, 4), 7)}[svtype]
        t = cn.pickle_table(sig_buf, int(sig_off), width, ints, strs, -1 if sig_end is None else int(sig_end))
        if t is None:
            return None
        n = int(t[0])
        a, b = (np.frombuffer(x, np.int64).copy() for x in t[2])     # (writable, like every other store's columns)
        spans = [(np.frombuffer(o, np.int64), np.frombuffer(l, np.int32)) for o, l in t[3]]
        wr = None                                             # the chromosome's reads block, walked (WalkedReads)
        nr = 0
        if reads_buf is not None:
            if reads_cache is not None and reads_key is not None:
                wr = reads_cache.reads_get(reads_key)
            if wr is None:
                rt = cn.pickle_table(reads_buf, int(reads_off), 5, (0, 1, 2), (3, 4),      # (start, end, is_primary, read, chr), main script :733
                                     -1 if reads_end is None else int(reads_end))
                if rt is None:
                    return None
                wr = WalkedReads.from_table(rt)
                if reads_cache is not None and reads_key is not None and wr.n:
                    reads_cache.reads_put(reads_key, wr)
            nr = wr.n
            n_block = wr.n

        def small(buf, sp):                                   # a field with a handful of distinct values -> (values, ids)
            ids = np.empty(len(sp[0]), np.int32)
            blob, uo, ul = cn.span_intern(((buf, sp[0], sp[1], ids),))
            return list(SpanList(blob, np.frombuffer(uo, np.int64), np.frombuffer(ul, np.int32))), ids

        aux = np.zeros(n, np.int32)
        ins_seq, strands = {}, ("++", "--")
        name_span = spans[{"DEL": 0, "DUP": 0, "INS": 0, "INV": 1, "TRA": 2}[svtype]]
        cd = []
        if svtype == "INS":
            if t[4]:                                      # every string of the stream is ASCII (the walker validated them): len = bytes
                aux[:] = spans[1][1]
            else:
                cn.span_cplen(sig_buf, spans[1][0], spans[1][1], aux)
            ins_seq = SpanList(sig_buf, spans[1][0], spans[1][1])
        elif svtype == "INV":
            sd, sid = small(sig_buf, spans[0])
            strands = tuple(sorted(sd)) or ("++", "--")
            if n:
                aux = np.array([strands.index(s_) for s_ in sd], np.int32)[sid]
        elif svtype == "TRA":
            td, t_id = small(sig_buf, spans[0])
            cd, c_id = small(sig_buf, spans[1])
        rid = np.empty(n, np.int32)
        spec = [(sig_buf, name_span[0], name_span[1], rid)]
        kw = {}
        rd = []
        if nr:
            r_start, r_end, r_primary = wr.start, wr.end, wr.primary
            r_name = (wr.name_off, wr.name_len)
            sel = None
            if gt_margin is not None and n:
                keep = _reads_near(a, b if svtype in ("DUP", "INV") else None, r_start, r_end, int(gt_margin))
                if keep is not None:
                    if not keep.any():
                        keep[0] = True                      # (a block with reads stays a block with reads: DR = 0, not "no reads block")
                    sel = np.flatnonzero(keep)                # (one index list, a take per column: a boolean mask is re-scanned by every column)
                    r_start, r_end, r_primary = r_start.take(sel), r_end.take(sel), r_primary.take(sel)
                    r_name = (r_name[0].take(sel), r_name[1].take(sel))
                    nr = len(r_start)
            r_id = np.empty(nr, np.int32)
            spec.append((reads_buf, r_name[0], r_name[1], r_id))
            if wr.one_chr:                                    # (the block of one chromosome: its name read once)
                o0, l0 = int(wr.chr_off[0]), int(wr.chr_len[0])
                rd, r_chr = [bytes(memoryview(reads_buf)[o0:o0 + l0]).decode("utf-8", "surrogatepass")], None
            else:
                rd, r_chr = small(reads_buf, wr.chr_spans(sel))
        blob, uo, ul = cn.span_intern(tuple(spec))          # read names of the signatures, then of the reads, ONE id space
        uniq = SpanList(blob, np.frombuffer(uo, np.int64), np.frombuffer(ul, np.int32))
        cs = {chrom}
        cs.update(cd)
        cs.update(rd)
        chroms = sorted(cs)
        crank = {c: i for i, c in enumerate(chroms)}
        if svtype == "TRA" and n:
            clut = np.array([crank[c] for c in cd], np.int32)
            tlut = np.array([BND_CODE.get(t_, 4) for t_ in td], np.int32)
            aux = clut[c_id] * 8 + tlut[t_id]
        if nr:
            if len(rd) == 1:                                # (the usual block: one chromosome - nothing to regroup)
                kw = dict(reads_off=np.array([0, nr] if crank[rd[0]] == 0 else [0] * (crank[rd[0]] + 1) + [nr] * (len(chroms) - crank[rd[0]]), np.int64),
                          r_start=r_start, r_end=r_end, r_primary=r_primary.astype(np.uint8), r_id=r_id)
            else:
                rc = np.array([crank[c] for c in rd], np.int64)[r_chr]
                o = np.argsort(rc, kind="stable")
                kw = dict(reads_off=np.searchsorted(rc[o], np.arange(len(chroms) + 1)).astype(np.int64),
                          r_start=r_start[o], r_end=r_end[o], r_primary=r_primary[o].astype(np.uint8), r_id=r_id[o])
        elif reads_buf is not None and n_block > 0:
            # every read of the block was out of reach of the task's windows: an empty block, still a reads table (a task
            # WITHOUT a reads block loses its calls, INDEL:443-444; one whose reads cover nothing keeps them with DR = 0)
            kw = dict(reads_off=np.zeros(len(chroms) + 1, np.int64), r_start=np.zeros(0, np.int64), r_end=np.zeros(0, np.int64),
                      r_primary=np.zeros(0, np.uint8), r_id=np.zeros(0, np.int32))
        return cls(chroms=chroms, a=a, b=b, read_id=rid, aux=aux, seg_index={(svtype, chrom): (0, n)} if n else {},
                   names=NameTable(uniq), ins_seq=ins_seq if svtype == "INS" else {}, strands=strands, **kw)

    @classmethod
    def from_reference_workdir(cls, work_dir, sigs_index=None, contig_len=None):
        """Read the reference's own `<TYPE>.pickle` / `reads.pickle` / `sigindex.pickle` files
        (main script :817-857, 1092-1093) into the flat layout.  Needs only `pickle`."""
        if not work_dir.endswith("/"):
            work_dir += "/"
        if sigs_index is None:
            with open(work_dir + "sigindex.pickle", "rb") as f:
                sigs_index = pickle.load(f)
        per_type, reads = {}, []
        for t in TYPES:
            per_type[t] = []
            for ch, off in sigs_index.get(t, {}).items():
                with open("%s%s.pickle" % (work_dir, t), "rb") as f:
                    f.seek(off)
                    per_type[t].extend(pickle.load(f))
        for ch, off in sigs_index.get("reads", {}).items():
            with open(work_dir + "reads.pickle", "rb") as f:
                f.seek(off)
                reads.extend(pickle.load(f))
        return cls.from_tuple_lists(per_type, reads, contig_len=contig_len)

    @classmethod
    def from_sigs_dir(cls, work_dir, contig_len=None):
        """Read the legacy text signature files the reference writes under --write_old_sigs (main script :766-816,
        cuteSV_Description.py:96-98): `<TYPE>.sigs` with the tab-separated columns
            DEL / DUP   TYPE chr pos|pos1 len|pos2 read          INS   TYPE chr pos len read seq
            INV         TYPE chr strand pos1 pos2 read           TRA   TYPE chr1 bnd_type pos1 chr2 pos2 read
        and `reads.sigs` (chr start end is_primary read).  Positions were written with %d, so a split-read INS position
        x.5 arrives truncated - which is what every consumer of the tuple does with it anyway (int(pos), INDEL:271)."""
        per_type, reads = {t: [] for t in TYPES}, []
        for t in TYPES:
            path = os.path.join(work_dir, t + ".sigs")
            if not os.path.exists(path):
                continue
            with open(path) as f:
                for line in f:
                    x = line.rstrip("\n").split("\t")
                    if t in ("DEL", "DUP"):
                        per_type[t].append((int(x[2]), int(x[3]), x[4], t, x[1]))
                    elif t == "INS":
                        per_type[t].append((int(x[2]), int(x[3]), x[4], x[5], t, x[1]))
                    elif t == "INV":
                        per_type[t].append((x[2], int(x[3]), int(x[4]), x[5], t, x[1]))
                    else:
                        per_type[t].append((x[2], int(x[3]), x[4], int(x[5]), x[6], t, x[1]))
        path = os.path.join(work_dir, "reads.sigs")
        if os.path.exists(path):
            with open(path) as f:
                for line in f:
                    x = line.rstrip("\n").split("\t")
                    reads.append((int(x[1]), int(x[2]), int(x[3]), x[4], x[0]))
        return cls.from_tuple_lists(per_type, reads, contig_len=contig_len)

    # ------------------------------------------------------------------ the inverse, as files (tests / bench: the drop-in's input)
    def write_reference_workdir(self, work_dir, types=TYPES):
        """Lay the store out as the reference's own work directory (main script :817-857, 1085-1093): `<TYPE>.pickle` - one
        pickled list of the reference's tuples per chromosome at a recorded byte offset - `reads.pickle` likewise, and the index
        dict {TYPE: {chr: offset}, "reads": {...}} (also written as `sigindex.pickle`).  The inverse of from_reference_workdir,
        column-wise (no Python loop per signature: a 30x genome in seconds).  Returns the index."""
        import pickle
        if not work_dir.endswith("/"):
            work_dir += "/"
        index = {}
        for t in TYPES:
            index[t] = {}
            with open(work_dir + t + ".pickle", "wb") as f:
                if t not in types:
                    continue
                for ch in self.chroms:
                    if (t, ch) not in self.seg_index:
                        continue
                    beg, end = self.seg_index[(t, ch)]
                    if end <= beg:
                        continue
                    n = end - beg
                    a, b = self.a[beg:end].tolist(), self.b[beg:end].tolist()
                    nm = self.names.take(self.read_id[beg:end])
                    ax = self.aux[beg:end].tolist()
                    tt, cc = [t] * n, [ch] * n
                    if t in ("DEL", "DUP"):
                        blk = list(zip(a, b, nm, tt, cc))
                    elif t == "INS":
                        if self.ins_seq is None and self.names.names is None:
                            base = "ACGT" * (max(ax, default=0) // 4 + 1)
                            seqs = [base[:k] for k in ax]
                        else:
                            seqs = [self.sequence(i) for i in range(beg, end)]
                        blk = list(zip(a, b, nm, seqs, tt, cc))
                    elif t == "INV":
                        blk = list(zip([self.strands[k] for k in ax], a, b, nm, tt, cc))
                    else:
                        blk = list(zip([BND_NAME[k & 7] if (k & 7) < 4 else "X" for k in ax], a, [self.chroms[k >> 3] for k in ax], b, nm, tt, cc))
                    index[t][ch] = f.tell()
                    pickle.dump(blk, f)
        index["reads"] = {}
        with open(work_dir + "reads.pickle", "wb") as f:
            if self.reads_off is not None:
                for c, ch in enumerate(self.chroms):
                    lo, hi = int(self.reads_off[c]), int(self.reads_off[c + 1])
                    if hi <= lo:
                        continue
                    blk = list(zip(self.r_start[lo:hi].tolist(), self.r_end[lo:hi].tolist(), self.r_primary[lo:hi].tolist(),
                                   self.names.take(self.r_id[lo:hi]), [ch] * (hi - lo)))
                    index["reads"][ch] = f.tell()
                    pickle.dump(blk, f)
        with open(work_dir + "sigindex.pickle", "wb") as f:
            pickle.dump(index, f)
        return index

    # ------------------------------------------------------------------ the inverse (tests / golden generation)
    def tuple_lists(self):
        """Reference-format tuple lists per type and the reads list (inverse of from_tuple_lists)."""
        out = {t: [] for t in TYPES}
        for (t, ch), (beg, end) in self.seg_index.items():
            names = self.names.take(self.read_id[beg:end])
            for k, i in enumerate(range(beg, end)):
                a, b, nm, ax = int(self.a[i]), int(self.b[i]), names[k], int(self.aux[i])
                if t == "DEL":
                    out[t].append((a, b, nm, "DEL", ch))
                elif t == "INS":
                    out[t].append((a, b, nm, self.sequence(i), "INS", ch))
                elif t == "DUP":
                    out[t].append((a, b, nm, "DUP", ch))
                elif t == "INV":
                    out[t].append((self.strands[ax], a, b, nm, "INV", ch))
                else:
                    code = ax & 7
                    out[t].append((BND_NAME[code] if code < 4 else "X", a, self.chroms[ax >> 3], b, nm, "TRA", ch))
        reads = []
        if self.reads_off is not None:
            for c, ch in enumerate(self.chroms):
                lo, hi = int(self.reads_off[c]), int(self.reads_off[c + 1])
                names = self.names.take(self.r_id[lo:hi])
                for k, i in enumerate(range(lo, hi)):
                    reads.append((int(self.r_start[i]), int(self.r_end[i]), int(self.r_primary[i]), names[k], ch))
        return out, reads
