"""The rebuild step on the GPU: unsorted signature rows -> a `SigStore` in the order
contract of cuteSV's process_process_sigs_type (main script :750-857).

`rebuild_columns` is the thin face of `csv_rebuild_signatures` (stable LSD radix sort of a row permutation on
(segment, [aux], pos, len/pos2, read id) + adjacent de-duplication, cutesv_amd/csrc/sort.hip.h);
`store_from_unsorted` assembles the flat store from per-type unsorted columns the extraction step produced.

INS rows are special: the reference sorts them by (chr, int(pos), len, read, sequence) and removes a row only when the
WHOLE tuple repeats - including the sequence and the x.5 of a split-read position ((a + b) / 2, main script :228,
:774-775, :958-969).  The GPU sorts on the integer columns (stable, so equal keys stay in input order) and leaves INS
segments un-deduplicated; the host then finishes the few groups of INS rows that agree in (chr, int(pos), len, read):
ordered by sequence, exact duplicates dropped.  Everything else is exact on the integer columns alone."""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib
from .columns import SigStore, NameTable, TYPES


class RebuildIn(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_seg", C.c_int32), ("flags", C.c_int32), ("seg_aux_major", C.c_void_p),
                ("seg_id", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p), ("read_id", C.c_void_p), ("aux", C.c_void_p),
                ("seg_nodedup", C.c_void_p), ("read_rank", C.c_void_p), ("n_rank", C.c_int64), ("tie_order", C.c_void_p), ("tie_user", C.c_void_p)]


# csv_tie_order_fn (include/cutesv_hip.h): int (*)(void* user, int64 n_groups, const int64* group_off, const int32* src_row, int32* order, uint8* drop)
TIE_ORDER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint8))


class RebuildOut(C.Structure):
    _fields_ = [("n_out", C.c_int64), ("seg_id", C.c_void_p), ("a", C.c_void_p), ("b", C.c_void_p), ("read_id", C.c_void_p),
                ("aux", C.c_void_p), ("src_row", C.c_void_p), ("ms_device", C.c_float), ("n_passes", C.c_int32),
                ("seg_count", C.c_void_p), ("n_ins_ties", C.c_int64), ("dev_seg_id", C.c_void_p), ("dev_a", C.c_void_p), ("dev_b", C.c_void_p),
                ("dev_read_id", C.c_void_p), ("dev_aux", C.c_void_p), ("dev_src_row", C.c_void_p), ("n_tie_rows", C.c_int64), ("n_tie_dropped", C.c_int64)]


def tie_callback(seq_of_src, half_of_src):
    """csv_tie_order_fn for INS rows (include/cutesv_hip.h): every tie group - rows that agree in (segment, int(pos), len, read)
    - ordered by its sequences (Python's stable sort, like the reference's list.sort with the sequence as the last key, main
    script :774-775) and adjacent rows whose sequence AND x.5 flag are equal too dropped (:958-969).  seq_of_src(row) /
    half_of_src(row): the caller's data by input row.  Returns the ctypes callback (keep a reference while it is in use)."""
    def fn(_user, n_groups, group_off, src_row, order, drop):
        try:
            for g in range(n_groups):
                g0, g1 = group_off[g], group_off[g + 1]
                rows = list(range(g0, g1))
                seq = {i: seq_of_src(src_row[i]) for i in rows}        # (one look-up per row)
                rows.sort(key=seq.__getitem__)
                prev = None
                for pos, i in enumerate(rows):
                    order[i] = pos
                    cur = (seq[i], int(half_of_src(src_row[i])))
                    drop[i] = 1 if (prev is not None and cur == prev) else 0
                    prev = cur
            return 0
        except Exception:                      # noqa: BLE001  (an exception must not cross the C frame)
            import traceback
            traceback.print_exc()
            return 1
    return TIE_ORDER_FN(fn)


def rebuild_columns(ctx, seg_id, a, b, read_id, aux, seg_aux_major, seg_nodedup=None, keep_on_device=False, tie_order=None, src_row_out=None):
    """-> dict(seg_id, a, b, read_id, aux, src_row, ms_device, n_passes, seg_count, n_ins_ties): sorted, de-duplicated rows.
    tie_order: a `tie_callback(...)`: the tie groups of the keep-every-row segments are settled inside the call (n_ins_ties 0).
    keep_on_device: the sorted columns stay in device memory (CSV_RB_KEEP_ON_DEVICE): the dict then holds `dev` (device
    addresses of a / b / read_id / aux, valid until the context's next rebuild / extraction call) and, from the host side,
    only src_row and seg_count - 4 instead of 28 bytes per row cross PCIe.
    src_row_out: an int32 array of at least n entries to receive src_row (a page-locked one lands by DMA)."""
    L = lib()
    L.csv_rebuild_signatures.restype = C.c_int
    L.csv_rebuild_signatures.argtypes = [C.c_void_p, C.POINTER(RebuildIn), C.POINTER(RebuildOut)]
    seg_id = np.ascontiguousarray(seg_id, np.int32); a = np.ascontiguousarray(a, np.int64); b = np.ascontiguousarray(b, np.int64)
    read_id = np.ascontiguousarray(read_id, np.int32); aux = np.ascontiguousarray(aux, np.int32)
    major = np.ascontiguousarray(seg_aux_major, np.uint8)
    nodedup = None if seg_nodedup is None else np.ascontiguousarray(seg_nodedup, np.uint8)
    n = len(a)
    if src_row_out is not None and (src_row_out.dtype != np.int32 or len(src_row_out) < n or not src_row_out.flags.c_contiguous):
        raise ValueError("src_row_out must be a contiguous int32 array of at least %d entries" % n)
    if keep_on_device:                                    # (the five sorted columns stay on the device: no host arrays for them)
        o = dict(src_row=src_row_out if src_row_out is not None else np.empty(n, np.int32))
    else:
        o = dict(seg_id=np.empty(n, np.int32), a=np.empty(n, np.int64), b=np.empty(n, np.int64), read_id=np.empty(n, np.int32),
                 aux=np.empty(n, np.int32), src_row=src_row_out if src_row_out is not None else np.empty(n, np.int32))
    seg_count = np.zeros(len(major), np.int64)
    rin = RebuildIn(n=n, n_seg=len(major), flags=_abi.RB_KEEP_ON_DEVICE if keep_on_device else 0, seg_aux_major=major.ctypes.data,
                    seg_id=seg_id.ctypes.data, a=a.ctypes.data, b=b.ctypes.data, read_id=read_id.ctypes.data, aux=aux.ctypes.data,
                    seg_nodedup=None if nodedup is None else nodedup.ctypes.data,
                    tie_order=None if tie_order is None else C.cast(tie_order, C.c_void_p))
    if keep_on_device:
        rout = RebuildOut(src_row=o["src_row"].ctypes.data, seg_count=seg_count.ctypes.data)
    else:
        rout = RebuildOut(seg_id=o["seg_id"].ctypes.data, a=o["a"].ctypes.data, b=o["b"].ctypes.data, read_id=o["read_id"].ctypes.data,
                          aux=o["aux"].ctypes.data, src_row=o["src_row"].ctypes.data, seg_count=seg_count.ctypes.data)
    ctx._check(L.csv_rebuild_signatures(ctx._h, C.byref(rin), C.byref(rout)))
    k = int(rout.n_out)
    out = {key: v[:k] for key, v in o.items()} if not keep_on_device else {"src_row": o["src_row"][:k]}
    out["ms_device"] = float(rout.ms_device)
    out["n_passes"] = int(rout.n_passes)
    out["seg_count"] = seg_count
    out["n_ins_ties"] = int(rout.n_ins_ties)
    out["n_tie_rows"], out["n_tie_dropped"] = int(rout.n_tie_rows), int(rout.n_tie_dropped)
    out["n_out"] = k
    if keep_on_device:
        out["dev"] = dict(a=rout.dev_a, b=rout.dev_b, read_id=rout.dev_read_id, aux=rout.dev_aux, seg_id=rout.dev_seg_id, src_row=rout.dev_src_row)
    return out


_STAGE = (("seg", np.int32), ("a", np.int64), ("b", np.int64), ("rid", np.int32), ("aux", np.int32), ("src_row", np.int32))


_FILL_CHUNK = 1 << 19                                   # rows per copy job
_POOL = []


def _fill_pool():
    if not _POOL:
        import concurrent.futures
        import os
        _POOL.append(concurrent.futures.ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 2) // 2)), thread_name_prefix="csv-stage"))
    return _POOL[0]


def _staging(ctx, n):
    """views [:n] of the context's page-locked staging columns (grown by half when too small)"""
    from .engine import pinned_empty
    st = getattr(ctx, "_rb_stage", None)
    if st is None or st["cap"] < n:
        cap = max(1024, n + n // 2 if st is not None else n)
        st = {"cap": cap}
        for k, dt in _STAGE:
            st[k] = pinned_empty(cap, dt)
        ctx._rb_stage = st
    return {k: st[k][:n] for k, _ in _STAGE}


def rebuild_to_device_batch(ctx, chroms, per_type, params_segment, reads=None):
    """The rebuild -> cluster hand-off without a host round trip (the reference's dataflow main script :750-857 -> :1113-1199):
    unsorted per-type rows (as store_from_unsorted takes them) are sorted and de-duplicated on the device and STAY there;
    returns (batch, tasks, src_row) where `batch` is an `_abi.HostBatch.on_device` whose columns are the rebuild's device
    buffers, `tasks` the (type, chromosome) pairs of its segments in the reference's order and `src_row` the input row of every
    sorted row (to carry read names / sequences on the host).  Like the batch's device columns, `src_row` lives in the
    context's staging memory: both are valid until the context's next rebuild call (copy src_row to keep it longer).
    params_segment(svtype, chrom_index, begin, end) -> csv_segment record.
    INS rows with `seq` (and `half`): rows that tie on (chromosome, int(pos), len, read) are ordered by their sequences and
    de-duplicated on the whole tuple as the reference does - the few tie rows' indices visit the host through the library's
    tie_order callback, the columns do not (r03 raised on the first tie and sent the whole genome through host memory).
    Without `seq` the integer columns decide."""
    order = sorted(range(len(chroms)), key=lambda i: chroms[i])
    crank = np.zeros(len(chroms), np.int64)
    crank[order] = np.arange(len(chroms))
    # The five columns are written ONCE, in the ABI's widths, into page-locked staging arrays the context keeps (no per-type
    # temporaries, no concatenate pass, and the upload is a DMA at the link's rate instead of a staged copy of pageable memory:
    # the 80 MB of a 30x genome's rows took 4 of the chain's 8.6 ms)
    live = [(ti, t) for ti, t in enumerate(TYPES) if t in per_type and len(per_type[t]["a"])]
    n_rows = sum(len(per_type[t]["a"]) for _, t in live)
    cat = _staging(ctx, n_rows)
    # (one core copies ~20 GB/s: the pieces go to a few threads - numpy's copy / take loops run without the interpreter lock)
    jobs = []
    lo = 0
    for ti, t in live:
        d = per_type[t]
        hi = lo + len(d["a"])
        ch = np.asarray(d["chrom"])
        if len(ch) and (int(ch.min()) < 0 or int(ch.max()) >= len(chroms)):
            raise ValueError("%s rows: chromosome index outside [0, %d)" % (t, len(chroms)))
        seg_of_chrom = (ti * len(chroms) + crank).astype(np.int32)
        src = {"a": np.asarray(d["a"]), "b": np.asarray(d["b"]), "rid": np.asarray(d["read_id"]), "aux": np.asarray(d["aux"])}
        for c0 in range(0, hi - lo, _FILL_CHUNK):
            c1 = min(hi - lo, c0 + _FILL_CHUNK)
            jobs.append((np.take, (seg_of_chrom, ch[c0:c1]), dict(out=cat["seg"][lo + c0:lo + c1], mode="clip")))
            for k, v in src.items():
                jobs.append((np.copyto, (cat[k][lo + c0:lo + c1], v[c0:c1]), dict(casting="unsafe")))
        lo = hi
    if len(jobs) > 8:
        for f in [_fill_pool().submit(fn, *args, **kw) for fn, args, kw in jobs]:
            f.result()
    else:
        for fn, args, kw in jobs:
            fn(*args, **kw)
    n_seg = len(TYPES) * len(chroms)
    major = np.zeros(n_seg, np.uint8)
    nodedup = np.zeros(n_seg, np.uint8)
    for ti, t in enumerate(TYPES):
        if t in ("INV", "TRA"):
            major[ti * len(chroms):(ti + 1) * len(chroms)] = 1
    ti_ins = TYPES.index("INS")
    cb = None
    ins = per_type.get("INS")
    if ins is not None and ins.get("seq") is not None and len(ins["a"]):
        nodedup[ti_ins * len(chroms):(ti_ins + 1) * len(chroms)] = 1
        ins_base = sum(len(per_type[t]["a"]) for t in TYPES[:ti_ins] if t in per_type)
        seqs = ins["seq"]
        half = ins.get("half")
        cb = tie_callback(lambda s_: seqs[s_ - ins_base], (lambda s_: half[s_ - ins_base]) if half is not None else (lambda s_: 0))
    r = rebuild_columns(ctx, cat["seg"], cat["a"], cat["b"], cat["rid"], cat["aux"], major, nodedup, keep_on_device=True, tie_order=cb,
                        src_row_out=cat["src_row"])
    if r["n_ins_ties"]:                                     # (with a tie callback the device order is final; never with `assert`: -O strips it)
        raise ValueError("%d INS rows tie on (position, length, read) and were not settled on the device: the batch is not in the "
                         "reference's order" % r["n_ins_ties"])
    off = np.r_[0, np.cumsum(r["seg_count"])]
    segs, tasks = [], []
    for s in range(n_seg):
        if r["seg_count"][s] == 0:
            continue
        t, ci = TYPES[s // len(chroms)], order[s % len(chroms)]
        segs.append(params_segment(t, ci, int(off[s]), int(off[s + 1])))
        tasks.append((t, chroms[ci]))
    kw = {}
    if reads is not None:
        rc = np.asarray(reads["chrom"], np.int64)
        o = np.argsort(rc, kind="stable")
        kw = dict(reads_off=np.searchsorted(rc[o], np.arange(len(chroms) + 1)).astype(np.int64), r_start=np.asarray(reads["start"], np.int64)[o],
                  r_end=np.asarray(reads["end"], np.int64)[o], r_primary=np.asarray(reads["primary"], np.uint8)[o], r_id=np.asarray(reads["read_id"], np.int32)[o])
    batch = _abi.HostBatch.on_device(np.array(segs, dtype=_abi.SEGMENT_DTYPE), r["dev"], r["n_out"], n_chrom=len(chroms), keep=ctx, **kw)
    return batch, tasks, r["src_row"]


# ------------------------------------------------------------------------------------ the device-resident signature pool
def pool_reset(ctx):
    ctx._check(lib().csv_pool_reset(ctx._h))


def pool_rows(ctx):
    n = C.c_int64(0)
    ctx._check(lib().csv_pool_rows(ctx._h, C.byref(n)))
    return int(n.value)


def pool_append(ctx, seg_id, a, b, read, aux):
    """rows made on the host (the split-read candidates: they are built from text) -> the context's pool"""
    seg_id = np.ascontiguousarray(seg_id, np.int32); a = np.ascontiguousarray(a, np.int64); b = np.ascontiguousarray(b, np.int64)
    read = np.ascontiguousarray(read, np.int32); aux = np.ascontiguousarray(aux, np.int32)
    ctx._check(lib().csv_pool_append(ctx._h, len(a), seg_id.ctypes.data, a.ctypes.data, b.ctypes.data, read.ctypes.data, aux.ctypes.data))


def rebuild_pool(ctx, read_rank, seg_aux_major, seg_nodedup=None, keep_on_device=True, tie_order=None):
    """csv_rebuild_signatures over the context's pool (CSV_RB_FROM_POOL): the rows the extraction kernels left on the device
    (extract.cigar_signatures(pool=...)) and those appended with pool_append, sorted and de-duplicated; a row's read index is
    replaced by read_rank[index] (rank of the read's name in Python string order).  Same result dict as rebuild_columns;
    src_row numbers the pool's rows (extraction order)."""
    L = lib()
    L.csv_rebuild_signatures.restype = C.c_int
    L.csv_rebuild_signatures.argtypes = [C.c_void_p, C.POINTER(RebuildIn), C.POINTER(RebuildOut)]
    rank = np.ascontiguousarray(read_rank, np.int32)
    major = np.ascontiguousarray(seg_aux_major, np.uint8)
    nodedup = None if seg_nodedup is None else np.ascontiguousarray(seg_nodedup, np.uint8)
    n = pool_rows(ctx)
    o = dict(src_row=np.empty(n, np.int32))
    if not keep_on_device:
        o.update(seg_id=np.empty(n, np.int32), a=np.empty(n, np.int64), b=np.empty(n, np.int64), read_id=np.empty(n, np.int32), aux=np.empty(n, np.int32))
    seg_count = np.zeros(len(major), np.int64)
    rin = RebuildIn(n=0, n_seg=len(major), flags=_abi.RB_FROM_POOL | (_abi.RB_KEEP_ON_DEVICE if keep_on_device else 0), seg_aux_major=major.ctypes.data,
                    seg_nodedup=None if nodedup is None else nodedup.ctypes.data, read_rank=rank.ctypes.data, n_rank=len(rank),
                    tie_order=None if tie_order is None else C.cast(tie_order, C.c_void_p))
    rout = RebuildOut(seg_count=seg_count.ctypes.data, **{k: v.ctypes.data for k, v in o.items()})
    ctx._check(L.csv_rebuild_signatures(ctx._h, C.byref(rin), C.byref(rout)))
    k = int(rout.n_out)
    r = {name: v[:k] for name, v in o.items()}
    r.update(ms_device=float(rout.ms_device), n_passes=int(rout.n_passes), seg_count=seg_count, n_ins_ties=int(rout.n_ins_ties), n_out=k,
             n_tie_rows=int(rout.n_tie_rows), n_tie_dropped=int(rout.n_tie_dropped))
    if keep_on_device:
        r["dev"] = dict(a=rout.dev_a, b=rout.dev_b, read_id=rout.dev_read_id, aux=rout.dev_aux, seg_id=rout.dev_seg_id, src_row=rout.dev_src_row)
    return r


def finish_ins_ties(r, ins_segs, seq_of_src, half_of_src):
    """The INS tie groups of a sorted (not de-duplicated) row set `r` (dict of arrays from rebuild_columns): rows that agree
    in (segment, a, b, read_id).  Each group is ordered by sequence (stable: equal sequences keep the concatenation order
    of the extraction files, as the reference's stable sort does) and adjacent rows whose sequence and half-position are
    equal too are dropped (main script :774-775, :958-969).  Returns the index array to apply to r's arrays."""
    n = len(r["a"])
    idx = np.arange(n)
    if n < 2:
        return idx
    is_ins = np.isin(r["seg_id"], ins_segs)
    same = np.zeros(n, bool)
    same[1:] = (is_ins[1:] & (r["seg_id"][1:] == r["seg_id"][:-1]) & (r["a"][1:] == r["a"][:-1]) &
                (r["b"][1:] == r["b"][:-1]) & (r["read_id"][1:] == r["read_id"][:-1]))
    if not same.any():
        return idx
    starts = np.flatnonzero(same & ~np.r_[False, same[:-1]]) - 1          # first row of every tie group
    keep = np.ones(n, bool)
    order = idx.copy()
    for g0 in starts.tolist():
        g1 = g0 + 1
        while g1 < n and same[g1]:
            g1 += 1
        rows = list(range(g0, g1))
        src = r["src_row"][g0:g1].tolist()
        rows.sort(key=lambda i: seq_of_src(src[i - g0]))                    # Python's sort is stable
        order[g0:g1] = rows
        prev = None
        for pos, i in enumerate(rows):
            cur = (seq_of_src(src[i - g0]), int(half_of_src(src[i - g0])))
            if prev is not None and cur == prev:
                keep[g0 + pos] = False
            prev = cur
    return order[keep]


def store_from_unsorted(ctx, chroms, per_type, names=None, strands=("++", "--"), reads=None):
    """per_type: {"DEL": dict(chrom=int[], a=, b=, read_id=, aux=), ...} unsorted rows (chrom = index into `chroms`).
    An INS entry may carry `seq` (list of str, one per row) and `half` (0/1 per row: the position is x.5): the rows are
    then ordered and de-duplicated exactly as the reference does and the store holds the sequences; without them INS rows
    are de-duplicated on their integer columns only.
    Segments come out in the reference's order: types as main_ctrl submits them, chromosomes by name.
    `reads`: optional dict(chrom, start, end, primary, read_id): blocks keep their input order (csv_cluster_batch orders
    every block by start on the device)."""
    order = sorted(range(len(chroms)), key=lambda i: chroms[i])
    crank = np.zeros(len(chroms), np.int64)
    crank[order] = np.arange(len(chroms))
    cols = {k: [] for k in ("seg", "a", "b", "rid", "aux")}
    ins_base, ins_n = 0, 0
    n_rows = 0
    for ti, t in enumerate(TYPES):
        if t not in per_type or len(per_type[t]["a"]) == 0:
            continue
        d = per_type[t]
        ch = np.asarray(d["chrom"], np.int64)
        if t == "INS":
            ins_base, ins_n = n_rows, len(ch)
        cols["seg"].append((ti * len(chroms) + crank[ch]).astype(np.int32))
        cols["a"].append(np.asarray(d["a"], np.int64)); cols["b"].append(np.asarray(d["b"], np.int64))
        cols["rid"].append(np.asarray(d["read_id"], np.int32)); cols["aux"].append(np.asarray(d["aux"], np.int32))       # (the ABI's widths: no conversion pass later)
        n_rows += len(ch)
    cat = {k: np.concatenate(v) if v else np.zeros(0, np.int64) for k, v in cols.items()}
    n_seg = len(TYPES) * len(chroms)
    major = np.zeros(n_seg, np.uint8)
    nodedup = np.zeros(n_seg, np.uint8)
    ins_seq_in = per_type.get("INS", {}).get("seq") if "INS" in per_type else None
    ti_ins = TYPES.index("INS")
    for ti, t in enumerate(TYPES):
        if t in ("INV", "TRA"):
            major[ti * len(chroms):(ti + 1) * len(chroms)] = 1
    if ins_seq_in is not None:
        nodedup[ti_ins * len(chroms):(ti_ins + 1) * len(chroms)] = 1
    r = rebuild_columns(ctx, cat["seg"], cat["a"], cat["b"], cat["rid"], cat["aux"], major, nodedup)
    ins_seq = None
    if ins_seq_in is not None:
        half = per_type["INS"].get("half")
        half = np.zeros(ins_n, np.uint8) if half is None else np.asarray(half, np.uint8)
        sel = finish_ins_ties(r, np.arange(ti_ins * len(chroms), (ti_ins + 1) * len(chroms)),
                              lambda s: ins_seq_in[s - ins_base], lambda s: half[s - ins_base])
        for k in ("seg_id", "a", "b", "read_id", "aux", "src_row"):
            r[k] = r[k][sel]
        is_ins = (r["seg_id"] // len(chroms)) == ti_ins
        ins_seq = {int(i): ins_seq_in[int(r["src_row"][i]) - ins_base] for i in np.flatnonzero(is_ins).tolist()}
    seg_sorted = r["seg_id"]
    bounds = np.flatnonzero(np.r_[True, seg_sorted[1:] != seg_sorted[:-1], True]) if len(seg_sorted) else np.zeros(1, np.int64)
    seg_index = {}
    for i in range(len(bounds) - 1):
        s = int(seg_sorted[bounds[i]])
        seg_index[(TYPES[s // len(chroms)], chroms[order[s % len(chroms)]])] = (int(bounds[i]), int(bounds[i + 1]))
    kw = {}
    if reads is not None:
        rc = np.asarray(reads["chrom"], np.int64)
        o = np.argsort(rc, kind="stable")                       # by chromosome only, as main script :810 leaves the block
        off = np.searchsorted(rc[o], np.arange(len(chroms) + 1)).astype(np.int64)
        kw = dict(reads_off=off, r_start=np.asarray(reads["start"], np.int64)[o], r_end=np.asarray(reads["end"], np.int64)[o],
                  r_primary=np.asarray(reads["primary"], np.uint8)[o], r_id=np.asarray(reads["read_id"], np.int32)[o])
    st = SigStore(chroms=list(chroms), a=r["a"], b=r["b"], read_id=r["read_id"], aux=r["aux"], seg_index=seg_index,
                  names=names or NameTable(), strands=tuple(strands), ins_seq=ins_seq, **kw)
    return st, r
