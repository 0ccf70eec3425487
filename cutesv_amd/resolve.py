"""Drop-in face of the clustering stage: the reference's five task callables and a batched form.

The reference's phase 3 (cuteSV main script :1113-1199) submits one `run_*` call per (chromosome, type)
to a multiprocessing pool; each call returns `(chr, rows)`.  This module offers

  * `run_del / run_ins / run_inv / run_dup / run_tra(args)` with the reference's exact argument tuples
    (cuteSV_resolveINDEL.py:17-18, 222-223, 435-439; cuteSV_resolveINV.py:6-7; cuteSV_resolveDUP.py:17-18;
    cuteSV_resolveTRA.py:30) and return value, so `main_ctrl` can dispatch to them unchanged, and
  * `cluster_stage(store, params)` — every task of a genome in ONE batch per GPU (one H2D, one launch
    sequence, one D2H), returning `{chr: rows}` in the order `main_ctrl` concatenates them
    (:1191-1197: DEL, INS, INV, DUP, TRA per chromosome).

There is no CPU path: the work is done by libcutesv_hip.so; a missing library raises on first use.
Each process owns one `engine.Context` (one HIP device, created lazily AFTER fork).
"""
import logging
import os

from . import _abi, engine, rows as rows_mod
from .columns import SigStore, Params, TYPES

_ctx = None
_stores = {}


def device_index():
    """GPU of this worker: CUTESV_AMD_DEVICE, else LOCAL_RANK, else 0."""
    for k in ("CUTESV_AMD_DEVICE", "LOCAL_RANK"):
        if os.environ.get(k, "") != "":
            return int(os.environ[k])
    return 0


def context():
    global _ctx
    if _ctx is None:
        _ctx = engine.Context(device_index())
    return _ctx


_maps = {}


def _mapped(path):
    """the file, memory-mapped read-only (kept for the life of the process: a pool worker runs many tasks on the same files);
    None for a file that cannot be mapped (empty, missing)"""
    import mmap
    try:
        stt = os.stat(path)
        key = (path, stt.st_size, stt.st_mtime_ns)
        m = _maps.get(path)
        if m is not None and m[0] == key:
            return m[1]
        if stt.st_size == 0:
            return None
        with open(path, "rb") as f:
            mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        _maps[path] = (key, mm)
        return mm
    except (OSError, ValueError):
        return None


def _store_for(work_dir, sigs_index, svtype, chrom, need_reads):
    """Signatures of ONE task as flat columns: the mmap'ed `<work_dir>cutesv_amd.cols/` if our rebuild step wrote it
    (shared by every task of the process), otherwise just this task's pickled list - and, when it genotypes, its
    chromosome's reads block - converted from the reference's files (main script :817-857).  A pool worker therefore
    pays for the tasks it runs, not for the genome (the reference's workers do the same, INDEL:52-58)."""
    cols = os.path.join(work_dir, "cutesv_amd.cols")
    if os.path.isdir(cols):
        st = _stores.get(work_dir)
        if st is None:
            st = _stores[work_dir] = SigStore.load(cols)
        return st
    import pickle
    want_reads = need_reads and chrom in sigs_index.get("reads", {})
    if not os.environ.get("CUTESV_AMD_UNPICKLE"):      # (set: always through pickle.load, the r03 path)
        # the pickles walked in C straight out of the mapped files (no Python object per tuple); None: something in the stream
        # that the walker does not know - then pickle itself reads it
        sig_map = _mapped("%s%s.pickle" % (work_dir, svtype))
        reads_map = _mapped("%sreads.pickle" % work_dir) if want_reads else None
        if sig_map is not None and (reads_map is not None or not want_reads):
            st = SigStore.from_task_pickles(svtype, chrom, sig_map, sigs_index[svtype][chrom],
                                            reads_map, sigs_index["reads"][chrom] if want_reads else None)
            if st is not None:
                return st
    with open("%s%s.pickle" % (work_dir, svtype), "rb") as f:
        f.seek(sigs_index[svtype][chrom])
        sigs = pickle.load(f)
    reads = []
    if want_reads:
        with open("%sreads.pickle" % work_dir, "rb") as f:
            f.seek(sigs_index["reads"][chrom])
            reads = pickle.load(f)
    chroms = None
    if svtype == "TRA":                               # chr2 ranks must cover every mate chromosome of the list
        chroms = sorted({chrom} | {x[2] for x in sigs})
    return SigStore.from_task_lists(svtype, chrom, sigs, reads, chroms=chroms)


def run_batch(store, segments, tasks, ctx=None):
    """segments: csv_segment records for `tasks` [(type, chr)] -> {(type, chr): rows}"""
    import numpy as np
    ctx = ctx or context()
    segs = np.array(segments, dtype=_abi.SEGMENT_DTYPE)
    kw = {}
    if len(segs) and segs["genotype"].any() and store.reads_off is not None:
        kw = dict(reads_off=store.reads_off, r_start=store.r_start, r_end=store.r_end,
                  r_primary=store.r_primary, r_id=store.r_id)
        if ((segs["svtype"] == _abi.TRA) & (segs["genotype"] != 0)).any():
            kw["contig_len"] = store.contig_len
    hb = _abi.HostBatch(segs, store.a, store.b, store.read_id, store.aux, n_chrom=len(store.chroms), **kw)
    res = ctx.cluster_batch(hb, reuse=True)           # (consumed right here: the arrays may be recycled by the next call)
    per_seg = rows_mod.rows_by_segment(store, hb.segments, res)
    return {t: per_seg[k] for k, t in enumerate(tasks)}


def cluster_stage(store, params, tasks=None, ctx=None, lazy=False):
    """The whole phase 3 for the tasks of `store` this process owns -> {chr: rows}.
    lazy=True: the per-chromosome values are rows.LazyRows - list-like (extend / sort by int(row[2]) / iterate / index, what
    main_ctrl and generate_output do with them: main script :1191-1197, GT:242-252) but backed by the structure of arrays: a
    row's strings exist only once somebody reads the row, and a consumer that takes the arrays themselves
    (`vcf.emit_stage(results, ...)`) never creates one (a 30x genome: ~11 ms of CPython str creation for 25 k rows)."""
    tasks = tasks or store.tasks()
    segs = [store.segment(t, ch, params) for t, ch in tasks]
    if lazy:
        return _cluster_stage_lazy(store, segs, tasks, ctx)
    by_task = run_batch(store, segs, tasks, ctx)
    results = {}
    for t in TYPES:                                   # main script :1191-1197 extends in submission order
        for (tt, ch) in tasks:
            if tt == t:
                results.setdefault(ch, []).extend(by_task[(tt, ch)])
    return results


def _batch_of(store, segments):
    import numpy as np
    segs = np.array(segments, dtype=_abi.SEGMENT_DTYPE)
    kw = {}
    if len(segs) and segs["genotype"].any() and store.reads_off is not None:
        kw = dict(reads_off=store.reads_off, r_start=store.r_start, r_end=store.r_end, r_primary=store.r_primary, r_id=store.r_id)
        if ((segs["svtype"] == _abi.TRA) & (segs["genotype"] != 0)).any():
            kw["contig_len"] = store.contig_len
    return _abi.HostBatch(segs, store.a, store.b, store.read_id, store.aux, n_chrom=len(store.chroms), **kw)


def _cluster_stage_lazy(store, segs, tasks, ctx):
    import numpy as np
    ctx = ctx or context()
    hb = _batch_of(store, segs)
    res = ctx.cluster_batch(hb, reuse=True)
    backing, ranges = rows_mod.lazy_rows_by_segment(store, hb.segments, res, ctx)     # (`res` is lent to the rows, or copied)
    results = {}
    for t in TYPES:                                   # main script :1191-1197 extends in submission order
        for k, (tt, ch) in enumerate(tasks):
            if tt == t:
                lo, hi = ranges[k]
                results.setdefault(ch, rows_mod.LazyRows()).extend(rows_mod.LazyRows(backing, np.arange(lo, hi, dtype=np.int64)))
    return results


def _one(work_dir, chrom, svtype, sigs_index, seg_of_store, need_reads=False):
    if chrom not in sigs_index.get(svtype, {}):       # INDEL:44-45, DUP:19-20, INV:33-34, TRA:31-32
        return (chrom, [])
    if not work_dir.endswith("/"):
        work_dir += "/"                               # (main_ctrl normalises it, main script :993-996)
    store = _store_for(work_dir, sigs_index, svtype, chrom, need_reads)
    if (svtype, chrom) not in store.seg_index:
        return (chrom, [])
    seg = seg_of_store(store)
    rows = run_batch(store, [seg], [(svtype, chrom)])[(svtype, chrom)]
    logging.info("Finished %s:%s." % (chrom, svtype))
    return (chrom, rows)


def _indel(args, svtype):
    (path, chrom, _svtype, read_count, threshold_gloab, max_cluster_bias, minimum_support_reads,
     _bam, action, _gt_round, remain_reads_ratio, sigs_index) = args

    def seg(store):
        beg, end = store.seg_index[(svtype, chrom)]
        return _abi.make_segment(svtype, store.chroms.index(chrom), beg, end, max_cluster_bias, read_count,
                                 diff_ratio=threshold_gloab, remain_reads_ratio=remain_reads_ratio,
                                 gt_bias=max_cluster_bias if svtype == "DEL" else 1000,       # INDEL:103 / :312
                                 min_support_reads=minimum_support_reads, genotype=bool(action))
    return _one(path, chrom, svtype, sigs_index, seg, need_reads=bool(action))


def run_del(args):
    return _indel(args, "DEL")


def run_ins(args):
    return _indel(args, "INS")


def run_inv(args):
    path, chrom, _svtype, read_count, max_cluster_bias, sv_size, _bam, action, MaxSize, _gt_round, sigs_index = args

    def seg(store):
        beg, end = store.seg_index[("INV", chrom)]
        return _abi.make_segment("INV", store.chroms.index(chrom), beg, end, max_cluster_bias, read_count,
                                 sv_size=sv_size, max_size=MaxSize, gt_bias=max_cluster_bias, genotype=bool(action))
    return _one(path, chrom, "INV", sigs_index, seg, need_reads=bool(action))


def run_dup(args):
    path, chrom, read_count, max_cluster_bias, sv_size, _bam, action, MaxSize, _gt_round, sigs_index = args

    def seg(store):
        beg, end = store.seg_index[("DUP", chrom)]
        return _abi.make_segment("DUP", store.chroms.index(chrom), beg, end, max_cluster_bias, read_count,
                                 sv_size=sv_size, max_size=MaxSize, gt_bias=max_cluster_bias, genotype=bool(action))
    return _one(path, chrom, "DUP", sigs_index, seg, need_reads=bool(action))


_warned_tra = False


def run_tra(args):
    """TRA task.  With --genotype the reference re-opens the BAM per call (call_gt, cuteSV_resolveTRA.py:258-309).
    CUTESV_AMD_TRA_GT selects how that is reproduced:
      bam          (default) cluster on the GPU, then the reference's own loop over pysam on the host (tra_bam.py):
                   identical to the reference on any BAM;
      reads_table  genotype on the GPU from the reads table (k_genotype_tra): identical only when the windows hold no
                   secondary / low-mapq alignments (main script :711-733) - a warning is logged once;
      off          leave the '.' fields."""
    global _warned_tra
    path, chrom, read_count, overlap_size, max_cluster_bias, bam, action, gt_round, sigs_index = args
    mode = os.environ.get("CUTESV_AMD_TRA_GT", "bam") if action else "off"
    if mode not in ("bam", "reads_table", "off"):
        raise ValueError("CUTESV_AMD_TRA_GT must be bam, reads_table or off")
    on_gpu = mode == "reads_table"
    if on_gpu and not _warned_tra:
        logging.warning("TRA calls are genotyped from the reads table (CUTESV_AMD_TRA_GT=reads_table): DR / GT / PL can differ "
                        "from cuteSV's BAM-based call_gt where secondary or low-mapq alignments overlap a breakpoint window")
        _warned_tra = True

    def seg(store):
        beg, end = store.seg_index[("TRA", chrom)]
        if on_gpu and store.contig_len is None:
            # call_gt only takes the reference lengths from the BAM (cuteSV_resolveTRA.py:264, 291); the
            # alignments come from the reads table (include/cutesv_hip.h)
            from .bam_header import reference_lengths
            import numpy as np
            lens = reference_lengths(bam)
            store.contig_len = np.array([lens[c] for c in store.chroms], np.int64)
        return _abi.make_segment("TRA", store.chroms.index(chrom), beg, end, max_cluster_bias, read_count,
                                 diff_ratio=overlap_size, gt_bias=max_cluster_bias, genotype=on_gpu,
                                 gt_round=gt_round)
    if on_gpu and not os.path.isdir(os.path.join(path, "cutesv_amd.cols")):
        # the reads table must hold BOTH chromosomes of every pair: load every reads block for this task
        res = _one_tra_reads_table(path, chrom, sigs_index, seg)
    else:
        res = _one(path, chrom, "TRA", sigs_index, seg)
    if mode == "bam" and res[1]:
        from .tra_bam import genotype_rows
        res = (res[0], genotype_rows(res[1], bam, max_cluster_bias, gt_round))
    return res


def _one_tra_reads_table(work_dir, chrom, sigs_index, seg_of_store):
    import pickle
    if chrom not in sigs_index.get("TRA", {}):
        return (chrom, [])
    if not work_dir.endswith("/"):
        work_dir += "/"
    with open("%sTRA.pickle" % work_dir, "rb") as f:
        f.seek(sigs_index["TRA"][chrom])
        sigs = pickle.load(f)
    chroms = sorted({chrom} | {x[2] for x in sigs})
    reads = []
    for c in chroms:
        if c in sigs_index.get("reads", {}):
            with open("%sreads.pickle" % work_dir, "rb") as f:
                f.seek(sigs_index["reads"][c])
                reads.extend(pickle.load(f))
    store = SigStore.from_tuple_lists({"TRA": sigs}, reads, chroms=chroms)
    rows = run_batch(store, [seg_of_store(store)], [("TRA", chrom)])[("TRA", chrom)]
    logging.info("Finished %s:TRA." % chrom)
    return (chrom, rows)
