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


def _store_for(work_dir, sigs_index):
    """Signatures of a work dir as flat columns: `<work_dir>cutesv_amd.cols/` if our rebuild step wrote
    it, otherwise converted once per process from the reference's pickles (main script :817-857)."""
    st = _stores.get(work_dir)
    if st is None:
        cols = os.path.join(work_dir, "cutesv_amd.cols")
        st = SigStore.load(cols) if os.path.isdir(cols) else SigStore.from_reference_workdir(work_dir, sigs_index)
        _stores[work_dir] = st
    return st


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


def cluster_stage(store, params, tasks=None, ctx=None):
    """The whole phase 3 for the tasks of `store` this process owns -> {chr: rows}."""
    tasks = tasks or store.tasks()
    segs = [store.segment(t, ch, params) for t, ch in tasks]
    by_task = run_batch(store, segs, tasks, ctx)
    results = {}
    for t in TYPES:                                   # main script :1191-1197 extends in submission order
        for (tt, ch) in tasks:
            if tt == t:
                results.setdefault(ch, []).extend(by_task[(tt, ch)])
    return results


def _one(work_dir, chrom, svtype, sigs_index, seg_of_store):
    if chrom not in sigs_index.get(svtype, {}):       # INDEL:44-45, DUP:19-20, INV:33-34, TRA:31-32
        return (chrom, [])
    store = _store_for(work_dir, sigs_index)
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
    return _one(path, chrom, svtype, sigs_index, seg)


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
    return _one(path, chrom, "INV", sigs_index, seg)


def run_dup(args):
    path, chrom, read_count, max_cluster_bias, sv_size, _bam, action, MaxSize, _gt_round, sigs_index = args

    def seg(store):
        beg, end = store.seg_index[("DUP", chrom)]
        return _abi.make_segment("DUP", store.chroms.index(chrom), beg, end, max_cluster_bias, read_count,
                                 sv_size=sv_size, max_size=MaxSize, gt_bias=max_cluster_bias, genotype=bool(action))
    return _one(path, chrom, "DUP", sigs_index, seg)


def run_tra(args):
    path, chrom, read_count, overlap_size, max_cluster_bias, bam, action, gt_round, sigs_index = args

    def seg(store):
        beg, end = store.seg_index[("TRA", chrom)]
        if action and store.contig_len is None:
            # call_gt only takes the reference lengths from the BAM (cuteSV_resolveTRA.py:264, 291); the
            # alignments come from the reads table (include/cutesv_hip.h)
            from .bam_header import reference_lengths
            import numpy as np
            lens = reference_lengths(bam)
            store.contig_len = np.array([lens[c] for c in store.chroms], np.int64)
        return _abi.make_segment("TRA", store.chroms.index(chrom), beg, end, max_cluster_bias, read_count,
                                 diff_ratio=overlap_size, gt_bias=max_cluster_bias, genotype=bool(action),
                                 gt_round=gt_round)
    return _one(path, chrom, "TRA", sigs_index, seg)
