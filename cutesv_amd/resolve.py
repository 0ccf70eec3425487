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

Who owns the GPU.  Called in the process that imported this module (a script, a test, one process per GPU running
`cluster_stage`), a task runs on that process's own `engine.Context` (one HIP device, created lazily).  Called in a worker
of the reference's forked `Pool` (main script :1113) the tasks of ALL workers run on ONE context per GPU, owned by a broker
process (`broker.py`): a worker walks its pickle, hands the columns over in shared memory and builds its rows from the
structure of arrays it gets back.  N contexts on a device would each pay the runtime's start (and their arenas, tables and
page-locked landing zones) inside the stage.  CUTESV_AMD_BROKER = auto (default) | 1 (always) | 0 (a context per worker,
created after fork).  `warm_up()` in the pool's parent starts the broker ahead of the stage.
"""
import logging
import os
import time

from . import _abi, engine, rows as rows_mod
from .columns import SigStore, Params, TYPES

_ctx = None
_ctx_pid = None
_stores = {}
_brokers = []                      # warm_up(): the broker processes this (parent) process started

# what the row builders read of a result (rows_layout.h; `search_pos`, `dv`, `call_cluster` stay on the device)
ROW_FIELDS = ("call_aux", "cipos", "cilen", "seq_pick", "dr", "gl_idx")


def device_list():
    """GPUs the stage may use: CUTESV_AMD_DEVICES = "0,2,3" or a count "8" (default: one device, see device_index)"""
    v = os.environ.get("CUTESV_AMD_DEVICES", "").strip()
    if not v:
        return None
    if "," in v:
        return [int(x) for x in v.split(",") if x.strip() != ""]
    return list(range(int(v)))


def _worker_number():
    """1-based index of this process among its pool's workers (multiprocessing numbers them); 0 outside a pool"""
    import multiprocessing as mp
    ident = getattr(mp.current_process(), "_identity", ())
    return int(ident[0]) if ident else 0


def device_index():
    """GPU of this process: CUTESV_AMD_DEVICE; else the pool worker's share of CUTESV_AMD_DEVICES (worker k of the pool ->
    device list[(k - 1) mod n]: the pool hands a task to whichever worker is free, so the devices fill evenly); else
    LOCAL_RANK; else 0."""
    if os.environ.get("CUTESV_AMD_DEVICE", "") != "":
        return int(os.environ["CUTESV_AMD_DEVICE"])
    devs = device_list()
    if devs:
        return devs[(max(_worker_number(), 1) - 1) % len(devs)]
    if os.environ.get("LOCAL_RANK", "") != "":
        return int(os.environ["LOCAL_RANK"])
    return 0


def in_pool_worker():
    import multiprocessing as mp
    return mp.parent_process() is not None


def use_broker():
    mode = os.environ.get("CUTESV_AMD_BROKER", "auto").lower()
    if mode in ("0", "off", "no"):
        return False
    if mode in ("1", "on", "yes"):
        return True
    return in_pool_worker()


def context():
    """this process's engine: an `engine.Context`, or - in a pool worker - a `broker.Client` of the GPU's broker (same
    `cluster_batch`).  Never inherited across a fork: a context created by the parent is not used by its children."""
    global _ctx, _ctx_pid
    if _ctx is not None and _ctx_pid not in (None, os.getpid()):
        _ctx = None                                   # (forked with a parent's engine: HIP state does not survive a fork)
    if _ctx is None:
        if use_broker():
            from . import broker
            _ctx = broker.Client.connect(device_index())
        else:
            _ctx = engine.Context(device_index())
        _ctx_pid = os.getpid()
    return _ctx


def warm_up(devices=None, linger=None):
    """Start the GPU broker(s) of the pool THIS process is about to fork (main script :1113), so that the HIP runtime's start
    and the context's first allocations overlap whatever the caller does until the first task arrives (in cuteSV: the whole
    extraction phase).  Optional: without it the first worker to need a broker starts it.  Returns the broker names."""
    from . import broker
    devs = devices if devices is not None else (device_list() or [device_index()])
    # what every forked worker would otherwise do in its first task (~10 ms each, on the stage's critical path when the pool
    # has as many workers as tasks): the CPU-side extension modules and the whole cal_GL table - inherited through fork.
    # (Never the HIP library: a parent that has touched the runtime cannot fork workers that use it.)
    from . import _rows_native, _cols_native, genotype          # noqa: F401
    genotype.fill_table()
    prefix = "cutesv_amd-%d-%d" % (os.getuid(), os.getpid())
    os.environ["CUTESV_AMD_BROKER_NAME"] = prefix    # (inherited by the workers: they look for exactly these sockets)
    names = []
    for d in devs:
        name = broker.socket_name(os.getpid(), d)
        if broker._try_connect(name) is None:
            # (started ahead of the stage, the broker also page-locks staging for batches of this size - signatures, reads - now)
            _brokers.append(broker.spawn(name, d, os.getpid(), linger=linger, prealloc=os.environ.get("CUTESV_AMD_BROKER_PREALLOC", "4000000,8000000")))
        names.append(name)
    return names


def shut_down(devices=None):
    """stop the brokers warm_up() started (they also leave by themselves when this process ends)"""
    from . import broker
    devs = devices if devices is not None else (device_list() or [device_index()])
    for d in devs:
        s = broker._try_connect(broker.socket_name(os.getpid(), d))
        if s is not None:
            try:
                broker.Client(s, d, "").shutdown()
            except broker.BrokerError:
                pass
            s.close()
    for p in _brokers:
        try:
            p.wait(timeout=10)
        except Exception:          # noqa: BLE001
            pass
    del _brokers[:]
    os.environ.pop("CUTESV_AMD_BROKER_NAME", None)


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


def _block_end(offsets, chrom, file_len):
    """where the pickled block of `chrom` ends in its file: the blocks lie back to back (main script :817-857 dumps them one after
    another and records where each begins), so the next larger offset of the index, or the end of the file.  A hint for the
    walker's buffers, nothing more."""
    try:
        off = int(offsets[chrom])
        later = [int(o) for o in offsets.values() if int(o) > off]
        return min(later) if later else int(file_len)
    except (TypeError, ValueError, KeyError):
        return None


def _store_for(work_dir, sigs_index, svtype, chrom, need_reads, gt_margin=None):
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
            cache = key = None
            if want_reads and not os.environ.get("CUTESV_AMD_NO_READS_CACHE"):
                eng = _ctx if _ctx is not None else (context() if use_broker() else None)
                if eng is not None and hasattr(eng, "reads_get"):          # (a pool worker: the GPU's broker hands walked blocks round)
                    k_ = _maps["%sreads.pickle" % work_dir][0]
                    cache, key = eng, (k_[0], k_[1], k_[2], int(sigs_index["reads"][chrom]))
            st = SigStore.from_task_pickles(svtype, chrom, sig_map, sigs_index[svtype][chrom],
                                            reads_map, sigs_index["reads"][chrom] if want_reads else None,
                                            gt_margin=None if os.environ.get("CUTESV_AMD_ALL_READS") else gt_margin,
                                            reads_cache=cache, reads_key=key,
                                            sig_end=_block_end(sigs_index[svtype], chrom, len(sig_map)),
                                            reads_end=_block_end(sigs_index["reads"], chrom, len(reads_map)) if want_reads else None)
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


_last_marks = [0.0, 0.0]


def run_batch(store, segments, tasks, ctx=None, timeline=None):
    """segments: csv_segment records for `tasks` [(type, chr)] -> {(type, chr): rows}"""
    global _ctx
    own = ctx is None
    ctx = ctx or context()
    hb = _batch_of(store, segments)
    if timeline:
        _last_marks[0] = time.time()
    try:
        res = ctx.cluster_batch(hb, reuse=True, fields=ROW_FIELDS)     # (consumed right here: the arrays may be recycled by the next call)
    except Exception as e:                              # noqa: BLE001
        from . import broker
        if not (own and isinstance(e, broker.BrokerError)):
            raise
        # the GPU's broker went away under this worker (killed, crashed): the task is tried once more on a fresh connection -
        # which starts a new broker if nobody listens - instead of failing every later task of the worker on a dead socket
        logging.warning("cutesv_amd: %s - reconnecting" % (e,))
        try:
            _ctx.close()
        except Exception:                               # noqa: BLE001
            pass
        _ctx = None
        ctx = context()
        res = ctx.cluster_batch(hb, reuse=True, fields=ROW_FIELDS)
    if timeline:
        _last_marks[1] = time.time()
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
    nw = store.narrow or {}                           # (int32 twins of the positions / lengths: SigStore.pinned(), a mapped .cols directory)
    kw = {}
    slim16 = os.environ.get("CUTESV_AMD_NO_DELTA16") is None      # (the 16-bit / interleaved / packed forms SigStore.pinned() keeps: ABI v8)
    if len(segs) and segs["genotype"].any() and store.reads_off is not None:
        kw = dict(reads_off=store.reads_off, r_start=nw.get("r_start", store.r_start), r_end=nw.get("r_end", store.r_end),
                  r_primary=store.r_primary, r_id=store.r_id)
        if slim16:
            kw.update(r_delta=nw.get("r_delta"), r_len16=nw.get("r_len16"), r_idp=nw.get("r_idp"))
        if ((segs["svtype"] == _abi.TRA) & (segs["genotype"] != 0)).any():
            kw["contig_len"] = store.contig_len
    return _abi.HostBatch(segs, nw.get("a", store.a), nw.get("b", store.b), store.read_id, store.aux, n_chrom=len(store.chroms),
                          a_delta=nw.get("a_delta") if slim16 else None,
                          rows8=nw.get("rows8") if os.environ.get("CUTESV_AMD_NO_ROWS8") is None else None, **kw)


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


def _one(work_dir, chrom, svtype, sigs_index, seg_of_store, need_reads=False, gt_margin=None):
    if chrom not in sigs_index.get(svtype, {}):       # INDEL:44-45, DUP:19-20, INV:33-34, TRA:31-32
        return (chrom, [])
    if not work_dir.endswith("/"):
        work_dir += "/"                               # (main_ctrl normalises it, main script :993-996)
    tl = os.environ.get("CUTESV_AMD_TIMELINE")        # (a directory: one line per task with its phases' clock readings)
    t0 = time.time() if tl else 0.0
    store = _store_for(work_dir, sigs_index, svtype, chrom, need_reads, gt_margin)
    if (svtype, chrom) not in store.seg_index:
        return (chrom, [])
    seg = seg_of_store(store)
    t1 = time.time() if tl else 0.0
    rows = run_batch(store, [seg], [(svtype, chrom)], timeline=tl)[(svtype, chrom)]
    logging.info("Finished %s:%s." % (chrom, svtype))
    if tl:
        with open(os.path.join(tl, "w%d.tl" % os.getpid()), "a") as f:
            f.write("%s %s %d %d %.6f %.6f %.6f %.6f %.6f\n" % (svtype, chrom, store.n_sig, store.n_reads, t0, t1, _last_marks[0], _last_marks[1], time.time()))
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
    return _one(path, chrom, svtype, sigs_index, seg, need_reads=bool(action), gt_margin=max_cluster_bias if svtype == "DEL" else 1000)


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
    return _one(path, chrom, "INV", sigs_index, seg, need_reads=bool(action), gt_margin=max_cluster_bias)


def run_dup(args):
    path, chrom, read_count, max_cluster_bias, sv_size, _bam, action, MaxSize, _gt_round, sigs_index = args

    def seg(store):
        beg, end = store.seg_index[("DUP", chrom)]
        return _abi.make_segment("DUP", store.chroms.index(chrom), beg, end, max_cluster_bias, read_count,
                                 sv_size=sv_size, max_size=MaxSize, gt_bias=max_cluster_bias, genotype=bool(action))
    return _one(path, chrom, "DUP", sigs_index, seg, need_reads=bool(action), gt_margin=max_cluster_bias)


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


# ------------------------------------------------------------------------------------------------ the caller, restated
def main_ctrl_phase3(work_dir, sigs_index, params, threads, fns=None, bam="bam", start_method="fork", on_error=None):
    """The reference's phase 3 as `main_ctrl` runs it (main script :1113-1199), for tests and bench.py: a
    `Pool(processes=threads)`, one `map_async(run_X, [tuple])` per (chromosome, type) in DEL, INS, INV, DUP, TRA order with the
    reference's argument tuples, `res.get()[0]`, `results[chr].extend(rows)`; an exception of a task is swallowed after
    logging, as the reference does (:1198-1199).  fns: the five callables (default: this module's - the drop-in);
    params: columns.Params (the argparse values).  Returns {chr: rows}."""
    import multiprocessing as mp
    p = params
    fns = fns or dict(DEL=run_del, INS=run_ins, INV=run_inv, DUP=run_dup, TRA=run_tra)
    result = []
    with mp.get_context(start_method).Pool(processes=int(threads)) as analysis_pools:
        for chrom in sigs_index["DEL"]:
            para = [(work_dir, chrom, "DEL", p.min_support, p.diff_ratio_merging_DEL, p.max_cluster_bias_DEL, min(p.min_support, 5),
                     bam, p.genotype, p.gt_round, p.remain_reads_ratio, sigs_index)]
            result.append(analysis_pools.map_async(fns["DEL"], para))
        for chrom in sigs_index["INS"]:
            para = [(work_dir, chrom, "INS", p.min_support, p.diff_ratio_merging_INS, p.max_cluster_bias_INS, min(p.min_support, 5),
                     bam, p.genotype, p.gt_round, p.remain_reads_ratio, sigs_index)]
            result.append(analysis_pools.map_async(fns["INS"], para))
        for chrom in sigs_index["INV"]:
            para = [(work_dir, chrom, "INV", p.min_support, p.max_cluster_bias_INV, p.min_size, bam, p.genotype, p.max_size, p.gt_round, sigs_index)]
            result.append(analysis_pools.map_async(fns["INV"], para))
        for chrom in sigs_index["DUP"]:
            para = [(work_dir, chrom, p.min_support, p.max_cluster_bias_DUP, p.min_size, bam, p.genotype, p.max_size, p.gt_round, sigs_index)]
            result.append(analysis_pools.map_async(fns["DUP"], para))
        for chrom in sigs_index["TRA"]:
            para = [(work_dir, chrom, p.min_support, p.diff_ratio_filtering_TRA, p.max_cluster_bias_TRA, bam, p.genotype, p.gt_round, sigs_index)]
            result.append(analysis_pools.map_async(fns["TRA"], para))
        results = {}
        for res in result:
            try:
                chrom, svs = res.get()[0]
                if chrom not in results:
                    results[chrom] = []
                results[chrom].extend(svs)
            except Exception as e:          # noqa: BLE001  (main script :1198-1199 logs and carries on)
                logging.info("LocalError: %r" % (e,))
                if on_error is not None:
                    on_error(e)
    return results
