"""Python restatement of the reference's phase-3 pool (TEST / BASELINE INFRASTRUCTURE ONLY).

Purpose: the reference is pure Python and cannot travel to the GPU box, so bench.py's
`cpu_baseline` leg times THIS file there as the stand-in for "cuteSV --threads <host cores>":
the same execution model (one task per (chromosome, type) in a multiprocessing.Pool, a Python
loop per signature, dict / sorted / numpy-scalar calls per cluster, the sweep-line genotype),
written from the specification in SURVEY.md §8 / Appendix A, not from the reference's text.
Its rows are pinned to the reference's by tests/test_oracle_golden.py (small golden cases).

Mapping to the reference (file:line in /root/reference/src/cuteSV/):
    chain()            cuteSV_resolveINDEL.py:55-100, cuteSV_resolveDUP.py:28-70, cuteSV_resolveINV.py:45-92,
                       cuteSV_resolveTRA.py:39-102
    indel_cluster()    cuteSV_resolveINDEL.py:110-219, 319-432
    dup_cluster()      cuteSV_resolveDUP.py:79-131
    inv_cluster()      cuteSV_resolveINV.py:101-203
    tra_cluster()      cuteSV_resolveTRA.py:106-254
    sweep_cover()      cuteSV_genotype.py:95-159        genotype_rows()  cuteSV_resolveINDEL.py:441-479 etc.
    likelihood()       cuteSV_genotype.py:14-56
"""
from math import log10
from multiprocessing import Pool

import numpy as np

_ERR, _PRIOR = 0.1, float(1 / 3)


# ---------------------------------------------------------------------------------------- genotype math
def likelihood(c0, c1):
    if (c0, c1) == (3, 1):
        return "0/1", "3,3,24", 3, 3.0
    if (c0, c1) == (6, 2):
        return "0/1", "3,3,45", 3, 3.0
    tot = c0 + c1
    if tot > 100:
        c0 = int(100 * float(c0 / tot))
        c1 = 100 - c0
    raw = [np.float64(pow(1 - _ERR, c0) * pow(_ERR, c1) * (1 - _PRIOR) / 2),
           np.float64(pow(0.5, c0 + c1) * _PRIOR),
           np.float64(pow(_ERR, c0) * pow(1 - _ERR, c1) * (1 - _PRIOR) / 2)]
    lg = np.array([log10(x) for x in raw])
    top = max(lg)
    norm = list(np.minimum(lg - (top + log10(sum(pow(10.0, x - top) for x in lg))), 0.0))
    p = [pow(10, x) for x in norm]
    pl = [int(np.around(-10 * log10(x))) for x in p]
    gq = max(int(-10 * log10(p[1] + p[2])), int(-10 * log10(p[0] + p[2])), int(-10 * log10(p[0] + p[1])))
    return ("0/0", "0/1", "1/1")[norm.index(max(norm))], "%d,%d,%d" % tuple(pl), gq, abs(np.around(-10 * log10(p[0]), 1))


def ci_text(sd, n):
    w = int(1.96 * sd / n ** 0.5)
    return "-%d,%d" % (w, w)


def sweep_cover(windows, reads):
    """Sweep line over window and read end points; returns per window the set of primary read names that
    span it.  Event order at equal coordinates: window right, read start, read end, window left."""
    ev = []
    for i, r in enumerate(reads):
        ev.append((r[0], 1, i))
        ev.append((r[1], 2, i))
    for j, w in enumerate(windows):
        ev.append((w[0], 3, j))
        ev.append((w[1], 0, j))
    ev.sort(key=lambda e: (e[0], e[1]))
    open_reads, at_left, cover = set(), {}, {}
    for coord, kind, idx in ev:
        if kind == 1:
            open_reads.add(idx)
        elif kind == 2:
            open_reads.discard(idx)
        elif kind == 3:
            at_left[idx] = set(open_reads)
        else:
            cover[idx] = at_left[idx] & open_reads
    return [set(reads[i][3] for i in cover[j] if reads[i][2] == 1) for j in range(len(windows))]


def _gt_fields(cover_names, support):
    dr = 0
    for nm in cover_names:
        if nm not in support:
            dr += 1
    gt, pl, gq, qual = likelihood(dr, len(support))
    return str(dr), str(gt), str(pl), str(gq), str(qual)


# ---------------------------------------------------------------------------------------- per-cluster refinement
def indel_cluster(members, chrom, svtype, read_count, ratio, min_reads, keep_ratio, genotype, out):
    per_read = {}
    for e in members:                      # first appearance keeps the slot; a strictly longer one replaces it
        cur = per_read.get(e[2])
        if cur is None or e[1] > cur[1]:
            per_read[e[2]] = e
    if len(per_read) < read_count:
        return
    by_len = sorted(per_read.values(), key=lambda e: e[1])
    gap = ratio * np.mean([e[1] for e in by_len])
    alleles = [[by_len[0]]]
    for prev, e in zip(by_len, by_len[1:]):
        if e[1] - prev[1] > gap:
            alleles.append([])
        alleles[-1].append(e)
    alleles.sort(key=len)
    for al in alleles:
        n = len(al)
        if n < min_reads:
            continue
        keep = max(int(keep_ratio * n), 1)
        pos = [e[0] for e in al]
        lens = [e[1] for e in al]
        pm = np.mean(pos)
        order = sorted(range(n), key=lambda i: abs(pos[i] - pm))
        kept = [pos[i] for i in order[:keep]]
        bp, search = np.mean(kept), kept[0]
        lm = np.mean(lens)
        order = sorted(range(n), key=lambda i: abs(lens[i] - lm))
        size = np.mean([lens[i] for i in order[:keep]])
        cipos, cilen = ci_text(np.std(pos), n), ci_text(np.std(lens), n)
        names = [e[2] for e in al]
        if svtype == "INS":
            seq = None
            for e in al:
                if len(e[3]) >= int(size):
                    bp, seq = e[0], e[3][:int(size)]
                    break
            if seq is None:
                continue
            if genotype:
                out.append([chrom, svtype, int(bp), int(size), n, cipos, cilen, int(bp), names, seq])
            else:
                out.append([chrom, svtype, str(int(bp)), str(int(size)), str(n), cipos, cilen, ".", "./.", ".,.,.", ".", ".",
                            ",".join(names), seq])
        else:
            if genotype:
                out.append([chrom, svtype, int(bp), int(-size), n, cipos, cilen, int(search), names])
            else:
                out.append([chrom, svtype, str(int(bp)), str(int(-size)), str(n), cipos, cilen, ".", "./.", ".,.,.", ".", ".",
                            ",".join(names)])


def dup_cluster(members, chrom, read_count, bias, min_size, max_size, genotype, out):
    if len(set(e[2] for e in members)) < read_count:
        return
    members = sorted(members, key=lambda e: e[1])
    groups = [[members[0]]]
    for prev, e in zip(members, members[1:]):
        if e[1] - prev[1] > bias:
            groups.append([])
        groups[-1].append(e)
    for g in groups:
        names = list(set(e[2] for e in g))
        if len(names) < read_count:
            continue
        lo, hi = int(len(g) * 0.4), int(len(g) * 0.6)
        if lo == hi:
            b1, b2 = g[lo][0], g[lo][1]
        else:
            mid = g[lo:hi]
            b1, b2 = int(sum(e[0] for e in mid) / len(mid)), int(sum(e[1] for e in mid) / len(mid))
        if b2 - b1 >= min_size and (b2 - b1 <= max_size or max_size == -1):
            if genotype:
                out.append([chrom, "DUP", b1, b2, names])
            else:
                out.append([chrom, "DUP", str(b1), str(b2 - b1), str(len(names)), ".", "./.", ".,.,.", ".", ".", ",".join(names)])


def inv_cluster(members, chrom, read_count, bias, min_size, max_size, genotype, out):
    strand = members[0][3]
    if len(set(e[2] for e in members)) < read_count:
        return
    members = sorted(members, key=lambda e: e[1])
    groups = [[members[0]]]
    for prev, e in zip(members, members[1:]):
        if e[1] - prev[1] > bias:
            groups.append([])
        groups[-1].append(e)
    for g in groups:
        if len(g) < read_count:
            continue
        names = list(dict.fromkeys(e[2] for e in g))
        b1, b2 = round(sum(e[0] for e in g) / len(g)), round(sum(e[1] for e in g) / len(g))
        size = b2 - b1
        if size >= min_size and len(names) >= read_count and (size <= max_size or max_size == -1):
            if genotype:
                out.append([chrom, "INV", b1, size, len(names), strand, names, b2])
            else:
                out.append([chrom, "INV", str(int(b1)), str(int(size)), str(len(names)), ".", "./.", strand, ".,.,.", ".", ".",
                            ",".join(names)])


_BND = {"A": "N[%s[", "B": "N]%s]", "C": "[%s[N", "D": "]%s]N"}


def tra_cluster(members, chr1, chr2, read_count, overlap, bias, out):
    kind = members[0][3]
    members = sorted(members, key=lambda e: e[1])
    groups = [[members[0][0], members[0][1], [members[0][2]]]]      # the first element is visited again below
    last = members[0][1]
    for e in members:
        if e[1] - last > bias:
            groups.append([e[0], e[1], [e[2]]])
        else:
            groups[-1][0] += e[0]; groups[-1][1] += e[1]; groups[-1][2].append(e[2])
        last = e[1]
    if len(set(e[2] for e in members)) < read_count:
        return
    groups.sort(key=lambda g: -len(set(g[2])))
    if kind not in _BND:
        return
    shift = 1 if kind in ("A", "C") else 0

    def emit(g):
        n = len(g[2])
        p1, p2 = int(g[0] / n), int(g[1] / n)
        names = set(g[2])
        out.append([chr1, _BND[kind] % ("%s:%s" % (chr2, p2 + shift)), str(p1), chr2, str(p2), str(len(names)),
                    ".", "./.", ".,.,.", ".", ".", ",".join(names)])

    if len(groups) > 1 and len(set(groups[1][2])) >= 0.5 * read_count:
        if len(set(groups[0][2])) + len(set(groups[1][2])) >= len(members) * overlap:
            emit(groups[0]); emit(groups[1])
    elif len(set(groups[0][2])) >= len(members) * overlap:
        emit(groups[0])


# ---------------------------------------------------------------------------------------- tasks
def _chain(sigs, brk, read_count, handle):
    cur = []
    for e in sigs:
        if cur and (brk(cur[-1], e) or (cur[-1][0] == 0 and cur[-1][1] == 0)):
            if len(cur) >= read_count and not (cur[-1][0] == 0 and cur[-1][1] == 0):
                handle(cur)
            cur = []
        cur.append(e)
    if len(cur) >= read_count and not (cur[-1][0] == 0 and cur[-1][1] == 0):
        handle(cur)


def run_task(task):
    """task = (svtype, chrom, sigs, reads_or_None, params-dict) -> (chrom, rows)"""
    svtype, chrom, sigs, reads, p = task
    rc, gt = p["min_support"], p["genotype"]
    out = []
    if svtype in ("DEL", "INS"):
        bias = p["max_cluster_bias_" + svtype]
        ratio = p["diff_ratio_merging_" + svtype]
        keep = min(p["remain_reads_ratio"], 1)
        _chain(sigs, lambda a, b: b[0] - a[0] > bias, rc,
               lambda c: indel_cluster(c, chrom, svtype, rc, ratio, min(rc, 5), keep, gt, out))
        if not gt:
            return chrom, out
        if reads is None:
            return chrom, []
        half = bias if svtype == "DEL" else 1000
        cover = sweep_cover([(max(c[7] - half, 0), c[7] + half) for c in out], reads)
        rows = []
        for c, cov in zip(out, cover):
            g = _gt_fields(cov, c[8])
            row = [c[0], c[1], str(c[2]), str(c[3]), str(c[4]), c[5], c[6], g[0], g[1], g[2], g[3], g[4], ",".join(c[8])]
            if svtype == "INS":
                row.append(c[9])
            rows.append(row)
        return chrom, rows
    if svtype == "DUP":
        bias = p["max_cluster_bias_DUP"]
        _chain(sigs, lambda a, b: b[0] - a[0] > bias, rc,
               lambda c: dup_cluster(c, chrom, rc, bias, p["min_size"], p["max_size"], gt, out))
        if not gt:
            return chrom, out
        if reads is None:
            return chrom, []
        wins = []
        for c in out:
            nb = min(bias, c[3] - c[2])
            wins.append((max(c[2] - nb / 2, 0), c[2] + nb / 2))
        for c in out:
            nb = min(bias, c[3] - c[2])
            wins.append((max(c[3] - nb / 2, 0), c[3] + nb / 2))
        cover = sweep_cover(wins, reads)
        rows = []
        for i, c in enumerate(out):
            g = _gt_fields(cover[i] | cover[i + len(out)], c[4])
            rows.append([c[0], c[1], str(c[2]), str(c[3] - c[2]), str(len(c[4])), g[0], g[1], g[2], g[3], g[4], ",".join(c[4])])
        return chrom, rows
    if svtype == "INV":
        bias = p["max_cluster_bias_INV"]
        _chain(sigs, lambda a, b: b[0] - a[0] > bias or b[1] - a[1] > bias or b[3] != a[3], rc,
               lambda c: inv_cluster(c, chrom, rc, bias, p["min_size"], p["max_size"], gt, out))
        if not gt:
            return chrom, out
        if reads is None:
            return chrom, []
        wins = [(max(c[2] - bias / 2, 0), c[2] + bias / 2) for c in out] + [(max(c[7] - bias / 2, 0), c[7] + bias / 2) for c in out]
        cover = sweep_cover(wins, reads)
        rows = []
        for i, c in enumerate(out):
            g = _gt_fields(cover[i] | cover[i + len(out)], c[6])
            rows.append([c[0], c[1], str(int(c[2])), str(int(c[3])), str(c[4]), g[0], g[1], c[5], g[2], g[3], g[4], ",".join(c[6])])
        return chrom, rows
    # TRA: signatures are (pos1, pos2, read, type, chr2); clusters never span chr2 or type
    bias = p["max_cluster_bias_TRA"]
    _chain(sigs, lambda a, b: b[0] - a[0] > bias or b[3] != a[3] or b[4] != a[4], rc,
           lambda c: tra_cluster(c, chrom, c[0][4], rc, p["diff_ratio_filtering_TRA"], bias, out))
    return chrom, out


def tasks_from_store(store, params, tasks=None):
    """Convert the flat store into per-task Python lists (done OUTSIDE any timed region)."""
    p = dict(params.__dict__)
    names = store.names
    out = []
    reads_cache = {}
    for svtype, chrom in (tasks or store.tasks()):
        beg, end = store.seg_index[(svtype, chrom)]
        a = store.a[beg:end].tolist(); b = store.b[beg:end].tolist(); ax = store.aux[beg:end].tolist()
        nm = names.take(store.read_id[beg:end])
        if svtype == "INS":
            if store.ins_seq is None and names.names is None:       # synthetic stores: 'ACGT' repeated to the aux length (SigStore.sequence)
                base = "ACGT" * (max(ax, default=0) // 4 + 1)
                sigs = list(zip(a, b, nm, [base[:n] for n in ax]))
            else:
                sigs = [(a[i], b[i], nm[i], store.sequence(beg + i)) for i in range(end - beg)]
        elif svtype == "INV":
            sigs = [(a[i], b[i], nm[i], store.strands[ax[i]]) for i in range(end - beg)]
        elif svtype == "TRA":
            sigs = [(a[i], b[i], nm[i], "ABCDXXXX"[ax[i] & 7], store.chroms[ax[i] >> 3]) for i in range(end - beg)]
        else:
            sigs = list(zip(a, b, nm))
        reads = None
        if params.genotype and svtype != "TRA" and store.has_reads(chrom):
            if chrom not in reads_cache:
                c = store.chroms.index(chrom)
                lo, hi = int(store.reads_off[c]), int(store.reads_off[c + 1])
                rn = names.take(store.r_id[lo:hi])
                reads_cache[chrom] = list(zip(store.r_start[lo:hi].tolist(), store.r_end[lo:hi].tolist(),
                                              store.r_primary[lo:hi].tolist(), rn))
            reads = reads_cache[chrom]
        out.append((svtype, chrom, sigs, reads, p))
    return out


def run_pool(task_list, processes):
    """The reference's phase 3: one map_async per task, results gathered per chromosome."""
    if processes <= 1:
        res = [run_task(t) for t in task_list]
    else:
        with Pool(processes=processes) as pool:
            handles = [pool.map_async(run_task, [t]) for t in task_list]
            res = [h.get()[0] for h in handles]
    return res


# fork-inherited task table: workers index into it, so the timed region of run_pool_forked holds the
# compute and the rows travelling back (as in the reference, whose workers receive only small argument
# tuples) but not the shipping of signature lists through pipes.
_TASKS = None


def _run_indexed(i):
    return run_task(_TASKS[i])


def run_pool_forked(task_list, processes):
    global _TASKS
    import multiprocessing as mp
    _TASKS = task_list
    ctx = mp.get_context("fork")
    with ctx.Pool(processes=processes) as pool:
        handles = [pool.map_async(_run_indexed, [i]) for i in range(len(task_list))]
        res = [h.get()[0] for h in handles]
    _TASKS = None
    return res


# ---------------------------------------------------------------------------------------- the five task callables, on the reference's files
# The reference's workers receive the argument tuples of main script :1117-1188 and read their task from `<TYPE>.pickle` /
# `reads.pickle` at the offsets of `sigs_index` (INDEL:52-58, 445-448; DUP:25-27; INV:42-44; TRA:36-38).  These five do the
# same for the restatement, so that bench.py's `mode1_stage` leg times both sides on the SAME files: both pay `pickle.load`.
def _block(work_dir, name, offset):
    import pickle
    with open("%s%s.pickle" % (work_dir, name), "rb") as f:
        f.seek(offset)
        return pickle.load(f)


def _reads_of(work_dir, chrom, sigs_index, genotype):
    if not genotype or chrom not in sigs_index.get("reads", {}):
        return None
    return _block(work_dir, "reads", sigs_index["reads"][chrom])       # (start, end, is_primary, read, chr): a prefix of what sweep_cover reads


def _indel_task(args, svtype):
    (work_dir, chrom, _t, read_count, ratio, bias, _msr, _bam, genotype, gt_round, keep, sigs_index) = args
    if chrom not in sigs_index[svtype]:
        return chrom, []
    sigs = [(int(e[0]),) + tuple(e[1:4]) for e in _block(work_dir, svtype, sigs_index[svtype][chrom])]     # int(pos): INDEL:271
    p = {"min_support": read_count, "genotype": genotype, "max_cluster_bias_" + svtype: bias, "diff_ratio_merging_" + svtype: ratio,
         "remain_reads_ratio": keep, "gt_round": gt_round}
    return run_task((svtype, chrom, sigs, _reads_of(work_dir, chrom, sigs_index, genotype), p))


def ref_run_del(args):
    return _indel_task(args, "DEL")


def ref_run_ins(args):
    return _indel_task(args, "INS")


def ref_run_inv(args):
    work_dir, chrom, _t, read_count, bias, min_size, _bam, genotype, max_size, gt_round, sigs_index = args
    if chrom not in sigs_index["INV"]:
        return chrom, []
    sigs = [(e[1], e[2], e[3], e[0]) for e in _block(work_dir, "INV", sigs_index["INV"][chrom])]          # (strand, pos1, pos2, read, ..)
    p = {"min_support": read_count, "genotype": genotype, "max_cluster_bias_INV": bias, "min_size": min_size, "max_size": max_size}
    return run_task(("INV", chrom, sigs, _reads_of(work_dir, chrom, sigs_index, genotype), p))


def ref_run_dup(args):
    work_dir, chrom, read_count, bias, min_size, _bam, genotype, max_size, gt_round, sigs_index = args
    if chrom not in sigs_index["DUP"]:
        return chrom, []
    sigs = [(int(e[0]), e[1], e[2]) for e in _block(work_dir, "DUP", sigs_index["DUP"][chrom])]
    p = {"min_support": read_count, "genotype": genotype, "max_cluster_bias_DUP": bias, "min_size": min_size, "max_size": max_size}
    return run_task(("DUP", chrom, sigs, _reads_of(work_dir, chrom, sigs_index, genotype), p))


def ref_run_tra(args):
    work_dir, chrom, read_count, overlap, bias, _bam, _genotype, gt_round, sigs_index = args
    if chrom not in sigs_index["TRA"]:
        return chrom, []
    sigs = [(e[1], e[3], e[4], e[0], e[2]) for e in _block(work_dir, "TRA", sigs_index["TRA"][chrom])]    # (type, pos1, chr2, pos2, read, ..)
    p = {"min_support": read_count, "genotype": False, "max_cluster_bias_TRA": bias, "diff_ratio_filtering_TRA": overlap}
    return run_task(("TRA", chrom, sigs, None, p))


REF_FNS = dict(DEL=ref_run_del, INS=ref_run_ins, INV=ref_run_inv, DUP=ref_run_dup, TRA=ref_run_tra)


def _noop(_):
    return None


def pool_startup_seconds(processes, n_tasks):
    """what the pool itself costs: the same fork Pool, `n_tasks` no-op tasks, torn down - no signature is looked at"""
    import multiprocessing as mp
    import time
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(processes=processes) as pool:
        hs = [pool.map_async(_noop, [i]) for i in range(n_tasks)]
        for h in hs:
            h.get()
    return time.perf_counter() - t0
