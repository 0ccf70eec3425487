"""Sharding of the clustering stage over the GPUs of a node.

(chromosome, type) segments are independent — the reference already treats them as separate pool
tasks (main script :1116-1189) and TRA signatures of a (chr1, chr2) pair live entirely in chr1's block
(:801) — so there is NO data-path collective: each rank (one process per GPU) clusters its own
units and the host concatenates the per-chromosome row lists exactly as main script :1191-1197
does.

A unit of work is a chromosome, or - when whole chromosomes do not balance (chr1 is ~8 % of a human genome: at
8 GPUs it alone is 2/3 of a rank's share) - a PIECE of one: every DEL / INS / INV / DUP segment of the chromosome is cut
at a gap of more than max_cluster_bias between neighbouring signatures near the same genomic coordinate.  A chained
cluster cannot span such a gap (INDEL:61, DUP:35, INV:56), so the pieces' rows, concatenated in order, are exactly the
segment's rows; the reference's own extraction windows cut chromosomes the same way (main script :1022-1044).  TRA
segments (sorted by mate chromosome first, a few thousand signatures per genome) are never cut.  All types of one unit
stay on one rank, and a rank is shipped only the reads that can touch its units' genotyping windows.
"""
import numpy as np

from . import _abi
from .columns import TYPES

CUT_TYPES = ("DEL", "INS", "INV", "DUP")


def chromosome_cost(store, chrom, genotype=False):
    n = sum(e - b for (t, c), (b, e) in store.seg_index.items() if c == chrom)
    if genotype and store.reads_off is not None:
        i = store.chroms.index(chrom)
        n += int(store.reads_off[i + 1] - store.reads_off[i])
    return n


def _bias_of(svtype, params):
    return {"DEL": params.max_cluster_bias_DEL, "INS": params.max_cluster_bias_INS, "INV": params.max_cluster_bias_INV,
            "DUP": params.max_cluster_bias_DUP}[svtype]


def _runs(store, svtype, b, e):
    """position-sorted runs of a segment's rows: the whole range, except INV, whose rows are ordered (strand, pos) (main script
    :792): one run per strand (the chain breaks at a strand change anyway, INV:56)"""
    if svtype != "INV" or e - b < 2:
        return [(b, e)]
    cut = (b + 1 + np.flatnonzero(store.aux[b + 1:e] != store.aux[b:e - 1])).tolist()
    return list(zip([b] + cut, cut + [e]))


def _cuts(store, chrom, params, n_pieces):
    """coordinates x_1 < ... < x_{k-1} that cut `chrom` into ~equal pieces: piece i holds, of every cuttable segment, the
    signatures with x_i <= pos < x_{i+1}; a coordinate is admissible for a segment when the signatures on either side of it
    are more than the segment's max_cluster_bias apart (INV: two neighbours with a position gap that large break the chain
    whatever their strands are, INV:56).  Returns a sorted list (possibly shorter than asked)."""
    segs = [(t, r) for t in CUT_TYPES if (t, chrom) in store.seg_index for r in _runs(store, t, *store.seg_index[(t, chrom)])]
    if not segs or n_pieces < 2:
        return []
    pos_all = np.sort(np.concatenate([store.a[b:e] for _, (b, e) in segs]))
    out = []
    for i in range(1, n_pieces):
        want = int(pos_all[len(pos_all) * i // n_pieces])
        best = None
        # the largest admissible coordinate <= want + slack for every segment at once: walk candidate gaps of the densest
        # segment near `want` and test the others
        t0, (b0, e0) = max(segs, key=lambda s: s[1][1] - s[1][0])
        a0 = store.a[b0:e0]
        k = int(np.searchsorted(a0, want))
        for step in range(0, 20000):
            for j in (k + step, k - step):
                if j <= 0 or j >= len(a0):
                    continue
                if int(a0[j]) - int(a0[j - 1]) <= _bias_of(t0, params):
                    continue
                x = int(a0[j])                                   # cut in front of signature j of the densest segment
                ok = True
                for t, (b, e) in segs:
                    if (t, (b, e)) == (t0, (b0, e0)):
                        continue
                    a = store.a[b:e]
                    m = int(np.searchsorted(a, x))
                    if 0 < m < len(a) and int(a[m]) - int(a[m - 1]) <= _bias_of(t, params):
                        ok = False
                        break
                if ok:
                    best = x
                    break
            if best is not None:
                break
        if best is not None and (not out or best > out[-1]):
            out.append(best)
    return out


def plan(store, world_size, params=None, genotype=False, max_imbalance=0.03):
    """-> list (per rank) of units (chrom, piece, n_pieces, lo, hi): the signatures of chromosome `chrom` with lo <= pos < hi
    (lo / hi None: unbounded).  Longest-processing-time-first over chromosomes; while the heaviest rank exceeds the mean by
    more than max_imbalance, the heaviest uncut chromosome on it is cut into pieces of about a quarter of a rank's share.
    Deterministic: every rank computes the same table without communicating."""
    chroms = sorted({c for (_, c) in store.seg_index})
    cost = {c: chromosome_cost(store, c, genotype) for c in chroms}
    total = sum(cost.values())
    pieces = {c: [] for c in chroms}                              # cut coordinates per chromosome

    def units_of(c):
        xs = pieces[c]
        if not xs:
            return [(c, 0, 1, None, None, cost[c])]
        bounds = [None] + xs + [None]
        n_sig = {}
        out = []
        for i in range(len(xs) + 1):
            lo, hi = bounds[i], bounds[i + 1]
            n = 0
            for t in CUT_TYPES:
                if (t, c) in store.seg_index:
                    for b, e in _runs(store, t, *store.seg_index[(t, c)]):
                        a = store.a[b:e]
                        n += int(np.searchsorted(a, hi) if hi is not None else len(a)) - int(np.searchsorted(a, lo) if lo is not None else 0)
            if i == 0 and ("TRA", c) in store.seg_index:
                b, e = store.seg_index[("TRA", c)]
                n += e - b
            if genotype and store.reads_off is not None:          # reads are shipped per piece (roughly its share of the block)
                ci = store.chroms.index(c)
                n += int(store.reads_off[ci + 1] - store.reads_off[ci]) // (len(xs) + 1)
            out.append((c, i, len(xs) + 1, lo, hi, n))
        return out

    def lpt():
        units = sorted((u for c in chroms for u in units_of(c)), key=lambda u: (-u[5], u[0], u[1]))
        load = [0] * world_size
        out = [[] for _ in range(world_size)]
        for u in units:
            r = min(range(world_size), key=lambda i: (load[i], i))
            out[r].append(u)
            load[r] += u[5]
        return out, load

    out, load = lpt()
    if params is not None and world_size > 1:
        for _ in range(4 * world_size):
            mean = total / world_size
            if max(load) <= mean * (1 + max_imbalance):
                break
            r = load.index(max(load))
            cand = [u for u in out[r] if u[2] == 1 and u[5] > mean * 0.3]
            if not cand:
                cand = [u for r2 in range(world_size) for u in out[r2] if u[2] == 1 and u[5] > mean * 0.3]
            if not cand:
                break
            c = max(cand, key=lambda u: u[5])[0]
            k = max(2, int(np.ceil(cost[c] / (mean * 0.25))))
            xs = _cuts(store, c, params, k)
            if not xs:
                cost[c] = -cost[c]                                 # (no admissible cut: never tried again)
                break
            pieces[c] = xs
            out, load = lpt()
    return [[u[:5] for u in us] for us in out]


def assign(store, world_size, genotype=False):
    """whole chromosomes only (the plan without cuts) -> list (per rank) of chromosomes"""
    return [[u[0] for u in us] for us in plan(store, world_size, None, genotype)]


def tasks_of_rank(store, rank, world_size, genotype=False, types=TYPES):
    mine = set(assign(store, world_size, genotype)[rank])
    return [(t, c) for (t, c) in store.tasks(types=types) if c in mine]


def rank_batch(store, params, units):
    """The batch of one rank: (segments, keys, reads kwargs).  keys[i] = (type, chrom, order) of segment i; the segments of a
    piece are sub-ranges of the store's segments with the segment's own scalars - ONE per position-sorted run: an INV piece is
    a sub-range of every strand's run, keyed strand-major (order = run * n_pieces + piece) so that merge_rows puts the rows back
    in the reference's order (all of '++', then all of '--'); the reads table holds, per chromosome, only the reads that can
    reach a genotyping window of the rank's pieces (windows lie within gt_bias of a signature position, pair types also around
    pos2)."""
    segs, keys = [], []
    need = {}                                                     # chrom -> [lo, hi] coordinate range the rank genotypes in
    for t in TYPES:
        for (c, piece, n_pieces, lo, hi) in units:
            if (t, c) not in store.seg_index:
                continue
            if t == "TRA" and piece != 0:
                continue
            rec0 = store.segment(t, c, params)
            b0, e0 = int(rec0["sig_begin"]), int(rec0["sig_end"])
            runs = _runs(store, t, b0, e0) if (t != "TRA" and n_pieces > 1) else [(b0, e0)]
            for ri, (rb, re_) in enumerate(runs):
                rec = rec0.copy()
                b, e = rb, re_
                if t != "TRA" and n_pieces > 1:
                    a = store.a[rb:re_]
                    b = rb + (int(np.searchsorted(a, lo)) if lo is not None else 0)
                    e = rb + (int(np.searchsorted(a, hi)) if hi is not None else re_ - rb)
                    rec["sig_begin"], rec["sig_end"] = b, e
                if e <= b:
                    continue
                segs.append(rec); keys.append((t, c, ri * n_pieces + piece))
                if rec["genotype"] and t != "TRA":
                    x0 = int(store.a[b:e].min()); x1 = int(max(store.a[b:e].max(), store.b[b:e].max() if t in ("DUP", "INV") else 0))
                    g = int(rec["gt_bias"]) + 2
                    cur = need.get(c)
                    need[c] = [min(x0 - g, cur[0]) if cur else x0 - g, max(x1 + g, cur[1]) if cur else x1 + g]
                elif rec["genotype"]:
                    need[c] = [-(1 << 62), 1 << 62]               # TRA windows sit on two chromosomes: keep whole blocks
    kw = {}
    if store.reads_off is not None and any(s["genotype"] for s in segs):
        tra = any(s["genotype"] and s["svtype"] == _abi.TRA for s in segs)
        off, cols = [0], dict(r_start=[], r_end=[], r_primary=[], r_id=[])
        for ci, c in enumerate(store.chroms):
            lo_i, hi_i = int(store.reads_off[ci]), int(store.reads_off[ci + 1])
            if tra:
                sel = slice(lo_i, hi_i)
            elif c in need:
                rs, re_ = store.r_start[lo_i:hi_i], store.r_end[lo_i:hi_i]
                sel = lo_i + np.flatnonzero((rs <= need[c][1]) & (re_ >= need[c][0]))
            else:
                sel = slice(lo_i, lo_i)
            for k in cols:
                cols[k].append(getattr(store, k)[sel])
            off.append(off[-1] + len(cols["r_start"][-1]))
        kw = dict(reads_off=np.array(off, np.int64), **{k: np.concatenate(v) for k, v in cols.items()})
        if tra:
            kw["contig_len"] = store.contig_len
    return np.array(segs, dtype=_abi.SEGMENT_DTYPE) if segs else np.zeros(0, _abi.SEGMENT_DTYPE), keys, kw


def host_batch(store, params, units, pin=None):
    """-> (HostBatch of the rank's units, keys).  The signature columns are the store's (its int32 twins when it keeps them:
    SigStore.pinned()); the rank's reads subset is new memory: `pin` (engine.pinned_copy) page-locks it."""
    segs, keys, kw = rank_batch(store, params, units)
    nw = store.narrow or {}
    if "r_start" in kw:
        if "r_start" in nw:                               # (the store's reads fit int32: so does every subset)
            kw["r_start"] = kw["r_start"].astype(np.int32); kw["r_end"] = kw["r_end"].astype(np.int32)
        if pin is not None:
            kw = {k: (pin(v) if isinstance(v, np.ndarray) and k != "contig_len" else v) for k, v in kw.items()}
    return _abi.HostBatch(segs, nw.get("a", store.a), nw.get("b", store.b), store.read_id, store.aux, n_chrom=len(store.chroms), **kw), keys


def merge_rows(per_rank):
    """[{(type, chrom, order): rows}] of all ranks -> {chrom: rows} in the order main_ctrl concatenates task results
    (main script :1191-1197: DEL, INS, INV, DUP, TRA per chromosome), the pieces of a segment in the segment's own row order
    (coordinate order; INV: strand-major, rank_batch)."""
    every = {}
    for d in per_rank:
        for k, rows in d.items():
            assert k not in every, "unit %r on two ranks" % (k,)
            every[k] = rows
    out = {}
    for t in TYPES:
        for (tt, c, piece) in sorted(k for k in every if k[0] == t):
            out.setdefault(c, []).extend(every[(tt, c, piece)])
    return out


def merge_results(per_rank):
    """{chr: rows} dicts of all ranks -> one dict (whole chromosomes, disjoint across ranks)."""
    out = {}
    for d in per_rank:
        for c, rows in d.items():
            assert c not in out, "chromosome %s on two ranks" % c
            out[c] = rows
    return out
