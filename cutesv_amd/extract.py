"""The CIGAR scan of cuteSV's extraction step on the GPU.

`cigar_signatures` is the face of `csv_cigar_signatures` (cutesv_amd/csrc/cigar.hip.h): the flat BAM-encoded CIGAR array
of a batch of reads -> the INS / DEL signatures parse_read + generate_combine_sigs (main script :606-655, :515-575) make
of them.  `candidates` turns the flat result into the reference's candidate tuples (the inserted sequence is cut out of
the reads' query sequences here: the bases never travel to the GPU).  `split_signatures` is the face of
`csv_split_signatures` (split.hip.h): organize_split_signal + analysis_split_read (:50-513) on the numbers of a batch of
primary alignments and SA-tag entries.  BAM decode, the SA text and everything else of the extraction stay in the Python
driver with pysam, as north_star has it: a driver would collect
`read.cigartuples`, `read.reference_start`, `read.mapq >= min_mapq and read.query_length >= min_read_len` for a task's
reads, make one call here, and extend candidate["INS"] / candidate["DEL"] with the result.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._lib import lib


class CigarIn(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("cig_off", C.c_void_p), ("cigar", C.c_void_p), ("ref_start", C.c_void_p), ("use", C.c_void_p),
                ("min_siglength", C.c_int32), ("flags", C.c_int32), ("merge_ins_threshold", C.c_int64), ("merge_del_threshold", C.c_int64),
                ("seg_ins", C.c_int32), ("seg_del", C.c_int32), ("read_base", C.c_int64), ("query_len", C.c_void_p)]


_OUT = [("ins_read", np.int32, "i"), ("ins_pos", np.int64, "i"), ("ins_len", np.int64, "i"), ("ins_piece0", np.int64, "i"), ("ins_npiece", np.int32, "i"),
        ("piece_qoff", np.int32, "p"), ("piece_len", np.int32, "p"), ("del_read", np.int32, "d"), ("del_pos", np.int64, "d"), ("del_len", np.int64, "d")]


class CigarOut(C.Structure):
    _fields_ = ([("cap_sig_ins", C.c_int64), ("cap_piece_ins", C.c_int64), ("cap_sig_del", C.c_int64),
                 ("n_sig_ins", C.c_int64), ("n_piece_ins", C.c_int64), ("n_sig_del", C.c_int64)]
                + [(n, C.c_void_p) for n, _, _ in _OUT] + [("ms_device", C.c_float), ("reserved", C.c_int32)])


def encode_cigars(cigartuples_per_read):
    """[[(op, oplen), ...] per read] (pysam's read.cigartuples) -> (cig_off int64[n + 1], cigar uint32[n_ops]) in the BAM
    encoding oplen << 4 | op"""
    # (pysam returns None for a record without a CIGAR - an unmapped mate that fetch() still yields; the reference skips
    # such a read through its mapq gate, main script :614, and carries on)
    cigartuples_per_read = [c if c is not None else () for c in cigartuples_per_read]
    lens = np.fromiter((len(c) for c in cigartuples_per_read), np.int64, len(cigartuples_per_read))
    off = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.empty(int(off[-1]), np.uint32)
    k = 0
    for c in cigartuples_per_read:
        for op, ln in c:
            flat[k] = (int(ln) << 4) | int(op)
            k += 1
    return off, flat


def _run(fn, handle, cig_off, cigar, ref_start, use, min_siglength, merge_ins_threshold, merge_del_threshold, check, pool=None, host_outputs=True):
    cig_off = np.ascontiguousarray(cig_off, np.int64); cigar = np.ascontiguousarray(cigar, np.uint32)
    ref_start = np.ascontiguousarray(ref_start, np.int64)
    use = None if use is None else np.ascontiguousarray(use, np.uint8)
    n = len(ref_start)
    cin = CigarIn(n_reads=n, cig_off=cig_off.ctypes.data, cigar=cigar.ctypes.data if len(cigar) else None, ref_start=ref_start.ctypes.data,
                  use=None if use is None else use.ctypes.data, min_siglength=int(min_siglength),
                  merge_ins_threshold=int(merge_ins_threshold), merge_del_threshold=int(merge_del_threshold))
    qlen = None
    if pool is not None:                                  # CSV_CG_TO_POOL: the signatures also become rows of the context's pool
        qlen = None if pool.get("query_len") is None else np.ascontiguousarray(pool["query_len"], np.int32)
        cin.flags = _abi.CG_TO_POOL
        cin.seg_ins = int(pool["seg_ins"]); cin.seg_del = int(pool["seg_del"]); cin.read_base = int(pool["read_base"])
        cin.query_len = None if qlen is None else qlen.ctypes.data
    caps = dict(i=max(16, n // 4), p=max(16, n // 4), d=max(16, n // 4))
    for _ in range(2):
        if host_outputs or pool is None:
            arrs = {name: np.zeros(caps[k], dt) for name, dt, k in _OUT}
            cout = CigarOut(cap_sig_ins=caps["i"], cap_piece_ins=caps["p"], cap_sig_del=caps["d"], **{k: v.ctypes.data for k, v in arrs.items()})
        else:                                             # pool only: nothing but the counts comes back
            arrs = {name: np.zeros(0, dt) for name, dt, k in _OUT}
            big = int(cig_off[-1]) + 1
            cout = CigarOut(cap_sig_ins=big, cap_piece_ins=big, cap_sig_del=big)
        rc = fn(handle, C.byref(cin), C.byref(cout)) if handle is not None else fn(C.byref(cin), C.byref(cout))
        if rc == _abi.E_CAPACITY:
            caps = dict(i=int(cout.n_sig_ins) + 1, p=int(cout.n_piece_ins) + 1, d=int(cout.n_sig_del) + 1)
            continue
        check(rc)
        cut = dict(i=int(cout.n_sig_ins), p=int(cout.n_piece_ins), d=int(cout.n_sig_del))
        out = {name: arrs[name][:cut[k]] for name, _, k in _OUT}
        out["n_sig_ins"], out["n_sig_del"] = cut["i"], cut["d"]
        out["ms_device"] = float(cout.ms_device)
        return out
    raise RuntimeError("csv_cigar_signatures: capacity retry failed")


def cigar_signatures(ctx, cig_off, cigar, ref_start, use=None, min_siglength=10, merge_ins_threshold=100, merge_del_threshold=0, pool=None, host_outputs=True):
    """flat CIGARs of a batch of reads -> dict of the signature arrays of csv_cigar_out (defaults: cuteSV_Description.py:123-152).
    pool = dict(seg_ins, seg_del, read_base, query_len=None): the signatures ALSO become rows of the context's device-resident
    pool (CSV_CG_TO_POOL; INS rows in segment seg_ins, DEL rows in seg_del, read index = read_base + index in this batch), in the
    order INS then DEL - what `rebuild.rebuild_pool` sorts without the rows ever crossing PCIe; host_outputs=False (with a pool):
    the arrays of the result stay empty, only the counts come back."""
    L = lib()
    L.csv_cigar_signatures.restype = C.c_int
    L.csv_cigar_signatures.argtypes = [C.c_void_p, C.POINTER(CigarIn), C.POINTER(CigarOut)]
    return _run(L.csv_cigar_signatures, ctx._h, cig_off, cigar, ref_start, use, min_siglength, merge_ins_threshold, merge_del_threshold, ctx._check, pool=pool,
                host_outputs=host_outputs)


def candidates(sig, read_names, query_sequences, chrom):
    """signature arrays -> the reference's candidate tuples (main script :520-575):
    INS (pos, len, read, seq, "INS", chr), DEL (pos, len, read, "DEL", chr), in read order"""
    ins, dele = [], []
    qo, ql = sig["piece_qoff"].tolist(), sig["piece_len"].tolist()
    for r, pos, ln, p0, npc in zip(sig["ins_read"].tolist(), sig["ins_pos"].tolist(), sig["ins_len"].tolist(),
                                    sig["ins_piece0"].tolist(), sig["ins_npiece"].tolist()):
        q = query_sequences[r]
        seq = "".join(str(q[qo[p] : qo[p] + ql[p]]) for p in range(p0, p0 + npc))      # read.query_sequence[shift - oplen : shift] (:639-640)
        ins.append((pos, ln, read_names[r], seq, "INS", chrom))
    for r, pos, ln in zip(sig["del_read"].tolist(), sig["del_pos"].tolist(), sig["del_len"].tolist()):
        dele.append((pos, ln, read_names[r], "DEL", chrom))
    return ins, dele


# ------------------------------------------------------------------------------------ split reads (SA tag)
class SplitIn(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("ent_off", C.c_void_p), ("read_len", C.c_void_p), ("c0", C.c_void_p), ("c1", C.c_void_p),
                ("f0", C.c_void_p), ("f1", C.c_void_p), ("chr", C.c_void_p), ("mapq", C.c_void_p), ("strand", C.c_void_p), ("primary", C.c_void_p),
                ("sv_size", C.c_int64), ("max_size", C.c_int64), ("min_mapq", C.c_int32), ("max_split_parts", C.c_int32),
                ("flags", C.c_int32), ("pool_seg_base", C.c_int32 * 5), ("read_base", C.c_int64), ("query_len", C.c_void_p)]


_SPLIT_OUT = [("kind", np.uint8), ("read", np.int32), ("chr", np.int32), ("aux", np.int32), ("a", np.int64), ("b", np.int64), ("c", np.int64), ("d", np.int64)]


class SplitOut(C.Structure):
    _fields_ = [("cap", C.c_int64), ("n", C.c_int64)] + [(n, C.c_void_p) for n, _ in _SPLIT_OUT] + [("ms_device", C.c_float), ("reserved", C.c_int32)]


_CIGAR_RE = None


def clip_and_span(cigar_string):
    """acquire_clip_pos (main script :466-481) on the CIGAR text of an SA entry: [leading soft clip, trailing soft clip,
    reference span over M / D / = / X]"""
    global _CIGAR_RE
    if _CIGAR_RE is None:
        import re
        _CIGAR_RE = re.compile(r"(\d+)([MIDNSHP=XB])")
    ops = _CIGAR_RE.findall(cigar_string)
    first = int(ops[0][0]) if ops and ops[0][1] == "S" else 0
    last = int(ops[-1][0]) if ops and ops[-1][1] == "S" else 0
    return first, last, sum(int(n) for n, o in ops if o in "MD=X")


def encode_split_reads(reads, chrom_rank):
    """reads: [(primary_info or [], "chr,pos,strand,CIGAR,mapq,NM;..." SA tag value, query_length)] per read, as parse_read
    (main script :657-679) hands them to organize_split_signal; chrom_rank: {name: rank in Python string order}.
    -> dict of the flat csv_split_in arrays.  This is the text side of the step (the SA tag is a string): it stays here."""
    ent_off = [0]
    cols = {k: [] for k in ("c0", "c1", "f0", "f1", "chr", "mapq", "strand", "primary")}
    read_len = []
    for primary, sa, qlen in reads:
        read_len.append(int(qlen))
        if len(primary) > 0:
            for k, v in zip(("c0", "c1", "f0", "f1"), primary[:4]):
                cols[k].append(int(v))
            cols["chr"].append(chrom_rank[primary[4]]); cols["strand"].append(0 if primary[5] == "+" else 1)
            cols["mapq"].append(0); cols["primary"].append(1)
        for entry in sa.split(";")[:-1]:                       # (:676)
            seq = entry.split(",")
            first, last, span = clip_and_span(seq[3])
            cols["c0"].append(first); cols["c1"].append(last); cols["f0"].append(int(seq[1]) - 1); cols["f1"].append(span)      # (:497)
            cols["chr"].append(chrom_rank[seq[0]]); cols["strand"].append(0 if seq[2] == "+" else 1)
            cols["mapq"].append(int(seq[4])); cols["primary"].append(0)
        ent_off.append(len(cols["c0"]))
    dt = dict(c0=np.int64, c1=np.int64, f0=np.int64, f1=np.int64, chr=np.int32, mapq=np.int32, strand=np.uint8, primary=np.uint8)
    out = {k: np.asarray(v, dt[k]) for k, v in cols.items()}
    out["ent_off"] = np.asarray(ent_off, np.int64); out["read_len"] = np.asarray(read_len, np.int64)
    return out


def _run_split(fn, handle, enc, sv_size, min_mapq, max_split_parts, max_size, check, pool=None):
    a = {k: np.ascontiguousarray(v) for k, v in enc.items()}
    n = len(a["read_len"])
    ptr = lambda x: x.ctypes.data if len(x) else None                      # noqa: E731
    sin = SplitIn(n_reads=n, ent_off=a["ent_off"].ctypes.data, read_len=ptr(a["read_len"]), c0=ptr(a["c0"]), c1=ptr(a["c1"]), f0=ptr(a["f0"]),
                  f1=ptr(a["f1"]), chr=ptr(a["chr"]), mapq=ptr(a["mapq"]), strand=ptr(a["strand"]), primary=ptr(a["primary"]),
                  sv_size=int(sv_size), max_size=int(max_size), min_mapq=int(min_mapq), max_split_parts=int(max_split_parts))
    qlen = None
    if pool is not None:                                  # CSV_CG_TO_POOL: the candidates also become rows of the context's pool
        qlen = None if pool.get("query_len") is None else np.ascontiguousarray(pool["query_len"], np.int32)
        sin.flags = _abi.CG_TO_POOL
        sin.pool_seg_base = (C.c_int32 * 5)(*[int(x) for x in pool["seg_base"]])
        sin.read_base = int(pool["read_base"])
        sin.query_len = None if qlen is None else qlen.ctypes.data
    cap = max(16, 2 * n)
    for _ in range(2):
        arrs = {name: np.zeros(cap, dt) for name, dt in _SPLIT_OUT}
        sout = SplitOut(cap=cap, **{k: v.ctypes.data for k, v in arrs.items()})
        rc = fn(handle, C.byref(sin), C.byref(sout)) if handle is not None else fn(C.byref(sin), C.byref(sout))
        if rc == _abi.E_CAPACITY:
            cap = int(sout.n) + 1
            continue
        check(rc)
        out = {name: arrs[name][:int(sout.n)] for name, _ in _SPLIT_OUT}
        out["ms_device"] = float(sout.ms_device)
        return out
    raise RuntimeError("csv_split_signatures: capacity retry failed")


def split_signatures(ctx, enc, sv_size=30, min_mapq=20, max_split_parts=7, max_size=100000, pool=None):
    """flat split-read entries of a batch of reads (encode_split_reads) -> dict of the candidate arrays of csv_split_out
    (defaults: cuteSV_Description.py: --min_size 30, --min_mapq 20, --max_split_parts 7, --max_size 100000).
    pool = dict(seg_base=[segment of chromosome rank 0 for kind DEL, INS, DUP, INV, TRA], read_base, query_len=None): the
    candidates ALSO become rows of the context's device-resident pool (`pool_rows_of_split` is the same mapping on the host)."""
    L = lib()
    L.csv_split_signatures.restype = C.c_int
    L.csv_split_signatures.argtypes = [C.c_void_p, C.POINTER(SplitIn), C.POINTER(SplitOut)]
    return _run_split(L.csv_split_signatures, ctx._h, enc, sv_size, min_mapq, max_split_parts, max_size, ctx._check, pool=pool)


def pool_rows_of_split(sig, seg_base, read_base, query_len):
    """the rows csv_split_signatures appends to the pool for the candidates `sig` (host restatement, for tests and for callers
    that build the rows themselves): dict(seg, a, b, read, aux) in candidate order"""
    kind = sig["kind"].astype(np.int64); aux = sig["aux"].astype(np.int64)
    a = np.where((kind == 1) & ((aux & 2) != 0), sig["a"] >> 1, sig["a"])
    ql = np.asarray(query_len, np.int64)[sig["read"]]
    lo, hi = np.minimum(sig["c"], ql), np.minimum(sig["d"], ql)
    ax = np.where(kind == 1, np.maximum(hi - lo, 0), np.where(kind == 3, aux, np.where(kind == 4, sig["c"] * 8 + aux, 0)))
    return dict(seg=(np.asarray(seg_base, np.int64)[kind] + sig["chr"]).astype(np.int32), a=a.astype(np.int64), b=sig["b"].astype(np.int64),
                read=(read_base + sig["read"]).astype(np.int32), aux=ax.astype(np.int32))


_COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def split_candidates(sig, read_names, queries, chrom_names):
    """candidate arrays -> the reference's candidate tuples per SV type (main script :50-464), in the order of its five lists.
    queries[r] = the query string parse_read passes for read r (already reverse-complemented for reverse-strand reads, :673);
    the inserted sequences are cut out of it here."""
    cand = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
    rc_cache = {}
    for kind, r, ch, aux, a, b, c, d in zip(sig["kind"].tolist(), sig["read"].tolist(), sig["chr"].tolist(), sig["aux"].tolist(),
                                            sig["a"].tolist(), sig["b"].tolist(), sig["c"].tolist(), sig["d"].tolist()):
        name, chrom = read_names[r], chrom_names[ch]
        if kind == 0:
            cand["DEL"].append((a, b, name, "DEL", chrom))
        elif kind == 1:
            q = queries[r]
            if aux & 1:
                if r not in rc_cache:
                    rc_cache[r] = str(q).translate(_COMP)[::-1]
                q = rc_cache[r]
            cand["INS"].append((a / 2 if aux & 2 else a, b, name, str(q[c:d]), "INS", chrom))
        elif kind == 2:
            cand["DUP"].append((a, b, name, "DUP", chrom))
        elif kind == 3:
            cand["INV"].append(("--" if aux else "++", a, b, name, "INV", chrom))
        else:
            cand["TRA"].append(("ABCD"[aux], a, chrom_names[c], b, name, "TRA", chrom))
    return cand


# ------------------------------------------------------------------------------------ parse_read for a batch of reads
def parse_reads(reads, chrom, chrom_rank, sv_size, min_mapq, max_split_parts, min_read_len, min_siglength, merge_del_threshold,
                merge_ins_threshold, max_size, cigar_fn, split_fn):
    """What calling the reference's parse_read (main script :606-681) on every read of `reads`, in order, appends to
    candidate["DEL" | "INS" | "DUP" | "INV" | "TRA"] - with the CIGAR scan and the split-read analysis done per BATCH by
    `cigar_fn(cig_off, cigar, ref_start, use, **kw)` / `split_fn(enc, **kw)` (a context's cigar_signatures /
    split_signatures; the tests also pass the oracle's).  reads: pysam.AlignedSegment-like objects (query_length, flag,
    mapq, reference_start, reference_end, cigartuples, query_sequence, query_name, get_tags()); chrom_rank: {chromosome
    name: rank in Python string order} over every name an SA tag can mention.

    Everything that is text or per-read bookkeeping stays here, as the reference has it: the read-length gate (:607), the
    flag classes (:613), the clip lengths that make primary_info (:619-668), the SA tag (:671-679)."""
    cand = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
    keep = [r for r in reads if r.query_length >= min_read_len]                                    # (:607)
    if not keep:
        return cand
    cig_off, cigar = encode_cigars([r.cigartuples for r in keep])
    ref_start = np.fromiter((r.reference_start for r in keep), np.int64, len(keep))
    use = np.fromiter((1 if r.mapq >= min_mapq else 0 for r in keep), np.uint8, len(keep))          # (:614)
    sig = cigar_fn(cig_off, cigar, ref_start, use, min_siglength=min_siglength, merge_ins_threshold=merge_ins_threshold,
                   merge_del_threshold=merge_del_threshold)
    names = [r.query_name for r in keep]
    c_ins, c_del = candidates(sig, names, [r.query_sequence for r in keep], chrom)
    # reads with an SA tag on a primary record (flag 0 / 16, :657): their segments go through the split-read analysis
    sp_reads, sp_idx, sp_query = [], [], []
    for i, r in enumerate(keep):
        if r.flag not in (0, 16):
            continue
        sa = [v for k, v in r.get_tags() if k == "SA"]
        if not sa:
            continue
        primary = []
        if r.mapq >= min_mapq:
            ct = r.cigartuples or ((0, 0),)                                                         # (no CIGAR: no clips)
            left = ct[0][1] if ct[0][0] in (4, 5) else 0                                            # soft clip, or the hard clip that replaces it (:619-652)
            right = ct[-1][1] if ct[-1][0] in (4, 5) else 0
            primary = ([left, r.query_length - right, r.reference_start, r.reference_end, chrom, "+"] if r.flag == 0 else
                       [right, r.query_length - left, r.reference_start, r.reference_end, chrom, "-"])
        q = r.query_sequence if r.flag == 0 else str(r.query_sequence).translate(_COMP)[::-1]       # (:673-675)
        for tag in sa:                                                                              # (one call per SA tag, :671)
            sp_reads.append((primary, tag, r.query_length)); sp_idx.append(i); sp_query.append(q)
    s_cand = {t: [] for t in cand}
    s_read = {t: [] for t in cand}
    if sp_reads:
        enc = encode_split_reads(sp_reads, chrom_rank)
        ssig = split_fn(enc, sv_size=sv_size, min_mapq=min_mapq, max_split_parts=max_split_parts, max_size=max_size)
        s_cand = split_candidates(ssig, [names[i] for i in sp_idx], sp_query, sorted(chrom_rank, key=chrom_rank.get))
        kind_name = ("DEL", "INS", "DUP", "INV", "TRA")
        for k, rd in zip(ssig["kind"].tolist(), ssig["read"].tolist()):
            s_read[kind_name[k]].append(sp_idx[rd])
    # per type: read order; inside a read the CIGAR signatures come first (:656-657 before :671-679)
    for t, c_list, c_reads in (("INS", c_ins, sig["ins_read"].tolist()), ("DEL", c_del, sig["del_read"].tolist())):
        out, j = [], 0
        sl, sr = s_cand[t], s_read[t]
        for x, rd in zip(c_list, c_reads):
            while j < len(sl) and sr[j] < rd:
                out.append(sl[j]); j += 1
            out.append(x)
        out.extend(sl[j:])
        cand[t] = out
    for t in ("DUP", "INV", "TRA"):
        cand[t] = s_cand[t]
    return cand


# ------------------------------------------------------------------------------------ single_pipe: one extraction task
def single_pipe(alignments, chrom, task_start, chrom_rank, sv_size, min_mapq, max_split_parts, min_read_len, min_siglength, merge_del_threshold,
                merge_ins_threshold, max_size, cigar_fn, split_fn, bed_regions=None):
    """What the reference's single_pipe (main script :697-743) pickles for one task region: the five candidate lists of the
    reads that pass its gates and the reads table rows `(start, end, is_primary, name, chr)` (:709-733).

    alignments: what `samfile.fetch(chr, task[1], task[2])` yields, in that order.  Gates, as the reference applies them:
    secondary records (flag 256 / 272) are skipped (:711); a read belongs to the task in which it STARTS
    (`reference_start >= task[1]`, :725) and, with --include_bed, must overlap one of the chromosome's regions (:715-723);
    such a read goes through parse_read (here: one batched call, `parse_reads`), and enters the reads table when its mapq
    passes (:729-733) - whatever parse_read did with it (a read shorter than min_read_len still counts as coverage)."""
    recs = [r for r in alignments if r.flag != 256 and r.flag != 272]
    if recs:
        start = np.fromiter((r.reference_start for r in recs), np.int64, len(recs))
        keep = start >= task_start
        if bed_regions is not None:
            end = np.fromiter((r.reference_end for r in recs), np.int64, len(recs))
            in_bed = np.zeros(len(recs), bool)
            for b0, b1 in bed_regions:                       # not (pos_end <= b0 or pos_start >= b1)
                in_bed |= (end > b0) & (start < b1)
            keep &= in_bed
        recs = [r for r, k in zip(recs, keep.tolist()) if k]
    cand = parse_reads(recs, chrom, chrom_rank, sv_size, min_mapq, max_split_parts, min_read_len, min_siglength, merge_del_threshold,
                       merge_ins_threshold, max_size, cigar_fn, split_fn)
    reads_info = [(r.reference_start, r.reference_end, 1 if r.flag in (0, 16) else 0, r.query_name, chrom) for r in recs if r.mapq >= min_mapq]
    return cand, reads_info
