"""Shared test helpers: golden-fixture loading, row canonicalisation, engine adapters."""
import gzip
import hashlib
import json
import os

import numpy as np

from cutesv_amd import _abi, rows as rows_mod
from cutesv_amd.columns import SigStore, Params, NameTable

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
READS_FIELD = {"DEL": 12, "INS": 12, "DUP": 10, "INV": 11, "TRA": 11}
SET_ORDER_TYPES = ("DUP", "TRA")      # read lists built from Python sets in the reference (DUP:82,96; TRA:182)


def load_json(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".gz"):
        with gzip.open(path, "rt") as f:
            return json.load(f)
    with open(path) as f:
        return json.load(f)


def store_from_json(d):
    kw = {}
    if "reads_off" in d:
        kw = dict(reads_off=np.array(d["reads_off"], np.int64), r_start=np.array(d["r_start"], np.int64),
                  r_end=np.array(d["r_end"], np.int64), r_primary=np.array(d["r_primary"], np.uint8),
                  r_id=np.array(d["r_id"], np.int32))
    if d.get("contig_len") is not None:
        kw["contig_len"] = np.array(d["contig_len"], np.int64)
    return SigStore(chroms=d["chroms"], a=np.array(d["a"], np.int64), b=np.array(d["b"], np.int64),
                    read_id=np.array(d["read_id"], np.int32), aux=np.array(d["aux"], np.int32),
                    seg_index={(t, c): (b, e) for t, c, b, e in d["seg_index"]},
                    names=NameTable(d["names"], d["name_fmt"]), strands=tuple(d["strands"]),
                    ins_seq=None if d["ins_seq"] is None else {int(k): v for k, v in d["ins_seq"].items()}, **kw)


def canonical_row(t, row):
    row = list(row)
    k = READS_FIELD[t]
    row[k] = ",".join(sorted(row[k].split(",")))
    return row


def digest(t, rows):
    text = "\n".join("\t".join(canonical_row(t, r)) for r in rows)
    return hashlib.sha256(text.encode()).hexdigest()


def rows_by_task(store, params, engine, tasks=None):
    """engine(HostBatch) -> HostResult; returns {(type, chr): rows} for all tasks in one batch."""
    tasks = tasks or store.tasks()
    hb = store.host_batch(tasks, params)
    res = engine(hb)
    per_seg = rows_mod.rows_by_segment(store, hb.segments, res)      # native row builder (csv_rows_emit + _rowsplit)
    out = {t: per_seg[k] for k, t in enumerate(tasks)}
    # ... which must say exactly what the plain Python statement of the row layouts says
    plain = {t: [] for t in tasks}
    for k, row in rows_mod.materialise_py(store, hb.segments, res.trimmed()):
        plain[tasks[k]].append(row)
    assert plain == out, "native rows differ from rows.materialise_py"
    return out, res, hb


def assert_rows_equal(t, got, want, where=""):
    assert len(got) == len(want), "%s %s: %d rows, reference has %d" % (where, t, len(got), len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        if t in SET_ORDER_TYPES:
            g, w = canonical_row(t, g), canonical_row(t, w)
        assert list(g) == list(w), "%s %s row %d:\n got %s\nwant %s" % (where, t, i, g, w)


SOA_FIELDS = ("call_seg", "call_cluster", "call_aux", "bp1", "bp2", "support", "cipos", "cilen", "search_pos",
              "seq_pick", "dr", "dv", "gl_idx", "support_off")


def assert_soa_equal(got, want, store=None, set_order_segments=()):
    """bit-exact comparison of two HostResult.trimmed() dicts"""
    assert got["n_clusters"] == want["n_clusters"]
    for f in SOA_FIELDS:
        assert np.array_equal(got[f], want[f]), "field %s differs (first at %s)" % (
            f, np.flatnonzero(np.asarray(got[f]) != np.asarray(want[f]))[:5] if len(got[f]) == len(want[f]) else "len")
    go, wo = got["support_off"], want["support_off"]
    gs, ws = got["support_sig"], want["support_sig"]
    if len(set_order_segments) == 0:
        assert np.array_equal(gs, ws), "support_sig differs"
    else:
        seg = got["call_seg"]
        for c in range(len(seg)):
            x, y = gs[go[c]:go[c + 1]], ws[wo[c]:wo[c + 1]]
            if seg[c] in set_order_segments:
                assert sorted(store.read_id[x].tolist()) == sorted(store.read_id[y].tolist())
            else:
                assert np.array_equal(x, y), "support list of call %d differs" % c
    for f in ("cluster_id", "allele_id", "seg_status"):
        if got.get(f) is not None and want.get(f) is not None:
            assert np.array_equal(got[f], want[f]), "%s differs" % f


def write_reference_workdir(store, work_dir):
    """Lay a store out as the reference's <TYPE>.pickle / reads.pickle files + index dict
    (cuteSV main script :817-857): one pickled list per chromosome at a recorded byte offset."""
    import pickle
    from cutesv_amd.columns import TYPES
    per_type, reads = store.tuple_lists()
    index = {}
    for t in TYPES:
        index[t] = {}
        with open(os.path.join(work_dir, t + ".pickle"), "wb") as f:
            for ch in store.chroms:
                blk = [x for x in per_type[t] if x[-1] == ch]
                if blk:
                    index[t][ch] = f.tell()
                    pickle.dump(blk, f)
    index["reads"] = {}
    with open(os.path.join(work_dir, "reads.pickle"), "wb") as f:
        for ch in store.chroms:
            blk = [r for r in reads if r[-1] == ch]
            if blk:
                index["reads"][ch] = f.tell()
                pickle.dump(blk, f)
    return index


# ------------------------------------------------------------------------------------------------ rebuild golden (8f row 2)
def rebuild_case_inputs(case):
    """the candidates of a rebuild_order.json.gz case in the order process_process_sigs_type reads them (main script
    :753-761: worker files in pid order, batches in file order) -> ({type: [tuple]}, reads list)"""
    per = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
    reads = []
    for f in case["files"]:
        for t in per:
            for batch in f[t]:
                per[t].extend(tuple(x) for x in batch)
        for batch in f["reads"]:
            reads.extend(tuple(x) for x in batch)
    return per, reads


def rebuild_expected(case):
    """the reference's per-(type, chromosome) lists with int(pos) in place of the INS float position (the columns hold
    int(pos): every consumer takes int(), cuteSV_resolveINDEL.py:271)"""
    out = {}
    for t in ("DEL", "INS", "DUP", "INV", "TRA"):
        for ch, rows in case["out"][t]:
            out[(t, ch)] = [tuple([int(r[0])] + r[1:]) if t in ("DEL", "INS", "DUP") else tuple(r) for r in rows]
    return out


# ------------------------------------------------------------------------------------------------ CIGAR scan golden (8f row 4)
def cigar_case_inputs(case):
    """a cigar_sigs.json.gz case -> (cig_off, cigar, ref_start, use, names, seqs, kwargs of the scan)"""
    from cutesv_amd import extract
    p = case["params"]
    from cutesv_amd import synth
    reads = case["reads"]
    for r in reads:
        r["seq"] = synth.pseudo_sequence(r["seq_len"], r["seq_key"])
    off, flat = extract.encode_cigars([r["cigar"] for r in reads])
    start = np.array([r["start"] for r in reads], np.int64)
    # the gates of parse_read that are the caller's: query_length >= min_read_len (:607), mapq >= min_mapq (:614)
    use = np.array([1 if (len(r["seq"]) >= p["min_read_len"] and r["mapq"] >= p["min_mapq"]) else 0 for r in reads], np.uint8)
    kw = dict(min_siglength=p["min_siglength"], merge_ins_threshold=p["mi"], merge_del_threshold=p["md"])
    return off, flat, start, use, [r["name"] for r in reads], [r["seq"] for r in reads], kw


def assert_cigar_case(case, scan):
    """scan(cig_off, cigar, ref_start, use, **kw) -> signature arrays; compared with the reference's candidate lists"""
    from cutesv_amd import extract
    off, flat, start, use, names, seqs, kw = cigar_case_inputs(case)
    sig = scan(off, flat, start, use, **kw)
    ins, dele = extract.candidates(sig, names, seqs, "chr7")
    assert [list(x) for x in ins] == case["INS"], case["name"]
    assert [list(x) for x in dele] == case["DEL"], case["name"]
    return sig


# ------------------------------------------------------------------------------------------------ split-read golden (8f row 4)
SPLIT_CHROMS = ["1", "10", "2", "X"]                         # in Python string order, as the generator used them


def split_case_inputs(case):
    """a split_sigs.json.gz case -> (flat csv_split_in arrays, read names, queries, chromosome names by rank, kwargs)"""
    from cutesv_amd import extract, synth
    names = sorted(SPLIT_CHROMS)
    rank = {c: i for i, c in enumerate(names)}
    reads, p = case["reads"], case["params"]
    enc = extract.encode_split_reads([(r["primary"], r["sa"], r["qlen"]) for r in reads], rank)
    queries = [synth.pseudo_sequence(r["qlen"], r["key"]) for r in reads]
    kw = dict(sv_size=p["sv"], min_mapq=p["min_mapq"], max_split_parts=p["parts"], max_size=p["max_size"])
    return enc, [r["name"] for r in reads], queries, names, kw


def assert_split_case(case, scan):
    """scan(enc, **kw) -> candidate arrays; compared with the reference's five candidate lists (values AND the int / float
    type of the INS position, which the reference produces both ways)"""
    from cutesv_amd import extract
    enc, names, queries, chroms, kw = split_case_inputs(case)
    sig = scan(enc, **kw)
    cand = extract.split_candidates(sig, names, queries, chroms)
    for t in ("DEL", "INS", "DUP", "INV", "TRA"):
        got = [list(x) for x in cand[t]]
        assert got == case[t], (case["name"], t, len(got), len(case[t]))
        assert all(type(g[0]) is type(w[0]) for g, w in zip(got, case[t])), (case["name"], t)
    return sig


# ------------------------------------------------------------------------------------------------ whole parse_read golden (8f row 4)
class StubRecord:
    """what parse_read touches of a pysam.AlignedSegment, rebuilt from a parse_reads.json.gz record"""

    def __init__(self, d):
        from cutesv_amd import synth
        self.query_name, self.flag, self.mapq, self.reference_start = d["name"], d["flag"], d["mapq"], d["start"]
        self.cigartuples = [tuple(x) for x in d["cigar"]]
        self.query_sequence = synth.pseudo_sequence(d["seq_len"], d["seq_key"])
        self.query_length = d["seq_len"]
        self.reference_end = d["start"] + sum(ln for op, ln in self.cigartuples if op in (0, 2, 3, 7, 8))
        self._tags = [tuple(t) for t in d["tags"]]

    def get_tags(self):
        return self._tags


def assert_parse_case(case, cigar_fn, split_fn):
    from cutesv_amd import extract
    p = case["params"]
    rank = {c: i for i, c in enumerate(case["chroms"])}
    cand = extract.parse_reads([StubRecord(d) for d in case["reads"]], case["chrom"], rank, p["sv"], p["min_mapq"], p["parts"], p["min_read_len"],
                               p["min_siglength"], p["md"], p["mi"], p["max_size"], cigar_fn, split_fn)
    for t in ("DEL", "INS", "DUP", "INV", "TRA"):
        got = [list(x) for x in cand[t]]
        assert got == case[t], (case["name"], t, len(got), len(case[t]), next(((i, g, w) for i, (g, w) in enumerate(zip(got, case[t])) if g != w), None))
        assert all(type(g[0]) is type(w[0]) for g, w in zip(got, case[t])), (case["name"], t)


def assert_single_pipe_case(case, cigar_fn, split_fn):
    """extract.single_pipe == the reference's single_pipe (single_pipe.json.gz): candidates per type and the reads table rows"""
    from cutesv_amd import extract
    p = case["params"]
    rank = {c: i for i, c in enumerate(case["chroms"])}
    chrom, task_start = case["task"][0], case["task"][1]
    cand, reads_info = extract.single_pipe([StubRecord(d) for d in case["reads"]], chrom, task_start, rank, p["sv"], p["min_mapq"], p["parts"], p["min_read_len"],
                                           p["min_siglength"], p["md"], p["mi"], p["max_size"], cigar_fn, split_fn, bed_regions=case["bed"])
    for t in ("DEL", "INS", "DUP", "INV", "TRA"):
        got = [list(x) for x in cand[t]]
        assert got == case[t], (case["name"], t, len(got), len(case[t]))
    assert [list(x) for x in reads_info] == case["reads_table"], (case["name"], len(reads_info), len(case["reads_table"]))
    assert len(reads_info) > 20


# ------------------------------------------------------------------------------------------------ zero-width genotype windows
def zero_width_window_case():
    """max_cluster_bias_DEL = 0 with --genotype: every DEL window is (p, p).  The reference RAISES there (KeyError inside
    overlap_cover's sweep, cuteSV_genotype.py:95-159: the window's right end is processed before its left end) and main_ctrl
    swallows the exception, so the whole task's rows are lost (main script :1198-1199; checked against the reference itself
    in the build container).  The build does not reproduce the crash: it returns the calls, genotyped by the rule the sweep
    implements wherever it returns - cover = primary reads with start <= L and end >= R.  -> (store, params, expected rows)"""
    per = {"DEL": [(1000, 50 + (i % 3), "r%d" % i, "DEL", "1") for i in range(8)] + [(5000, 80, "q%d" % i, "DEL", "1") for i in range(6)]}
    reads = ([(900 - 10 * i, 1200 + 10 * i, 1, "r%d" % i, "1") for i in range(8)] +
             [(400, 6000, 1, "x1", "1"), (990, 1001, 1, "x2", "1"), (1000, 1000, 1, "x3", "1"), (1001, 2000, 1, "x4", "1"),
              (4000, 5000, 1, "x5", "1"), (5000, 7000, 0, "x6", "1")] + [(4900, 5100, 1, "q%d" % i, "1") for i in range(6)])
    st = SigStore.from_tuple_lists(per, reads)
    p = Params(min_support=3, genotype=True, max_cluster_bias_DEL=0)
    want = [['1', 'DEL', '1000', '-50', '8', '-0,0', '-0,0', '3', '0/1', '51,3,3', '3', '51.0', 'r0,r3,r6,r1,r4,r7,r2,r5'],
            ['1', 'DEL', '5000', '-80', '6', '-0,0', '-0,0', '2', '1/1', '41,4,2', '3', '40.6', 'q0,q1,q2,q3,q4,q5']]
    return st, p, want
