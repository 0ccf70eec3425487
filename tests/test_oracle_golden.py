"""CPU tests (-m "not gpu"): the oracle and the host logic against vectors produced by the reference.

These pin the C restatement (oracle/cutesv_oracle.c) and the host-side row/GL code to what the
reference itself returns (tests/golden/make_golden.py ran it in the build container).
"""
import numpy as np
import pytest

from cutesv_amd import synth, genotype
from cutesv_amd.columns import Params
from oracle import oracle
from helpers import (load_json, store_from_json, rows_by_task, assert_rows_equal, digest)


def _check_case(case):
    st = store_from_json(case["store"])
    p = Params(**case["params"])
    want = {(t, c): r for t, c, r in case["rows"]}
    got, res, hb = rows_by_task(st, p, oracle.cluster_batch, tasks=list(want.keys()))
    for key in want:
        assert_rows_equal(key[0], got[key], want[key], where="%s %s" % (case["name"], key))


def test_small_cases_rows_identical_to_reference():
    cases = load_json("small_cases.json.gz")
    assert len(cases) >= 14
    for case in cases:
        _check_case(case)


def test_known_answers():
    cases = load_json("known_answers.json")
    by = {c["name"]: c for c in cases}
    # the survey's hand-checked values are really what the reference returns
    tra = by["tra_double_count"]["rows"][0][2][0]
    assert tra[2] == "26" and tra[1] == "N[5:101[" and tra[5] == "5"
    ins = by["ins_tie_order"]["rows"][0][2][0]
    assert ins[2] == "1004" and ins[3] == "53" and ins[12] == "rb,rc,ra" and ins[5] == "-1,1" and ins[6] == "-5,5"
    inv = by["inv_bankers"]["rows"][0][2][0]
    assert inv[2] == "100" and inv[3] == "900" and inv[11] == "a,c,b,d"   # input is re-sorted by the rebuild key first
    for case in cases:
        _check_case(case)


def test_tra_genotyping_rows_identical_to_reference():
    # call_gt / count_coverage (cuteSV_resolveTRA.py:258-309) with the reads table as the alignment stream
    cases = load_json("tra_genotype.json.gz")
    assert len(cases) >= 6
    seen = set()
    for case in cases:
        _check_case(case)
        for _, _, rows in case["rows"]:
            seen |= {"dot" if r[7] == "./." else "gt" for r in rows}
    assert seen == {"dot", "gt"}


def test_gl_table_and_index():
    g = load_json("gl_table.json.gz")
    lib = oracle.lib()
    for c0, c1, gt, pl, gq, qual in g["table"] + g["samples"]:
        idx = genotype.gl_index(c0, c1)
        assert idx == lib.csvo_gl_index(c0, c1)
        assert genotype.gl_fields(idx) == (gt, pl, gq, qual), (c0, c1)
    assert genotype.gl_fields(genotype.gl_index(3, 10)) == ("1/1", "68,6,1", "5", "68.1")
    assert genotype.gl_fields(genotype.gl_index(17, 9)) == ("0/1", "20,0,96", "19", "19.6")


def test_numpy_std_mean_and_cipos_bit_exact():
    g = load_json("gl_table.json.gz")
    lib = oracle.lib()
    for vals, n, seed, ci, std_hex, mean_hex in g["cipos"]:
        if vals is None:
            vals = [int(x) for x in np.random.default_rng(seed).integers(10**8, 10**8 + 5000, n)]
        std = oracle.np_std(vals)
        assert std.hex() == std_hex
        assert (float(sum(vals)) / float(len(vals))).hex() == mean_hex
        v = lib.csvo_cipos(std, len(vals))
        assert "-%d,%d" % (v, v) == ci
    assert lib.csvo_cipos(12.5, 10) == 7
    # live cross-check against the numpy in this image (same pairwise summation)
    rng = np.random.default_rng(3)
    for n in (1, 7, 8, 9, 127, 128, 129, 1000, 8192, 8193, 30000):
        x = rng.standard_normal(n) * 1e3
        assert oracle.np_sum(x) == float(np.add.reduce(x))
        v = rng.integers(0, 250_000_000, n)
        assert oracle.np_std(v) == float(np.std(v.tolist()))


def test_overlap_cover_semantics():
    cases = load_json("overlap_cover.json.gz")
    for c in cases:
        reads = sorted(c["reads"], key=lambda r: r[0])
        names = sorted(set(r[3] for r in reads))
        rank = {n: i for i, n in enumerate(names)}
        L2 = [int(round(2 * s[0])) for s in c["svs"]]
        R2 = [int(round(2 * s[1])) for s in c["svs"]]
        got = oracle.cover_count([r[0] for r in reads], [r[1] for r in reads], [r[2] for r in reads],
                                 [rank[r[3]] for r in reads], L2, R2)
        assert got.tolist() == [len(x) for x in c["cover"]]


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3_s025", "cfg4_s002", "cfg5_s002"])
def test_config_digests(name, golden_dir):
    d = load_json("digests.json")[name]
    sites = dict(np.load(golden_dir + "/sim_sites.npz"))
    st = {"cfg1": lambda: synth.sim_all_types(sites, seed=20260101, chroms=["1"]),
          "cfg2": lambda: synth.sim_all_types(sites, seed=20260102),
          "cfg3_s025": lambda: synth.ont30(scale=0.25),
          "cfg4_s002": lambda: synth.hifi30_gt(scale=0.02),
          "cfg5_s002": lambda: synth.ont90_all(scale=0.02)}[name]()
    assert st.n_sig == d["n_sig"] and st.n_reads == d["n_reads"]
    p = Params(**d["params"])
    got, res, hb = rows_by_task(st, p, oracle.cluster_batch)
    assert len(got) == len(d["segments"])
    for (t, c), rows in got.items():
        n, h = d["segments"]["%s:%s" % (t, c)]
        assert len(rows) == n, (t, c)
        assert digest(t, rows) == h, (t, c)


def test_py_restatement_rows_identical_to_reference():
    """The Python stand-in that bench.py times as the reference's CPU path returns the reference's rows."""
    from oracle import py_restatement as pr
    cases = load_json("small_cases.json.gz") + load_json("known_answers.json")
    for case in cases:
        st = store_from_json(case["store"])
        p = Params(**case["params"])
        want = {(t, c): r for t, c, r in case["rows"]}
        tasks = list(want.keys())
        res = pr.run_pool(pr.tasks_from_store(st, p, tasks), processes=1)
        for (t, c), (chrom, rows) in zip(tasks, res):
            assert chrom == c
            assert_rows_equal(t, [[str(x) for x in r] for r in rows], want[(t, c)], where="py %s %s" % (case["name"], (t, c)))


def test_rebuild_order_identical_to_reference():
    """SigStore.from_tuple_lists (the host statement of the rebuild order) against the lists the reference's own
    process_process_sigs_type + remove_duplicates_sorted wrote for the same per-worker pickles (main script :750-857,
    :958-969): duplicates across workers, adversarial read-name orders, x.5 INS positions, INS rows that differ only in
    their sequence."""
    from helpers import rebuild_case_inputs, rebuild_expected
    from cutesv_amd.columns import SigStore
    for case in load_json("rebuild_order.json.gz"):
        per, reads = rebuild_case_inputs(case)
        st = SigStore.from_tuple_lists(per, reads)
        got, got_reads = st.tuple_lists()
        want = rebuild_expected(case)
        assert set(want) == set(st.seg_index), case["name"]
        for (t, ch), rows in want.items():
            mine = [x for x in got[t] if x[-1] == ch]
            assert mine == rows, (case["name"], t, ch)
        # reads: the reference keeps the concatenation order per chromosome (:810); the store holds the same rows
        # start-sorted, ties in that order
        for ch, rows in case["out"]["reads"]:
            ref_sorted = sorted((tuple(r) for r in rows), key=lambda r: r[0])
            assert [r for r in got_reads if r[-1] == ch] == ref_sorted, (case["name"], ch)


def test_cigar_scan_identical_to_reference():
    """oracle (csvo_cigar_signatures) == the candidate lists the reference's parse_read + generate_combine_sigs produced
    for the same reads (main script :606-655, :515-575): merging distances incl. the DEL quirk of :569, hard / soft clips,
    N and P operations, min_siglength 1"""
    from helpers import assert_cigar_case
    n = 0
    for case in load_json("cigar_sigs.json.gz"):
        sig = assert_cigar_case(case, oracle.cigar_signatures)
        n += len(sig["ins_pos"]) + len(sig["del_pos"])
    assert n > 3000


def test_split_read_analysis_identical_to_reference():
    """oracle (csvo_split_signatures) == the candidate lists the reference's organize_split_signal / analysis_split_read
    produced for synthetic primary + SA-tag inputs (tests/golden/make_golden_split.py): all five SV types, every rule of
    the two-segment and sliding three-segment analysis, the insertion-inside-a-translocation rule, both strands, mapq and
    segment-count gates"""
    from helpers import assert_split_case
    from oracle import oracle
    kinds = set()
    for case in load_json("split_sigs.json.gz"):
        sig = assert_split_case(case, oracle.split_signatures)
        kinds |= set(np.unique(sig["kind"]).tolist())
        if case["name"] == "ins_inside_tra":
            assert ((sig["kind"] == 1) & ((sig["aux"] & 2) == 0)).any()        # the integer-position form (:452)
    assert kinds == {0, 1, 2, 3, 4}


def test_whole_parse_read_identical_to_reference():
    """extract.parse_reads (the per-batch form of the extraction: CIGAR scan + split-read analysis + the text-side glue)
    with the oracle as the engine == what the reference's parse_read, called read after read on records with CIGARs, clips,
    flags and SA tags, appended to its five candidate lists (tests/golden/make_golden_parse.py)"""
    from helpers import assert_parse_case
    from oracle import oracle
    for case in load_json("parse_reads.json.gz"):
        assert_parse_case(case, oracle.cigar_signatures, oracle.split_signatures)


def test_single_pipe_task_gates_and_reads_table_vs_reference():
    """the extraction task loop (main script :697-743): gates + candidates + reads table rows, with the oracle's CIGAR / split engines"""
    from helpers import assert_single_pipe_case
    from oracle import oracle
    for case in load_json("single_pipe.json.gz"):
        assert_single_pipe_case(case, oracle.cigar_signatures, oracle.split_signatures)


def test_pool_rows_of_the_split_candidates_encode_the_reference_tuples():
    """the rows the split-read kernel appends to the device-resident pool (restated on the host by extract.pool_rows_of_split)
    carry exactly what the rebuild sorts of the reference's candidate tuples (split_sigs.json.gz: organize_split_signal's own
    output): positions, lengths / second positions, the INS sequence length, the INV strand code, the TRA mate and type"""
    from cutesv_amd import extract
    from cutesv_amd.columns import BND_CODE
    from helpers import split_case_inputs
    n = 0
    for case in load_json("split_sigs.json.gz"):
        enc, names, queries, chroms, kw = split_case_inputs(case)
        sig = oracle.split_signatures(enc, **kw)
        rows = extract.pool_rows_of_split(sig, [0, 100, 200, 300, 400], 5000, [len(q) for q in queries])
        cand = extract.split_candidates(sig, names, queries, chroms)
        at = {t: 0 for t in cand}
        rank = {c: i for i, c in enumerate(chroms)}
        for i, kind in enumerate(sig["kind"].tolist()):
            t = ("DEL", "INS", "DUP", "INV", "TRA")[kind]
            x = cand[t][at[t]]; at[t] += 1
            a, b, aux, seg, read = (int(rows[k][i]) for k in ("a", "b", "aux", "seg", "read"))
            assert read == 5000 + int(sig["read"][i]) and names[read - 5000] == (x[2] if t == "INS" else x[-3])
            assert seg == kind * 100 + rank[x[-1]]
            if t in ("DEL", "DUP"):
                assert (a, b, aux) == (x[0], x[1], 0)
            elif t == "INS":
                assert (a, b, aux) == (int(x[0]), x[1], len(x[3]))
            elif t == "INV":
                assert (aux, a, b) == ({"++": 0, "--": 1}[x[0]], x[1], x[2])
            else:
                assert (aux, a, b) == (rank[x[2]] * 8 + BND_CODE[x[0]], x[1], x[3])
            n += 1
    assert n > 1000


def test_zero_width_genotype_windows_return_calls():
    """where the reference raises and loses the task (helpers.zero_width_window_case), the oracle - and with it the build -
    returns the calls with the cover rule's DR: x1 (400..6000), x2 (990..1001) and x3 (1000..1000) cover (1000, 1000); x4 starts
    behind it; the support reads do not count"""
    from helpers import zero_width_window_case
    from cutesv_amd import rows as rows_mod
    from oracle import oracle
    st, p, want = zero_width_window_case()
    hb = st.host_batch(st.tasks(), p)
    got = rows_mod.rows_by_segment(st, hb.segments, oracle.cluster_batch(hb, per_sig=False))
    assert got == [want]
