"""GPU parity tests (-m gpu): the HIP path through the C ABI against the oracle and the golden fixtures."""
import os

import numpy as np
import pytest

from cutesv_amd import synth, engine, _abi
from cutesv_amd.columns import Params, SigStore
from helpers import (load_json, store_from_json, rows_by_task, assert_rows_equal, assert_soa_equal, digest)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = engine.Context(0)
    yield c
    c.close()


def _oracle():
    from oracle import oracle
    return oracle


def _hip_engine(ctx):
    return lambda hb: ctx.cluster_batch(hb, per_sig=True)


def _set_order_segments(hb):
    return set(np.flatnonzero((hb.segments["svtype"] == _abi.DUP) | (hb.segments["svtype"] == _abi.TRA)).tolist())


def _compare_soa(ctx, st, p, tasks=None):
    tasks = tasks or st.tasks()
    hb = st.host_batch(tasks, p)
    want = _oracle().cluster_batch(hb, per_sig=True).trimmed()
    got = ctx.cluster_batch(hb, per_sig=True).trimmed()
    # allele_id of DUP/TRA support lists depends on the (set-ordered) choice of representative signature
    so = _set_order_segments(hb)
    assert_soa_equal(got, want, store=st, set_order_segments=())
    return got


def test_small_cases_rows_identical_to_reference(ctx):
    for case in load_json("small_cases.json.gz"):
        st = store_from_json(case["store"])
        p = Params(**case["params"])
        want = {(t, c): r for t, c, r in case["rows"]}
        got, res, hb = rows_by_task(st, p, _hip_engine(ctx), tasks=list(want.keys()))
        for key in want:
            assert_rows_equal(key[0], got[key], want[key], where="%s %s" % (case["name"], key))


def test_known_answers(ctx):
    for case in load_json("known_answers.json"):
        st = store_from_json(case["store"])
        p = Params(**case["params"])
        want = {(t, c): r for t, c, r in case["rows"]}
        got, res, hb = rows_by_task(st, p, _hip_engine(ctx), tasks=list(want.keys()))
        for key in want:
            assert_rows_equal(key[0], got[key], want[key], where="%s %s" % (case["name"], key))


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("preset", ["default", "ont", "hifi", "keep"])
def test_soa_bit_exact_vs_oracle_small(ctx, seed, preset):
    p = {"default": Params(genotype=True), "ont": Params.ont(genotype=True),
         "hifi": Params.hifi(genotype=True, min_support=3), "keep": Params.ont(remain_reads_ratio=0.6, genotype=True)}[preset]
    st = synth.small_mixed(seed=100 + seed, dup_frac=0.2, n_loci=200)
    _compare_soa(ctx, st, p)


def test_large_clusters_all_tiers(ctx):
    # wide bias chains thousands of signatures together: exercises the workgroup tier in LDS and in global scratch
    st = synth.small_mixed(seed=77, n_sites=30, coverage=60, n_noise=30000, n_loci=3000, contig_len=400_000, dup_frac=0.3)
    p = Params(max_cluster_bias_DEL=3000, max_cluster_bias_INS=3000, max_cluster_bias_DUP=20000, max_cluster_bias_INV=20000,
               max_cluster_bias_TRA=5000, min_support=3, genotype=False)
    got = _compare_soa(ctx, st, p)
    sizes = np.bincount(got["cluster_id"][got["cluster_id"] >= 0])
    assert sizes.max() > 2048, sizes.max()


@pytest.mark.parametrize("n", [65733, 70789])
def test_allele_larger_than_the_static_pow_table(ctx, n):
    # cal_CIPOS divides by `num ** 0.5` = libm pow(), which is not sqrt() for these n (and 269 more below 300 000); the
    # device's pow table is sized to the batch's longest segment, so an allele of any size divides by the reference's value
    import math
    from cutesv_amd.columns import SigStore
    assert math.pow(n, 0.5) != math.sqrt(n)
    rng = np.random.default_rng(n)
    pos = np.sort(rng.integers(100_000, 100_000 + 40 * n, n))            # gaps ~40 < bias: one chained cluster
    ln = rng.integers(980, 1020, n)                                       # one allele at ratio 0.3
    per = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
    per["DEL"] = [(int(pos[i]), int(ln[i]), "p%06d" % i, "DEL", "1") for i in range(n)]
    st = SigStore.from_tuple_lists(per)
    got = _compare_soa(ctx, st, Params(min_support=5, max_size=-1, max_cluster_bias_DEL=5000, diff_ratio_merging_DEL=0.3))
    assert got["support"].max() == n and got["cipos"].max() > 0


def test_cipos_on_integer_boundaries(ctx):
    # cal_CIPOS = int(1.96 * std / n ** 0.5).  The register tier computes the variance exactly in integers and replays
    # numpy's float64 summation only when the value is too close to an integer to call; here every allele sits ON such a
    # boundary in exact arithmetic (half the reads at p, half at p + 2 k: std = k; n = 4, 16, 64 make 1.96 k / sqrt(n) an
    # integer for k = 50 j, 100 j, 200 j), where float64 lands a hair above or below: the replay must decide, as numpy does
    from cutesv_amd.columns import SigStore
    per = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
    pos, rid = 100_000, 0
    # (n = 100 .. 256: the one-wavefront-per-cluster tier, whose exact-variance form has the same fallback)
    for n, step in ((4, 50), (16, 100), (64, 200), (100, 250), (144, 300), (196, 50), (256, 400)):
        for j in range(1, 13):
            k = step * j
            for i in range(n):
                p_ = pos + (2 * k if i >= n // 2 else 0)
                ln = 500 + (2 * k if i % 2 else 0)                    # the lengths sit on a boundary too (two alleles would need ratio < ...: one allele at 0.9)
                per["DEL"].append((p_, ln, "b%06d" % rid, "DEL", "1"))
                rid += 1
            pos += 100_000
    st = SigStore.from_tuple_lists(per)
    got = _compare_soa(ctx, st, Params(min_support=3, max_size=-1, max_cluster_bias_DEL=20000, diff_ratio_merging_DEL=50.0))
    assert len(got["bp1"]) == 84 and (got["cipos"] > 0).all()


def test_register_tiers_fallback_routes(ctx):
    # lengths >= 2^26 leave the one-word rank key, repeated read names leave the hashed duplicate filter: the
    # 64-bit / exact routes of indel_unit<16>, <32> and <64>, next to ordinary clusters in the same wavefronts
    from cutesv_amd.columns import SigStore
    rng = np.random.default_rng(123)
    per = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
    pos = 10_000
    for site in range(240):
        m = [12, 25, 50, 9, 16, 17, 32, 33][site % 8]
        huge = site % 3 == 0
        base = int(rng.integers(2**27, 2**30)) if huge else int(rng.integers(50, 600))
        names = ["r%05d" % i for i in rng.choice(20000, m, replace=False)]
        if site % 5 == 0:
            names[3] = names[1]; names[m - 1] = names[0]                  # a read with several signatures
        for nm in names:
            ln = max(30, int(base * (1 + rng.normal(0, 0.02))))
            p_ = pos + int(rng.normal(0, 10))
            per["DEL"].append((p_, ln, nm, "DEL", "1"))
            per["INS"].append((p_ + 2, ln, nm, "A" * min(ln, 700), "INS", "1"))
        pos += 3000
    st = SigStore.from_tuple_lists(per)
    got = _compare_soa(ctx, st, Params.ont(min_support=5, max_size=-1))
    assert len(got["bp1"]) > 350 and (got["bp2"] >= 2**26).any()


@pytest.mark.parametrize("n", [2048, 4096, 4097, 6143, 64, 1])
def test_chain_tile_edges(ctx, n):
    # batches whose size sits on / next to the 2048-signature tile boundary of the chain kernels: the sentinel that
    # ends the last cluster lands in a full last wavefront region, clusters straddle tiles, the last one is long
    from cutesv_amd.columns import SigStore
    rng = np.random.default_rng(n)
    per = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
    pos, i = 1000, 0
    while i < n:
        m = int(rng.choice([1, 2, 5, 12, 20, 40, 70, 300])) if n - i > 400 else n - i      # the tail is one cluster
        m = min(m, n - i)
        for j in range(m):
            per["DEL"].append((pos + int(rng.integers(0, 30)), int(rng.integers(40, 60)), "q%06d" % (i + j), "DEL", "1"))
        pos += 5000
        i += m
    st = SigStore.from_tuple_lists(per)
    assert st.n_sig == n
    got = _compare_soa(ctx, st, Params.ont(min_support=3))
    assert got["n_clusters"] >= 1


def test_deep_pileup_next_to_ordinary_loci(ctx):
    """No depth limit (the reference has none, cuteSV_genotype.py:95-159): a 12 000x pile-up - clusters of ~10 k
    signatures, cover sets of ~12 k reads, far beyond the 32 KB LDS tables - in one batch with ordinary 30x loci.
    The deep calls take the global-memory hash tables; every call of the batch equals the oracle's."""
    base = synth.small_mixed(seed=5, coverage=30, n_contigs=2, contig_len=1_000_000)
    pile = synth.small_mixed(seed=6, coverage=12000, n_sites=3, n_contigs=2, contig_len=40_000, n_noise=0, n_loci=0, dup_frac=0.02)
    st = synth.concat_stores(base, pile)
    got = _compare_soa(ctx, st, Params(genotype=True, min_support=10, max_cluster_bias_DEL=200))
    tot = got["dr"].astype(np.int64) + got["dv"]
    assert tot.max() > 10000, tot.max()
    # (the pile-up generator leaves a few negative pos2 values in its DUP / INV segments: those clusters are silenced and
    # their segments flagged, exactly as the oracle does - compared inside _compare_soa)
    assert (got["seg_status"][:8] == 0).all()


def test_deep_pileup_is_repeatable(ctx):
    """the workgroup tier (clusters of thousands of signatures in global scratch) gives the same answer every time:
    300 runs of the pile-up batch, call count and supports against the oracle (a missing barrier between the DUP slot
    assignment and its readers once showed up as one extra call in ~1 % of runs)"""
    pile = synth.small_mixed(seed=6, coverage=12000, n_sites=3, n_contigs=2, contig_len=40_000, n_noise=0, n_loci=0, dup_frac=0.02)
    hb = pile.host_batch(pile.tasks(), Params(genotype=False, min_support=10, max_cluster_bias_DEL=200))
    want = _oracle().cluster_batch(hb).trimmed()
    for it in range(300):
        got = ctx.cluster_batch(hb).trimmed()
        assert len(got["bp1"]) == len(want["bp1"]), it
        assert np.array_equal(got["support_off"], want["support_off"]) and np.array_equal(got["bp1"], want["bp1"]), it


def test_key_range_is_a_per_segment_status(ctx):
    """a length outside [0, 2^42) silences its own cluster and flags its segment; every other call is untouched"""
    st = synth.small_mixed(seed=31, genotype=False)
    p = Params.ont()
    ref = ctx.cluster_batch(st.host_batch(st.tasks(), p)).trimmed()
    tasks = st.tasks()
    k_bad = tasks.index(("DEL", "2"))
    victim = int(ref["support_sig"][ref["support_off"][np.flatnonzero(ref["call_seg"] == k_bad)[0]]])
    st.b[victim] = 1 << 45
    got = _compare_soa(ctx, st, p)
    assert got["seg_status"][k_bad] == _abi.SEG_KEY_RANGE and got["seg_status"].sum() == _abi.SEG_KEY_RANGE
    assert len(got["bp1"]) == len(ref["bp1"]) - (1 if True else 0) or len(got["bp1"]) < len(ref["bp1"])


def test_reads_table_in_extraction_order(ctx, monkeypatch):
    """the reads block as cuteSV's extraction leaves it (a permutation of disjoint sorted runs, main script :697-735,
    :810): ordered on the device inside the run; an arbitrary shuffle takes the general sort; both give the rows of the
    start-sorted table; a false CSV_IN_READS_SORTED promise is refused"""
    st = synth.small_mixed(seed=91, n_sites=60, coverage=40, contig_len=2_000_000)
    p = Params.ont(genotype=True, genotype_tra=True, min_support=3)
    want = ctx.cluster_batch(st.host_batch(st.tasks(), p), per_sig=True).trimmed()
    runs, _ = synth.extraction_order(st, region=150_000, workers=5)
    assert (np.diff(runs.r_start) < 0).sum() > 20
    monkeypatch.setenv("CSV_READS_GAP", "20000")           # task regions are 150 kbp here (10 Mbp / gap 1 Mbp in production)
    got = _compare_soa(ctx, runs, p)
    assert_soa_equal(got, want)
    assert ctx.last_reads_mode() == 1                       # ordered by moving whole runs, not by the general sort
    rng = np.random.default_rng(3)
    import dataclasses
    perm = np.arange(st.n_reads)
    for c in range(len(st.chroms)):
        lo, hi = int(st.reads_off[c]), int(st.reads_off[c + 1])
        perm[lo:hi] = lo + rng.permutation(hi - lo)
    shuf = dataclasses.replace(st, r_start=st.r_start[perm], r_end=st.r_end[perm], r_primary=st.r_primary[perm], r_id=st.r_id[perm])
    # DEL / INS / DUP / INV genotypes are set-valued (order free); TRA's count_coverage walks in stable start order,
    # which a shuffle of equal starts may change: compare it through the oracle on the same shuffled table
    _compare_soa(ctx, shuf, p)
    assert ctx.last_reads_mode() == 2                       # the general stable radix sort
    hb = shuf.host_batch(shuf.tasks(), p)
    hb.c.flags |= _abi.IN_READS_SORTED
    with pytest.raises(engine.CsvError) as e:
        ctx.cluster_batch(hb)
    assert e.value.code == _abi.E_UNSORTED
    hb2 = st.host_batch(st.tasks(), p)
    hb2.c.flags |= _abi.IN_READS_SORTED                    # a true promise skips the ordering stage
    assert_soa_equal(ctx.cluster_batch(hb2, per_sig=True).trimmed(), want)


def test_two_contexts_shard_one_genome(ctx):
    """chromosomes sharded over two contexts (one per GPU in production; both on device 0 here) through the HIP path:
    merged rows == the unsharded stage's (main script :1191-1197 is the merge)"""
    from cutesv_amd import resolve, shard
    st = synth.small_mixed(seed=12, n_contigs=5, contig_len=800_000)
    p = Params.ont(genotype=True)
    other = engine.Context(0)
    try:
        parts = [resolve.cluster_stage(st, p, tasks=shard.tasks_of_rank(st, r, 2, genotype=True), ctx=c) for r, c in ((0, ctx), (1, other))]
    finally:
        other.close()
    merged = shard.merge_results(parts)
    full = resolve.cluster_stage(st, p, ctx=ctx)
    assert set(merged) == set(full) and all(merged[c] == full[c] for c in full)


def test_pinned_columns_and_native_rows(ctx):
    """columns in page-locked memory give the same calls; csv_rows_emit's blob splits into the same rows as the
    CPython builder"""
    from cutesv_amd import rows as rows_mod
    st = synth.small_mixed(seed=8)
    p = Params.ont(genotype=True)
    pst = st.pinned()
    hb, phb = st.host_batch(st.tasks(), p), pst.host_batch(pst.tasks(), p)
    a, b = ctx.cluster_batch(hb, per_sig=True), ctx.cluster_batch(phb, per_sig=True)
    assert_soa_equal(b.trimmed(), a.trimmed())
    rows, _ = rows_mod.materialise(st, hb.segments, a)
    blob, n = rows_mod.rows_blob(st, hb.segments, a)
    assert rows_mod._native().split(blob, n) == rows and n == len(rows) > 0


@pytest.mark.parametrize("seed,genotype,order", [(3, False, "sorted"), (4, True, "sorted"), (5, True, "extraction"), (6, True, "shuffled")])
def test_published_results_equal_copied_results(ctx, seed, genotype, order):
    """result arrays in page-locked memory are filled in place by k_publish (one synchronisation, no staging copy, no
    unpack loop); pageable ones through the copy path.  Same bytes either way, including the capacity retry, the per-
    signature outputs and a reads table that sends the batch through the general sort."""
    st = synth.small_mixed(seed=seed, genotype=genotype)
    if genotype and order == "extraction":
        st, _ = synth.extraction_order(st, seed=seed, region=20_000, workers=4)
    elif genotype and order == "shuffled":
        rng = np.random.default_rng(seed)
        for c in range(len(st.reads_off) - 1):
            lo, hi = int(st.reads_off[c]), int(st.reads_off[c + 1])
            perm = lo + rng.permutation(hi - lo)
            for name in ("r_start", "r_end", "r_primary", "r_id"):
                col = getattr(st, name)
                col[lo:hi] = col[perm].copy()
    hb = st.host_batch(st.tasks(), Params.ont(genotype=genotype))
    for per_sig in (False, True):
        copied = ctx.cluster_batch(hb, per_sig=per_sig).trimmed()
        ctx._res_cache = None
        published = ctx.cluster_batch(hb, per_sig=per_sig, cap_calls=8, cap_support=8, reuse=True)      # (forces the capacity retry)
        assert published.cap_calls >= len(copied["bp1"]) > 8
        assert_soa_equal(published.trimmed(), copied)
        again = ctx.cluster_batch(hb, per_sig=per_sig, reuse=True).trimmed()                            # the recycled arrays
        assert_soa_equal(again, copied)
    want = _oracle().cluster_batch(hb, per_sig=True).trimmed()
    assert_soa_equal(again, want, st)


def test_genotype_cover_overflow_pass(ctx):
    # ~1000x coverage: support + cover of a call exceeds the 4 KB hash set, so the second (32 KB) pass runs
    st = synth.small_mixed(seed=78, n_sites=6, coverage=1000, n_noise=100, n_loci=20, contig_len=200_000, n_contigs=2)
    got = _compare_soa(ctx, st, Params(genotype=True, min_support=10))
    tot = got["dr"].astype(np.int64) + got["dv"]
    assert tot.max() > 1500 and tot.max() < 6000, tot.max()


@pytest.mark.parametrize("name", ["cfg1", "cfg2", "cfg3_s025", "cfg4_s002", "cfg5_s002"])
def test_config_digests(ctx, name, golden_dir):
    d = load_json("digests.json")[name]
    sites = dict(np.load(golden_dir + "/sim_sites.npz"))
    st = {"cfg1": lambda: synth.sim_all_types(sites, seed=20260101, chroms=["1"]),
          "cfg2": lambda: synth.sim_all_types(sites, seed=20260102),
          "cfg3_s025": lambda: synth.ont30(scale=0.25),
          "cfg4_s002": lambda: synth.hifi30_gt(scale=0.02),
          "cfg5_s002": lambda: synth.ont90_all(scale=0.02)}[name]()
    p = Params(**d["params"])
    got, res, hb = rows_by_task(st, p, _hip_engine(ctx))
    for (t, c), rows in got.items():
        n, h = d["segments"]["%s:%s" % (t, c)]
        assert len(rows) == n, (t, c)
        assert digest(t, rows) == h, (t, c)


def test_full_size_cfg3_vs_oracle(ctx):
    st = synth.ont30()                      # ~2.8 M signatures: BASELINE config 3 at full size
    assert st.n_sig > 2_500_000
    _compare_soa(ctx, st, Params.ont())


@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
@pytest.mark.parametrize("order", ["extraction", "sorted"])
def test_full_size_cfg4_cfg5_vs_oracle(ctx, cfg, order):
    """BASELINE configs 4 and 5 at FULL size, bit for bit against the C oracle (which needs < 1 s for either): every SoA
    field, the support lists and the per-signature outputs; the reads table once as the extraction step leaves it (the
    device reorders it) and once start-sorted."""
    if cfg == "cfg5":
        st, p, min_sig, min_calls = synth.ont90_all(), Params.ont(genotype=True), 10_000_000, 50_000
    else:
        st, p, min_sig, min_calls = synth.hifi30_gt(), Params.hifi(genotype=True, min_support=3), 400_000, 20_000
    if order == "extraction":
        st, _ = synth.extraction_order(st)
    assert st.n_sig > min_sig
    got = _compare_soa(ctx, st, p)
    assert len(got["bp1"]) > min_calls
    assert (got["gl_idx"] >= 0).sum() > min_calls // 2


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cfg5"])
def test_full_size_int32_columns_vs_oracle(ctx, cfg):
    """The path bench.py and SigStore.pinned() actually use - int32 position / length columns (CSV_IN_SIG_I32), int32 reads
    (CSV_IN_READS_I32), page-locked, results published straight into page-locked arrays - at FULL size against the C oracle on
    the int64 columns: every SoA field, the support lists, the per-signature outputs.  (r03 compared this path with the oracle
    only on a small mixed workload; the register tier now keeps int32 coordinates in 32-bit registers, a different
    instantiation from the int64 one.)"""
    from helpers import assert_soa_equal
    st, p = {"cfg3": (lambda: (synth.ont30(), Params.ont())),
             "cfg4": (lambda: (synth.extraction_order(synth.hifi30_gt())[0], Params.hifi(genotype=True, min_support=3))),
             "cfg5": (lambda: (synth.extraction_order(synth.ont90_all())[0], Params.ont(genotype=True)))}[cfg]()
    pst = st.pinned()
    hb = pst.host_batch(pst.tasks(), p)
    assert hb.a.dtype == np.int32 and hb.c.flags & _abi.IN_SIG_I32
    if hb.r_start is not None:
        assert hb.r_start.dtype == np.int32 and hb.c.flags & _abi.IN_READS_I32
    want = _oracle().cluster_batch(st.host_batch(st.tasks(), p), per_sig=True).trimmed()
    got = ctx.cluster_batch(hb, per_sig=True, reuse=True).trimmed()
    assert_soa_equal(got, want, st)
    # ... and resident: upload once, run twice (tiers on demand), deliver into the same page-locked arrays
    ctx.upload(hb, per_sig=True)
    ctx.run(); ctx.run()
    res = ctx.result_buffers(per_sig=True)
    assert_soa_equal(ctx.download(per_sig=True, into=res).trimmed(), want, st)


@pytest.mark.parametrize("cfg", ["cfg4", "cfg5"])
def test_full_size_partition_and_rerun_properties(ctx, cfg):
    """BASELINE configs 4 (HiFi 30x, 6.2 M reads, genotyping) and 5 (ONT 90x, 11 M signatures, all five types,
    genotyping) at FULL size, where the reference would need hours: size-independent properties instead of
    row-by-row comparison.
      * determinism / idempotence: a second run over the resident batch returns the same bytes;
      * partition invariance: the stage is independent per (type, chromosome) segment (main script :1116-1189), so
        clustering two disjoint halves of the chromosomes separately must give, segment by segment, exactly the calls
        of the whole-genome batch (ids and offsets shifted, values identical)."""
    import hashlib
    if cfg == "cfg5":
        st, p, min_sig, min_calls = synth.ont90_all(), Params.ont(genotype=True), 10_000_000, 50_000
    else:
        st, p, min_sig, min_calls = synth.hifi30_gt(), Params.hifi(genotype=True, min_support=3), 400_000, 20_000
    assert st.n_sig > min_sig
    tasks = st.tasks()

    def per_segment(tk):
        hb = st.host_batch(tk, p)
        ctx.upload(hb, per_sig=True)
        ctx.run(); r1 = ctx.download().trimmed()
        ctx.run(); r2 = ctx.download().trimmed()
        for k in ("bp1", "bp2", "support", "cipos", "cilen", "search_pos", "seq_pick", "dr", "dv", "gl_idx", "support_off", "support_sig", "call_seg"):
            assert np.array_equal(r1[k], r2[k]), k
        out = {}
        seg = r1["call_seg"]
        bounds = np.searchsorted(seg, np.arange(len(tk) + 1))
        for i, t in enumerate(tk):
            lo, hi = int(bounds[i]), int(bounds[i + 1])
            h = hashlib.sha256()
            for k in ("bp1", "bp2", "support", "cipos", "cilen", "search_pos", "seq_pick", "dr", "dv", "gl_idx", "call_aux"):
                h.update(np.ascontiguousarray(r1[k][lo:hi]).tobytes())
            so = r1["support_off"]
            h.update(np.ascontiguousarray(so[lo:hi + 1] - so[lo]).tobytes())
            h.update(np.ascontiguousarray(r1["support_sig"][int(so[lo]):int(so[hi])]).tobytes())
            out[t] = (hi - lo, h.hexdigest())
        return out

    whole = per_segment(tasks)
    # consistency of the per-signature outputs with the call lists (whole-genome batch still resident)
    full = ctx.download(per_sig=True).trimmed()
    cid = full["cluster_id"]
    live = cid >= 0
    assert (np.diff(cid[live]) >= 0).all() and (np.diff(cid[live]) <= 1).all()         # dense, in file order
    segs = st.host_batch(tasks, p).segments
    indel_call = np.isin(full["call_seg"], np.flatnonzero(segs["svtype"] <= _abi.INS))
    so, ss = full["support_off"], full["support_sig"]
    call_of_support = np.repeat(np.arange(len(full["bp1"])), np.diff(so))
    pick = indel_call[call_of_support]
    assert np.array_equal(full["allele_id"][ss[pick]], call_of_support[pick])            # a supporting signature points back at its call
    assert np.array_equal(np.diff(so)[indel_call], full["support"][indel_call])          # DEL/INS: support = length of the read list
    assert np.array_equal(cid[ss], full["call_cluster"][call_of_support])               # ... and lies in the call's cluster
    half_a = [t for t in tasks if st.chroms.index(t[1]) % 2 == 0]
    half_b = [t for t in tasks if st.chroms.index(t[1]) % 2 == 1]
    parts = per_segment(half_a)
    parts.update(per_segment(half_b))
    assert set(parts) == set(whole)
    assert sum(n for n, _ in whole.values()) > min_calls
    for t in tasks:
        assert parts[t] == whole[t], t


def test_zero_width_genotype_windows(ctx):
    """max_cluster_bias 0 with genotyping: the reference raises inside overlap_cover and its main_ctrl drops the task; the
    build returns the calls (helpers.zero_width_window_case says why and what)"""
    from helpers import zero_width_window_case
    st, p, want = zero_width_window_case()
    got, res, hb = rows_by_task(st, p, _hip_engine(ctx))
    assert got[("DEL", "1")] == want
    _compare_soa(ctx, st, p)


def test_empty_and_ragged(ctx):
    st = synth.small_mixed(seed=5, genotype=False)
    p = Params.ont()
    # a batch with no segments, and a batch made of a strict subset of non-adjacent segments
    hb = st.host_batch([], p)
    res = ctx.cluster_batch(hb)
    assert res.n_calls == 0 and res.n_support == 0
    tasks = [t for i, t in enumerate(st.tasks()) if i % 2 == 0]
    _compare_soa(ctx, st, p, tasks=tasks)


def test_tra_genotype_needs_the_reference_lengths(ctx):
    st = synth.small_mixed(seed=5)
    hb = st.host_batch([t for t in st.tasks() if t[0] == "TRA"][:1], Params())
    hb.segments["genotype"] = 1                      # no reads table / contig_len in this batch
    with pytest.raises(engine.CsvError) as e:
        ctx.cluster_batch(hb)
    assert e.value.code == _abi.E_INVALID


def test_tra_genotyping_rows_identical_to_reference(ctx):
    # call_gt / count_coverage (cuteSV_resolveTRA.py:258-309) over the reads table: every exit of the loop
    for case in load_json("tra_genotype.json.gz"):
        st = store_from_json(case["store"])
        p = Params(**case["params"])
        want = {(t, c): r for t, c, r in case["rows"]}
        got, res, hb = rows_by_task(st, p, _hip_engine(ctx), tasks=list(want.keys()))
        for key in want:
            assert_rows_equal(key[0], got[key], want[key], where="%s %s" % (case["name"], key))


@pytest.mark.parametrize("gt_round", [500, 30, 6])
def test_tra_genotyping_in_a_mixed_batch_vs_oracle(ctx, gt_round):
    st = synth.small_mixed(seed=400 + gt_round, n_sites=60, coverage=25, contig_len=400_000, n_loci=100)
    got = _compare_soa(ctx, st, Params.ont(genotype=True, genotype_tra=True, gt_round=gt_round, min_support=3))
    tra = np.flatnonzero(np.isin(got["call_seg"], [k for k, t in enumerate(st.tasks()) if t[0] == "TRA"]))
    assert len(tra) > 5 and (got["dv"][tra] > 0).all()
    if gt_round == 6:
        assert (got["gl_idx"][tra] < 0).any()


def test_drop_in_shims_on_a_reference_workdir(ctx, tmp_path, monkeypatch):
    """run_del/run_ins/run_inv/run_dup/run_tra with the reference's own argument tuples, reading the
    reference's own pickle layout, return the reference's rows."""
    from cutesv_amd import resolve
    from helpers import write_reference_workdir
    monkeypatch.setattr(resolve, "_ctx", ctx)
    for case in [c for c in load_json("small_cases.json.gz") if c["name"] in ("realnames_gt", "ont_gt", "hifi")]:
        st = store_from_json(case["store"])
        p = Params(**case["params"])
        d = str(tmp_path / case["name"]) + "/"
        os.makedirs(d)
        idx = write_reference_workdir(st, d)
        resolve._stores.clear()
        for t, c, want in case["rows"]:
            if t == "DEL":
                got = resolve.run_del((d, c, "DEL", p.min_support, p.diff_ratio_merging_DEL, p.max_cluster_bias_DEL,
                                       min(p.min_support, 5), "bam", p.genotype, p.gt_round, p.remain_reads_ratio, idx))
            elif t == "INS":
                got = resolve.run_ins((d, c, "INS", p.min_support, p.diff_ratio_merging_INS, p.max_cluster_bias_INS,
                                       min(p.min_support, 5), "bam", p.genotype, p.gt_round, p.remain_reads_ratio, idx))
            elif t == "INV":
                got = resolve.run_inv((d, c, "INV", p.min_support, p.max_cluster_bias_INV, p.min_size, "bam", p.genotype,
                                       p.max_size, p.gt_round, idx))
            elif t == "DUP":
                got = resolve.run_dup((d, c, p.min_support, p.max_cluster_bias_DUP, p.min_size, "bam", p.genotype,
                                       p.max_size, p.gt_round, idx))
            else:
                got = resolve.run_tra((d, c, p.min_support, p.diff_ratio_filtering_TRA, p.max_cluster_bias_TRA, "bam",
                                       False, p.gt_round, idx))
            assert got[0] == c
            assert_rows_equal(t, got[1], want, where="shim %s %s:%s" % (case["name"], t, c))
        # and the batched form returns the same rows per chromosome, in main_ctrl's concatenation order
        merged = resolve.cluster_stage(st, p, ctx=ctx)
        want_by_chr = {}
        for t in ("DEL", "INS", "INV", "DUP", "TRA"):
            for tt, c, rows in case["rows"]:
                if tt == t:
                    want_by_chr.setdefault(c, []).extend([(t, r) for r in rows])
        for c, rows in merged.items():
            assert len(rows) == len(want_by_chr.get(c, []))
            for g, (t, w) in zip(rows, want_by_chr[c]):
                assert_rows_equal(t, [g], [w], where="stage %s" % c)


def test_random_parameter_stress_vs_oracle(ctx):
    """random flags x random workload shapes, every SoA field bit-exact against the oracle"""
    rng = np.random.default_rng(2026)
    for it in range(24):
        p = Params(min_support=int(rng.integers(1, 12)), min_size=int(rng.choice([0, 30, 500])),
                   max_size=int(rng.choice([-1, 2000, 100000])), genotype=bool(rng.integers(0, 2)),
                   max_cluster_bias_INS=int(rng.choice([0, 20, 100, 1000, 5000])), diff_ratio_merging_INS=float(rng.choice([0.0, 0.1, 0.3, 0.9, 2.0])),
                   max_cluster_bias_DEL=int(rng.choice([0, 20, 200, 1000, 5000])), diff_ratio_merging_DEL=float(rng.choice([0.0, 0.2, 0.5, 1.5])),
                   max_cluster_bias_INV=int(rng.choice([10, 500, 5000])), max_cluster_bias_DUP=int(rng.choice([10, 500, 5000])),
                   max_cluster_bias_TRA=int(rng.choice([5, 50, 2000])), diff_ratio_filtering_TRA=float(rng.choice([0.2, 0.6, 1.0])),
                   remain_reads_ratio=float(rng.choice([0.3, 0.7, 1.0, 1.5])))
        st = synth.small_mixed(seed=9000 + it, n_sites=int(rng.integers(5, 60)), coverage=int(rng.choice([6, 20, 45, 90])),
                               dup_frac=float(rng.choice([0.0, 0.1, 0.6])), n_noise=int(rng.integers(0, 3000)),
                               n_loci=int(rng.integers(0, 300)), contig_len=int(rng.choice([300_000, 2_000_000])),
                               pos_sigma=float(rng.choice([1.0, 12.0, 60.0])), len_sigma=float(rng.choice([0.003, 0.04, 0.2])))
        _compare_soa(ctx, st, p)


def test_scaled_cfg4_and_cfg5_vs_oracle(ctx):
    _compare_soa(ctx, synth.hifi30_gt(scale=0.05), Params.hifi(genotype=True, min_support=3))
    _compare_soa(ctx, synth.ont90_all(scale=0.05), Params.ont(genotype=True))        # 90x: mostly mid-tier clusters


def test_end_to_end_vcf_text_from_the_gpu_path(ctx):
    """signature columns -> HIP kernels -> native VCF emit == the reference's generate_output text"""
    from cutesv_amd import vcf
    from test_vcf_emit import _canon
    small = {c["name"]: c for c in load_json("small_cases.json.gz")}
    for g in load_json("vcf_lines.json.gz"):
        case = small[g["case"]]
        st = store_from_json(case["store"])
        p = Params(**case["params"])
        ref = {c: synth.reference_sequence(g["ref_len"], seed=g["ref_seed0"] + i) for i, c in enumerate(st.chroms)}
        hb = st.host_batch([(t, c) for t, c, _ in case["rows"]], p)
        res = ctx.cluster_batch(hb)
        text, _ = vcf.emit_records(st, hb.segments, res, ref, min_size=p.min_size, max_size=p.max_size,
                                   genotype=p.genotype, **g["flags"])
        assert _canon(text) == _canon(g["text"]), (g["case"], g["flags"])


def test_input_order_contract_is_checked(ctx):
    st = synth.small_mixed(seed=21, genotype=False)
    hb = st.host_batch(st.tasks(), Params.ont())
    ctx.upload(hb)
    ctx.validate()                                        # generator output honours the rebuild order
    for col, delta in (("a", -10_000_000), ("read_id", 0)):
        bad = synth.small_mixed(seed=21, genotype=False)
        beg, end = bad.seg_index[("DEL", "2")]
        if col == "a":
            bad.a[beg + 5] += delta                       # out of position order
        else:
            bad.a[beg + 6], bad.b[beg + 6], bad.read_id[beg + 6] = bad.a[beg + 5], bad.b[beg + 5], bad.read_id[beg + 5]   # adjacent duplicate
        ctx.upload(bad.host_batch(bad.tasks(), Params.ont()))
        with pytest.raises(engine.CsvError) as e:
            ctx.validate()
        assert e.value.code == _abi.E_UNSORTED
        # ... and after any number of runs: the runs alternate between two counter arenas, the check has its own block
        # (advisor, r05: after an odd number of runs the error bit was written to one arena and read from the other)
        for runs in (1, 2, 3):
            ctx.upload(bad.host_batch(bad.tasks(), Params.ont()))
            for _ in range(runs):
                ctx.run()
            with pytest.raises(engine.CsvError) as e:
                ctx.validate()
            assert e.value.code == _abi.E_UNSORTED
    good = st.host_batch(st.tasks(), Params.ont())
    ctx.upload(good)
    ctx.run()
    ctx.validate()
    # a pipelined delivery in flight: the check refuses instead of touching counters a publish may be reading
    res = ctx.result_buffers()
    ctx.run()
    ctx.download(into=res)                                # (the pipelined delivery needs one synchronous download of the upload first)
    ctx.run()
    ctx.publish_async(res)
    with pytest.raises(engine.CsvError) as e:
        ctx.validate()
    assert e.value.code == _abi.E_STATE
    ctx.publish_wait()
    ctx.validate()


def test_gpu_rebuild_reproduces_the_reference_order_contract(ctx):
    """shuffled rows (+ injected exact duplicates) -> csv_rebuild_signatures == the store the tuple-sorting
    rebuild (main script :764-802, :958-969 restated in SigStore.from_tuple_lists / numpy lexsort) gives"""
    from cutesv_amd import rebuild
    rng = np.random.default_rng(5)
    for seed, kw in ((31, dict()), (32, dict(n_sites=120, coverage=60, n_noise=20000, n_loci=2000, dup_frac=0.3))):
        st = synth.small_mixed(seed=seed, genotype=False, **kw)
        per = {}
        for (t, ch), (b, e) in st.seg_index.items():
            d = per.setdefault(t, dict(chrom=[], a=[], b=[], read_id=[], aux=[]))
            d["chrom"].append(np.full(e - b, st.chroms.index(ch))); d["a"].append(st.a[b:e]); d["b"].append(st.b[b:e])
            d["read_id"].append(st.read_id[b:e])
            d["aux"].append(st.aux[b:e] if t in ("INS", "INV", "TRA") else np.zeros(e - b, np.int32))
        for t, d in per.items():
            cols = {k: np.concatenate(v) for k, v in d.items()}
            n = len(cols["a"])
            dup = rng.integers(0, n, max(1, n // 20))                  # exact duplicates, as overlapping extraction windows make
            perm = rng.permutation(n + len(dup))
            per[t] = {k: np.concatenate([v, v[dup]])[perm] for k, v in cols.items()}
        got, info = rebuild.store_from_unsorted(ctx, st.chroms, per)
        assert info["n_passes"] >= 5
        assert got.seg_index == st.seg_index
        assert np.array_equal(got.a, st.a) and np.array_equal(got.b, st.b) and np.array_equal(got.read_id, st.read_id)
        for (t, ch), (b, e) in st.seg_index.items():
            if t in ("INS", "INV", "TRA"):
                assert np.array_equal(got.aux[b:e], st.aux[b:e]), (t, ch)
        # and the clustering result on the rebuilt store is the same as on the original
        p = Params.ont()
        x = ctx.cluster_batch(got.host_batch(got.tasks(), p)).trimmed()
        y = ctx.cluster_batch(st.host_batch(st.tasks(), p)).trimmed()
        for k in ("bp1", "bp2", "support", "cipos", "cilen", "support_sig"):
            assert np.array_equal(x[k], y[k]), k


def test_rebuild_hands_its_columns_to_the_cluster_stage_on_the_device(ctx):
    """main script :750-857 -> :1113-1199 without a host round trip: the rebuild's sorted columns stay in device memory
    (CSV_RB_KEEP_ON_DEVICE) and csv_cluster_batch reads them there (CSV_IN_DEVICE_COLUMNS); only src_row and the rows per
    segment come back.  Same calls as rebuilding to the host and uploading again - and as clustering the original store."""
    from cutesv_amd import rebuild
    rng = np.random.default_rng(7)
    for seed, kw, p in ((41, dict(), Params.ont()),
                        (42, dict(n_sites=120, coverage=40, n_noise=20000, n_loci=2000, dup_frac=0.0), Params.ont(genotype=True))):
        st = synth.small_mixed(seed=seed, **kw)
        per = {}
        for (t, ch), (b, e) in st.seg_index.items():
            d = per.setdefault(t, dict(chrom=[], a=[], b=[], read_id=[], aux=[]))
            d["chrom"].append(np.full(e - b, st.chroms.index(ch))); d["a"].append(st.a[b:e]); d["b"].append(st.b[b:e])
            d["read_id"].append(st.read_id[b:e])
            d["aux"].append(st.aux[b:e] if t in ("INS", "INV", "TRA") else np.zeros(e - b, np.int32))
        for t, d in per.items():
            cols = {k: np.concatenate(v) for k, v in d.items()}
            n = len(cols["a"])
            if t == "INS":                                               # (INS duplicates would be tie groups: the host's business)
                perm = rng.permutation(n)
                per[t] = {k: v[perm] for k, v in cols.items()}
            else:
                dup = rng.integers(0, n, max(1, n // 20))
                perm = rng.permutation(n + len(dup))
                per[t] = {k: np.concatenate([v, v[dup]])[perm] for k, v in cols.items()}
        reads = None
        if st.reads_off is not None and p.genotype:
            ch_of = np.repeat(np.arange(len(st.chroms)), np.diff(st.reads_off))
            reads = dict(chrom=ch_of, start=st.r_start, end=st.r_end, primary=st.r_primary, read_id=st.r_id)

        def seg_of(t, ci, beg, end):
            rec = st.segment(t, st.chroms[ci], p).copy()
            rec["sig_begin"], rec["sig_end"] = beg, end
            return rec
        batch, tasks, src_row = rebuild.rebuild_to_device_batch(ctx, st.chroms, per, seg_of, reads=reads)
        got = ctx.cluster_batch(batch).trimmed()
        assert batch.n_sig == st.n_sig and len(src_row) == st.n_sig
        want = ctx.cluster_batch(st.host_batch(tasks, p)).trimmed()
        # (the segments of `tasks` are in the reference's order in both batches; signature indices are positions in the sorted
        # columns, which are the store's columns segment by segment)
        for k in ("call_seg", "bp1", "bp2", "support", "cipos", "cilen", "search_pos", "dr", "dv", "gl_idx", "support_off"):
            assert np.array_equal(got[k], want[k]), k
        assert got["n_clusters"] == want["n_clusters"] and len(got["bp1"]) > 20


def test_extraction_rows_stay_on_the_device_until_they_are_clustered(ctx):
    """main script :697-743 -> :750-857 -> :1113-1199 with the signature rows never in host memory: the CIGAR scan appends what it
    finds to the context's pool (CSV_CG_TO_POOL), rows made on the host join it (csv_pool_append), the rebuild sorts the pool
    (CSV_RB_FROM_POOL; read index -> rank of the read's name) and leaves the columns on the device (CSV_RB_KEEP_ON_DEVICE) for
    csv_cluster_batch (CSV_IN_DEVICE_COLUMNS).  Same rows and same calls as the chain through host memory."""
    from cutesv_amd import extract, rebuild
    from cutesv_amd.columns import TYPES
    rng = np.random.default_rng(11)
    n_chrom, tasks = 3, []
    base = 0
    for k in range(5):
        n = int(rng.integers(1500, 4000))
        off, cigar, start, use = synth.cigar_reads(n, seed=900 + k, mean_ops=60)
        start = rng.integers(0, 60_000, n).astype(np.int64)                    # (dense enough for clusters to form)
        qlen = rng.integers(200, 30_000, n).astype(np.int32)                   # some reads shorter than their insertions reach: clipped sequences
        tasks.append(dict(off=off, cigar=cigar, start=start, use=use, qlen=qlen, chrom=k % n_chrom, base=base, enc=synth.split_reads(n, seed=950 + k, n_chrom=n_chrom)))
        base += n
    n_reads = base
    names = ["read%07d" % x for x in rng.permutation(n_reads * 3)[:n_reads]]   # string order != extraction order
    rank = np.empty(n_reads, np.int32)
    rank[np.argsort(np.array(names), kind="stable")] = np.arange(n_reads, dtype=np.int32)
    seg_of = lambda t, ci: TYPES.index(t) * n_chrom + ci
    n_seg = len(TYPES) * n_chrom
    major = np.zeros(n_seg, np.uint8); nodedup = np.zeros(n_seg, np.uint8)
    for t in ("INV", "TRA"):
        major[TYPES.index(t) * n_chrom:(TYPES.index(t) + 1) * n_chrom] = 1
    nodedup[TYPES.index("INS") * n_chrom:(TYPES.index("INS") + 1) * n_chrom] = 1
    # a few rows the host made (stand-ins for split-read candidates), joined at the end
    extra = dict(seg=np.full(40, seg_of("DUP", 1), np.int32), a=rng.integers(0, 60_000, 40).astype(np.int64), b=rng.integers(1000, 90_000, 40).astype(np.int64),
                 read=rng.integers(0, n_reads, 40).astype(np.int32), aux=np.zeros(40, np.int32))
    extra = {k: np.concatenate([v, v[:7]]) for k, v in extra.items()}        # with duplicates: the rebuild drops them

    seg_base = [seg_of(t, 0) for t in ("DEL", "INS", "DUP", "INV", "TRA")]   # by candidate kind
    # ---- through host memory
    rows = {k: [] for k in ("seg", "a", "b", "read", "aux")}
    n_split = 0
    for t in tasks:
        sig = extract.cigar_signatures(ctx, t["off"], t["cigar"], t["start"], t["use"])
        ql = t["qlen"][sig["ins_read"]].astype(np.int64)
        seq = np.zeros(len(sig["ins_read"]), np.int64)
        for i, (p0, npc) in enumerate(zip(sig["ins_piece0"].tolist(), sig["ins_npiece"].tolist())):
            for q, l in zip(sig["piece_qoff"][p0:p0 + npc].tolist(), sig["piece_len"][p0:p0 + npc].tolist()):
                seq[i] += max(0, min(q + l, int(ql[i])) - min(q, int(ql[i])))                     # len(query_sequence[q : q + l])
        rows["seg"] += [np.full(len(seq), seg_of("INS", t["chrom"])), np.full(len(sig["del_read"]), seg_of("DEL", t["chrom"]))]
        rows["a"] += [sig["ins_pos"], sig["del_pos"]]; rows["b"] += [sig["ins_len"], sig["del_len"]]
        rows["read"] += [t["base"] + sig["ins_read"], t["base"] + sig["del_read"]]; rows["aux"] += [seq, np.zeros(len(sig["del_read"]), np.int64)]
        sp = extract.pool_rows_of_split(extract.split_signatures(ctx, t["enc"]), seg_base, t["base"], t["qlen"])      # the split-read candidates of the task
        for k in rows:
            rows[k].append(sp[k])
        n_split += len(sp["a"])
    assert n_split > 500
    for k in rows:
        rows[k].append(extra[k])
    cat = {k: np.concatenate(v) for k, v in rows.items()}
    assert ((cat["seg"] // n_chrom == TYPES.index("INS")) & (cat["aux"] != cat["b"])).sum() > 10   # clipped sequences exist
    want = rebuild.rebuild_columns(ctx, cat["seg"], cat["a"], cat["b"], rank[cat["read"]], cat["aux"], major, nodedup)

    # ---- on the device
    rebuild.pool_reset(ctx)
    for t in tasks:
        extract.cigar_signatures(ctx, t["off"], t["cigar"], t["start"], t["use"],
                                 pool=dict(seg_ins=seg_of("INS", t["chrom"]), seg_del=seg_of("DEL", t["chrom"]), read_base=t["base"], query_len=t["qlen"]))
        extract.split_signatures(ctx, t["enc"], pool=dict(seg_base=seg_base, read_base=t["base"], query_len=t["qlen"]))
    rebuild.pool_append(ctx, extra["seg"], extra["a"], extra["b"], extra["read"], extra["aux"])
    assert rebuild.pool_rows(ctx) == len(cat["a"])
    got = rebuild.rebuild_pool(ctx, rank, major, nodedup, keep_on_device=False)
    for k in ("seg_id", "a", "b", "read_id", "aux", "src_row"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["seg_count"], want["seg_count"]) and got["n_ins_ties"] == want["n_ins_ties"] and len(got["a"]) < len(cat["a"])

    # ---- ... and on into the clustering stage
    dev = rebuild.rebuild_pool(ctx, rank, major, nodedup, keep_on_device=True)
    off = np.r_[0, np.cumsum(dev["seg_count"])]
    segs = []
    for sgi in range(n_seg):
        if dev["seg_count"][sgi]:
            t = TYPES[sgi // n_chrom]
            segs.append(_abi.make_segment(t, sgi % n_chrom, int(off[sgi]), int(off[sgi + 1]), 200 if t == "DEL" else 100, 3, diff_ratio=0.3,
                                          sv_size=30, max_size=100000, min_support_reads=3))
    segs = np.array(segs, dtype=_abi.SEGMENT_DTYPE)
    batch = _abi.HostBatch.on_device(segs, dev["dev"], len(dev["src_row"]), n_chrom=n_chrom, keep=ctx)
    res_dev = ctx.cluster_batch(batch).trimmed()
    res_host = ctx.cluster_batch(_abi.HostBatch(segs, want["a"], want["b"], want["read_id"], want["aux"], n_chrom=n_chrom)).trimmed()
    for k in ("call_seg", "bp1", "bp2", "support", "cipos", "cilen", "search_pos", "seq_pick", "support_off", "support_sig"):
        assert np.array_equal(res_dev[k], res_host[k]), k
    assert len(res_dev["bp1"]) > 20
    rebuild.pool_reset(ctx)


def test_gpu_rebuild_identical_to_reference_rebuild(ctx):
    """csv_rebuild_signatures (+ the host finish of INS tie groups) on the raw, concatenated per-worker candidates ==
    the per-chromosome lists the reference's process_process_sigs_type wrote (rebuild_order.json.gz)"""
    from cutesv_amd import rebuild
    from cutesv_amd.columns import intern_names, BND_CODE
    from helpers import rebuild_case_inputs, rebuild_expected
    for case in load_json("rebuild_order.json.gz"):
        per, reads = rebuild_case_inputs(case)
        chroms = sorted({x[-1] for t in per for x in per[t]} | {x[2] for x in per["TRA"]} | {r[-1] for r in reads})
        cidx = {c: i for i, c in enumerate(chroms)}
        npos = {"DEL": 2, "INS": 2, "DUP": 2, "INV": 3, "TRA": 4}
        uniq, _ = intern_names([x[npos[t]] for t in per for x in per[t]] + [r[3] for r in reads])
        rank = {n: i for i, n in enumerate(uniq)}
        strands = sorted({x[0] for x in per["INV"]})
        cols = {}
        for t, lst in per.items():
            d = dict(chrom=[cidx[x[-1]] for x in lst], read_id=[rank[x[npos[t]]] for x in lst])
            if t in ("DEL", "DUP"):
                d.update(a=[int(x[0]) for x in lst], b=[int(x[1]) for x in lst], aux=[0] * len(lst))
            elif t == "INS":
                d.update(a=[int(x[0]) for x in lst], b=[int(x[1]) for x in lst], aux=[len(x[3]) for x in lst],
                         seq=[x[3] for x in lst], half=[int(x[0] != int(x[0])) for x in lst])
            elif t == "INV":
                d.update(a=[int(x[1]) for x in lst], b=[int(x[2]) for x in lst], aux=[strands.index(x[0]) for x in lst])
            else:
                d.update(a=[int(x[1]) for x in lst], b=[int(x[3]) for x in lst], aux=[cidx[x[2]] * 8 + BND_CODE[x[0]] for x in lst])
            cols[t] = d
        from cutesv_amd.columns import NameTable
        st, _ = rebuild.store_from_unsorted(ctx, chroms, cols, names=NameTable(uniq), strands=tuple(strands))
        got, _ = st.tuple_lists()
        want = rebuild_expected(case)
        assert set(want) == set(st.seg_index), case["name"]
        for (t, ch), rows in want.items():
            assert [x for x in got[t] if x[-1] == ch] == rows, (case["name"], t, ch)


def _rebuild_case_columns(case):
    """the raw candidates of a rebuild_order.json.gz case as unsorted per-type columns (+ names, strands, chromosomes)"""
    from cutesv_amd.columns import intern_names, BND_CODE
    from helpers import rebuild_case_inputs
    per, reads = rebuild_case_inputs(case)
    chroms = sorted({x[-1] for t in per for x in per[t]} | {x[2] for x in per["TRA"]} | {r[-1] for r in reads})
    cidx = {c: i for i, c in enumerate(chroms)}
    npos = {"DEL": 2, "INS": 2, "DUP": 2, "INV": 3, "TRA": 4}
    uniq, _ = intern_names([x[npos[t]] for t in per for x in per[t]] + [r[3] for r in reads])
    rank = {n: i for i, n in enumerate(uniq)}
    strands = sorted({x[0] for x in per["INV"]})
    cols = {}
    for t, lst in per.items():
        d = dict(chrom=[cidx[x[-1]] for x in lst], read_id=[rank[x[npos[t]]] for x in lst])
        if t in ("DEL", "DUP"):
            d.update(a=[int(x[0]) for x in lst], b=[int(x[1]) for x in lst], aux=[0] * len(lst))
        elif t == "INS":
            d.update(a=[int(x[0]) for x in lst], b=[int(x[1]) for x in lst], aux=[len(x[3]) for x in lst],
                     seq=[x[3] for x in lst], half=[int(x[0] != int(x[0])) for x in lst])
        elif t == "INV":
            d.update(a=[int(x[1]) for x in lst], b=[int(x[2]) for x in lst], aux=[strands.index(x[0]) for x in lst])
        else:
            d.update(a=[int(x[1]) for x in lst], b=[int(x[3]) for x in lst], aux=[cidx[x[2]] * 8 + BND_CODE[x[0]] for x in lst])
        cols[t] = d
    return per, cols, chroms, uniq, strands


def test_ins_ties_are_settled_without_the_columns_leaving_the_device(ctx):
    """main script :774-775, :958-969 through CSV_RB_KEEP_ON_DEVICE -> CSV_IN_DEVICE_COLUMNS: INS rows that agree in
    (chr, int(pos), len, read) are ordered by their sequences and de-duplicated on the whole tuple by the library's tie_order
    callback - only those rows' indices visit the host.  The sorted order (read back through src_row) is the reference's own
    (rebuild_order.json.gz, whose cases hold sequence-only and x.5-only differences), and clustering the device-resident
    columns gives the calls of the store the host-finished path builds.  r03 raised ValueError on the first tie."""
    from cutesv_amd import rebuild
    from cutesv_amd.columns import NameTable, TYPES
    from helpers import rebuild_expected
    p = Params.ont(min_support=2)
    tie_rows = 0
    for case in load_json("rebuild_order.json.gz"):
        per, cols, chroms, uniq, strands = _rebuild_case_columns(case)
        st, _ = rebuild.store_from_unsorted(ctx, chroms, cols, names=NameTable(uniq), strands=tuple(strands))     # host finish (r03)

        def seg_of(t, ci, beg, end):
            rec = st.segment(t, chroms[ci], p).copy()
            rec["sig_begin"], rec["sig_end"] = beg, end
            return rec
        batch, tasks, src_row = rebuild.rebuild_to_device_batch(ctx, chroms, cols, seg_of)
        # the device order, read back through src_row, is the reference's list for every (type, chromosome)
        base, flat = {}, []
        for t in TYPES:
            base[t] = len(flat)
            flat.extend((t, x) for x in per.get(t, []))
        want = rebuild_expected(case)
        got = {}
        for sr in src_row.tolist():
            t, x = flat[sr]
            row = tuple([int(x[0])] + list(x[1:])) if t in ("DEL", "INS", "DUP") else tuple(x)
            got.setdefault((t, x[-1]), []).append(row)
        assert set(got) == set(want), case["name"]
        for k, rows in want.items():
            assert got[k] == rows, (case["name"], k)
        # ... and the cluster stage reads those columns where they are
        assert batch.n_sig == st.n_sig
        a = ctx.cluster_batch(batch).trimmed()
        b = ctx.cluster_batch(st.host_batch(tasks, p)).trimmed()
        for f in ("call_seg", "bp1", "bp2", "support", "cipos", "cilen", "search_pos", "seq_pick", "support_off", "support_sig"):
            assert np.array_equal(a[f], b[f]), (case["name"], f)
        ins = cols["INS"]
        key = list(zip(ins["chrom"], ins["a"], ins["b"], ins["read_id"]))
        tie_rows += len(key) - len(set(key))
    assert tie_rows > 0                                                 # (the fixture does hold tie groups)


def test_cigar_scan_identical_to_reference_and_oracle(ctx):
    """csv_cigar_signatures (8f row 4): the reference's candidate lists on the golden reads, and the oracle's arrays bit for
    bit on a large random batch (reads of 1 .. 5000 operations, so that the 64-operation steps, the carried merge state
    and the two-pass offsets are all exercised), plus the empty shapes"""
    from cutesv_amd import extract
    from helpers import assert_cigar_case
    for case in load_json("cigar_sigs.json.gz"):
        assert_cigar_case(case, lambda *a, **k: extract.cigar_signatures(ctx, *a, **k))
    rng = np.random.default_rng(77)
    n = 20000
    nops = np.minimum(rng.geometric(1 / 60.0, n), 5000).astype(np.int64)
    nops[rng.integers(0, n, 50)] = 0                                    # reads without a CIGAR
    off = np.zeros(n + 1, np.int64); np.cumsum(nops, out=off[1:])
    tot = int(off[-1])
    op = rng.choice(np.array([0, 1, 2, 3, 4, 5, 6, 7, 8], np.uint32), tot, p=[0.4, 0.2, 0.2, 0.02, 0.02, 0.02, 0.02, 0.06, 0.06])
    ln = np.where(rng.random(tot) < 0.2, rng.integers(10, 3000, tot), rng.integers(1, 10, tot)).astype(np.uint32)
    cigar = (ln << np.uint32(4)) | op
    start = rng.integers(0, 200_000_000, n).astype(np.int64)
    use = (rng.random(n) < 0.9).astype(np.uint8)
    for kw in (dict(min_siglength=10, merge_ins_threshold=100, merge_del_threshold=0), dict(min_siglength=1, merge_ins_threshold=5000, merge_del_threshold=5000),
               dict(min_siglength=30, merge_ins_threshold=0, merge_del_threshold=300)):
        want = _oracle().cigar_signatures(off, cigar, start, use, **kw)
        got = extract.cigar_signatures(ctx, off, cigar, start, use, **kw)
        for k in want:
            if k != "ms_device":
                assert np.array_equal(got[k], want[k]), (k, kw)
        assert len(got["del_pos"]) > 1000 and len(got["ins_pos"]) > 1000
    empty = extract.cigar_signatures(ctx, np.zeros(1, np.int64), np.zeros(0, np.uint32), np.zeros(0, np.int64))
    assert len(empty["ins_pos"]) == 0 and len(empty["del_pos"]) == 0


def test_run_tra_shim_genotypes_like_the_reference(ctx, tmp_path, monkeypatch):
    """run_tra(action=True): clustering on the GPU + the reference's call_gt loop over the BAM (default), or everything on
    the GPU from the reads table (CUTESV_AMD_TRA_GT=reads_table); both give the reference's rows on the golden cases
    (whose BAM stand-in IS the reads table)"""
    import sys
    import types
    from cutesv_amd import resolve
    from helpers import write_reference_workdir
    from test_host_logic import _StubBam
    stub = types.ModuleType("pysam")
    stub.AlignmentFile = _StubBam
    monkeypatch.setitem(sys.modules, "pysam", stub)
    monkeypatch.setattr(resolve, "_ctx", ctx)
    for case in load_json("tra_genotype.json.gz")[:4]:
        st = store_from_json(case["store"])
        _StubBam.store = st
        p = Params(**case["params"])
        d = str(tmp_path / case["name"]) + "/"
        os.makedirs(d)
        idx = write_reference_workdir(st, d)
        for mode in ("bam", "reads_table"):
            monkeypatch.setenv("CUTESV_AMD_TRA_GT", mode)
            if mode == "reads_table":
                from cutesv_amd import bam_header
                monkeypatch.setattr(bam_header, "reference_lengths", lambda path: {c: int(l) for c, l in zip(st.chroms, st.contig_len)})
            for t, c, want in case["rows"]:
                if t != "TRA":
                    continue
                got = resolve.run_tra((d, c, p.min_support, p.diff_ratio_filtering_TRA, p.max_cluster_bias_TRA, "stub.bam",
                                       True, p.gt_round, idx))
                assert got[0] == c
                assert_rows_equal("TRA", got[1], want, where="run_tra %s %s %s" % (mode, case["name"], c))


def test_runs_with_and_without_the_tier_peek(ctx, monkeypatch):
    """A one-shot call whose column copies are still on the link waits for THIS run's k_chain_apply to say whether the tiers
    above 64 signatures have work and queues k_refine<64,256> / k_refine<256,2048> only then (r06: nothing is carried over
    from earlier runs; a resident run queues every tier).  Must not change a byte - with clusters above 64 signatures present
    or not, with the peek or without (CSV_NO_PEEK), one-shot from page-locked columns (bulk and gate-first), one-shot from
    pageable columns, resident."""
    for st, p in ((synth.small_mixed(seed=3), Params.ont(genotype=True)),
                  (synth.small_mixed(seed=4, coverage=150, n_sites=20), Params.ont(genotype=True, min_support=3))):
        hb = st.host_batch(st.tasks(), p)
        phb = _pinned_batch(hb)
        want = _oracle().cluster_batch(hb, per_sig=True).trimmed()
        for env in (None, "1"):
            if env:
                monkeypatch.setenv("CSV_NO_PEEK", env)
            else:
                monkeypatch.delenv("CSV_NO_PEEK", raising=False)
            ctx.upload(hb, per_sig=True)
            for _ in range(3):
                ctx.run()
            assert_soa_equal(ctx.download(per_sig=True).trimmed(), want)
            for lazy_min in ("0", "1000000000"):
                monkeypatch.setenv("CSV_LAZY_MIN", lazy_min)
                for _ in range(2):
                    assert_soa_equal(ctx.cluster_batch(phb, per_sig=True, reuse=True).trimmed(), want)
            monkeypatch.delenv("CSV_LAZY_MIN", raising=False)
            assert_soa_equal(ctx.cluster_batch(hb, per_sig=True).trimmed(), want)
    sizes = np.bincount(want["cluster_id"][want["cluster_id"] >= 0])
    assert sizes.max() > 64                                   # the second workload does exercise the big tiers


def _random_split_batch(rng, n_reads, n_chrom=6):
    """flat csv_split_in arrays of random reads: 0 .. 12 entries each, with or without a primary, segments that tile the
    read loosely and wander over a few chromosomes / both strands (every rule of the analysis fires somewhere)"""
    n_ent = rng.choice(np.array([0, 1, 2, 2, 2, 3, 3, 3, 4, 4, 5, 6, 8, 12]), n_reads)
    ent_off = np.zeros(n_reads + 1, np.int64); np.cumsum(n_ent, out=ent_off[1:])
    ne = int(ent_off[-1])
    read = np.repeat(np.arange(n_reads), n_ent)
    first = np.zeros(ne, bool); first[ent_off[:-1][n_ent > 0]] = True
    L = rng.integers(500, 30000, n_reads).astype(np.int64)
    primary = (first & (rng.random(ne) < 0.8)).astype(np.uint8)
    strand = (rng.random(ne) < 0.35).astype(np.uint8)
    # reads mostly keep their strand and chromosome
    base_st = np.repeat((rng.random(n_reads) < 0.5).astype(np.uint8), n_ent); strand = np.where(rng.random(ne) < 0.7, base_st, strand).astype(np.uint8)
    base_ch = np.repeat(rng.integers(0, n_chrom, n_reads), n_ent)
    chrom = np.where(rng.random(ne) < 0.8, base_ch, rng.integers(0, n_chrom, ne)).astype(np.int32)
    Lr = L[read]
    lo = (rng.random(ne) * Lr).astype(np.int64); hi = np.minimum(Lr, lo + 1 + (rng.random(ne) * Lr * 0.5).astype(np.int64))
    base_ref = np.repeat(rng.integers(100_000, 3_000_000, n_reads), n_ent)
    ref = np.maximum(0, base_ref + np.where(rng.random(ne) < 0.7, lo + rng.integers(-3000, 3000, ne), rng.integers(-2_000_000, 2_000_000, ne))).astype(np.int64)
    span = np.maximum(1, hi - lo + rng.integers(-50, 50, ne)).astype(np.int64)
    # primary: c0/c1 = read_start/read_end, f0/f1 = ref_start/ref_end; SA entry: clips (as the strand has them), start, span
    c0 = np.where(primary == 1, lo, np.where(strand == 0, lo, Lr - hi)); c1 = np.where(primary == 1, hi, np.where(strand == 0, Lr - hi, lo))
    f1 = np.where(primary == 1, ref + span, span)
    mapq = rng.choice(np.array([0, 3, 20, 30, 60], np.int32), ne)
    return dict(ent_off=ent_off, read_len=L, c0=c0.astype(np.int64), c1=c1.astype(np.int64), f0=ref, f1=f1.astype(np.int64), chr=chrom, mapq=mapq,
                strand=strand, primary=primary)


def test_split_read_analysis_identical_to_reference_and_oracle(ctx):
    """csv_split_signatures (8f row 4): the reference's five candidate lists on the golden reads, and the oracle's arrays
    bit for bit on large random batches under several parameter sets, plus the empty shapes"""
    from cutesv_amd import extract
    from helpers import assert_split_case
    for case in load_json("split_sigs.json.gz"):
        assert_split_case(case, lambda enc, **k: extract.split_signatures(ctx, enc, **k))
    rng = np.random.default_rng(91)
    enc = _random_split_batch(rng, 60000)
    seen = set()
    for kw in (dict(sv_size=30, min_mapq=20, max_split_parts=7, max_size=100000), dict(sv_size=1, min_mapq=0, max_split_parts=-1, max_size=-1),
               dict(sv_size=200, min_mapq=30, max_split_parts=3, max_size=5000), dict(sv_size=0, min_mapq=61, max_split_parts=12, max_size=100)):
        got, want = extract.split_signatures(ctx, enc, **kw), _oracle().split_signatures(enc, **kw)
        for k in ("kind", "read", "chr", "aux", "a", "b", "c", "d"):
            assert np.array_equal(got[k], want[k]), (kw, k, len(got[k]), len(want[k]))
        seen |= set(np.unique(got["kind"]).tolist())
        assert len(got["kind"]) > 0 or kw["min_mapq"] > 60
    assert seen == {0, 1, 2, 3, 4}
    empty = {k: v[:0] for k, v in enc.items() if k not in ("ent_off", "read_len")}
    assert len(extract.split_signatures(ctx, dict(empty, ent_off=np.zeros(1, np.int64), read_len=np.zeros(0, np.int64)))["kind"]) == 0
    assert len(extract.split_signatures(ctx, dict(empty, ent_off=np.zeros(4, np.int64), read_len=np.full(3, 1000, np.int64)))["kind"]) == 0


def test_whole_parse_read_on_the_gpu(ctx):
    """the same golden through csv_cigar_signatures + csv_split_signatures"""
    from cutesv_amd import extract
    from helpers import assert_parse_case
    for case in load_json("parse_reads.json.gz"):
        assert_parse_case(case, lambda *a, **k: extract.cigar_signatures(ctx, *a, **k), lambda enc, **k: extract.split_signatures(ctx, enc, **k))


def test_resident_reruns_keep_the_reads_order_state(ctx):
    """upload / run / run / download on a reads table in RANDOM order inside every block: the first run finds out that the
    table needs the general sort; a re-run that keeps the packed table of the upload (CSV_OPT_REUSE_READS_ORDER) must still
    know, and the download must still take the fallback - found by scripts/stress_gpu.py, where the second run genotyped
    from a table that had never been written"""
    import dataclasses
    rng = np.random.default_rng(3)
    st = synth.small_mixed(seed=70115, n_sites=40, coverage=20, genotype=True)
    perm = np.arange(st.n_reads)
    for c in range(len(st.chroms)):
        lo, hi = int(st.reads_off[c]), int(st.reads_off[c + 1])
        perm[lo:hi] = lo + rng.permutation(hi - lo)
    st = dataclasses.replace(st, r_start=st.r_start[perm], r_end=st.r_end[perm], r_primary=st.r_primary[perm], r_id=st.r_id[perm])
    p = Params.ont(genotype=True)
    hb = st.host_batch(st.tasks(), p)
    want = _oracle().cluster_batch(hb, per_sig=True).trimmed()
    for reuse in (1, 0):
        ctx.option(1, reuse)
        ctx.upload(hb, per_sig=True)
        ctx.run(); ctx.run(); ctx.run()
        got = ctx.download(per_sig=True).trimmed()
        assert ctx.last_reads_mode() == 2
        assert_soa_equal(got, want, store=st, set_order_segments=())
        ctx.run()                                              # and after the fallback: the general sort's table is kept
        assert_soa_equal(ctx.download(per_sig=True).trimmed(), want, store=st, set_order_segments=())
    ctx.option(1, 1)


def test_resident_reruns_of_deep_and_shallow_batches(ctx):
    """a deep batch (every refine tier, cover sets beyond the small hash tables: the genotype overflow pass has work) and a
    shallow one, uploaded alternately and run several times each, must give the oracle's result every time (r05 launched the
    upper tiers and the overflow pass only when an earlier run had asked for them; r06 launches them in every resident run)"""
    p = Params.ont(genotype=True, min_support=3)
    deep = synth.small_mixed(seed=4711, n_sites=6, coverage=900, genotype=True)
    shallow = synth.small_mixed(seed=4712, n_sites=40, coverage=12, genotype=True)
    cases = []
    for st in (deep, shallow):
        hb = st.host_batch(st.tasks(), p)
        cases.append((st, hb, _oracle().cluster_batch(hb, per_sig=True).trimmed()))
    for rep in range(2):
        for st, hb, want in cases:
            ctx.upload(hb, per_sig=True)
            for _ in range(4):
                ctx.run()
                ctx.sync()                                     # (the answers of the run are on the host before the next one is planned)
                assert_soa_equal(ctx.download(per_sig=True).trimmed(), want, store=st)
            for _ in range(3):
                ctx.run()                                      # ... and with the host running ahead of the device
            assert_soa_equal(ctx.download(per_sig=True).trimmed(), want, store=st)


def test_single_pipe_on_the_gpu(ctx):
    """one extraction task (main script :697-743) with the HIP CIGAR scan and split-read analysis: candidates and reads table
    rows equal the reference's single_pipe"""
    from cutesv_amd import extract
    from helpers import assert_single_pipe_case
    for case in load_json("single_pipe.json.gz"):
        assert_single_pipe_case(case, lambda *a, **k: extract.cigar_signatures(ctx, *a, **k), lambda enc, **k: extract.split_signatures(ctx, enc, **k))


def test_hundreds_of_chromosomes_with_genotyping(ctx, monkeypatch):
    """an assembly with 400 contigs, genotyped: more chromosomes than k_genotype keeps block offsets for in LDS, 400 block starts
    for the reads-order plan to merge with the runs it finds (reads in extraction order), segments of a few dozen signatures"""
    st = synth.small_mixed(seed=77, n_sites=1000, coverage=14, n_contigs=400, contig_len=120_000, n_noise=3000, n_loci=300)
    p = Params.ont(genotype=True, genotype_tra=True, min_support=3)
    got = _compare_soa(ctx, st, p)
    assert len(got["bp1"]) > 3000 and (got["dr"] >= 0).sum() > 3000 and len(st.tasks()) > 1500
    runs, _ = synth.extraction_order(st, region=60_000, workers=3)           # two task regions per contig
    assert (np.diff(runs.r_start) < 0).sum() > 500                           # descents at the 399 block starts and inside the blocks
    monkeypatch.setenv("CSV_READS_GAP", "20000")
    got2 = _compare_soa(ctx, runs, p)
    assert ctx.last_reads_mode() == 1 and len(got2["bp1"]) == len(got["bp1"])


def test_thousands_of_small_segments(ctx):
    """a reference with thousands of small contigs: 3000 (contig, type) segments of a few dozen signatures each, so that
    every chain tile spans dozens of segments (the per-row path, the segment search bounded by the tile's range, the
    look-back across segment boundaries); every field against the oracle"""
    from cutesv_amd.columns import SigStore
    rng = np.random.default_rng(12)
    per = {"DEL": [], "INS": []}
    for c in range(1500):
        ch = "ctg%05d" % c
        for site in range(int(rng.integers(1, 4))):
            pos = 1000 + site * 5000
            for r in range(int(rng.integers(3, 14))):
                per["DEL"].append((pos + int(rng.integers(-5, 5)), 300 + int(rng.integers(-3, 3)), "d%d_%d_%d" % (c, site, r), "DEL", ch))
                per["INS"].append((pos + 2000 + int(rng.integers(-5, 5)), 200 + int(rng.integers(-3, 3)), "i%d_%d_%d" % (c, site, r), "ACGT" * 50, "INS", ch))
    st = SigStore.from_tuple_lists(per, [])
    got = _compare_soa(ctx, st, Params.ont(min_support=3))
    assert len(st.tasks()) == 3000 and len(got["bp1"]) > 1500


def test_narrow_input_columns(ctx):
    """CSV_IN_SIG_I32 / CSV_IN_READS_I32: a page-locked store keeps int32 twins of the position / length columns (a third
    less data on the link) and the library widens them on the device: same calls as the int64 columns; a store whose
    values do not fit keeps its int64 columns"""
    st = synth.small_mixed(seed=21, genotype=True)
    p = Params.ont(genotype=True)
    wide = st.host_batch(st.tasks(), p)
    pst = st.pinned()
    narrow = pst.host_batch(pst.tasks(), p)
    assert narrow.a.dtype == np.int32 and narrow.r_start.dtype == np.int32
    assert narrow.c.flags & _abi.IN_SIG_I32 and narrow.c.flags & _abi.IN_READS_I32 and not (wide.c.flags & (_abi.IN_SIG_I32 | _abi.IN_READS_I32))
    for per_sig in (False, True):
        a, b = ctx.cluster_batch(wide, per_sig=per_sig).trimmed(), ctx.cluster_batch(narrow, per_sig=per_sig, reuse=True).trimmed()
        assert_soa_equal(b, a)
    ctx.upload(narrow); ctx.run(); ctx.run()
    assert_soa_equal(ctx.download().trimmed(), ctx.cluster_batch(wide).trimmed())
    assert_soa_equal(ctx.cluster_batch(narrow, per_sig=True).trimmed(), _oracle().cluster_batch(narrow, per_sig=True).trimmed(), st)
    # a length beyond 31 bits: no narrow twins for the signature columns (the reads keep theirs)
    big = synth.small_mixed(seed=21, genotype=True)
    big.b[5] = 1 << 33
    pbig = big.pinned()
    hb = pbig.host_batch(pbig.tasks(), p)
    assert hb.a.dtype == np.int64 and hb.r_start.dtype == np.int32
    assert_soa_equal(ctx.cluster_batch(hb, per_sig=True).trimmed(), _oracle().cluster_batch(hb, per_sig=True).trimmed(), big)


# ---------------------------------------------------------------------------------------------- gate-first calls, slim results (ABI v7)
def _pinned_batch(hb, rows8=False):
    """the same batch with every column in page-locked memory (what csv_cluster_batch needs to take the gate-first form);
    rows8: int32 positions / lengths and the interleaved {b, read_id} array next to them (ABI v8: the fetch reads that one)"""
    pc = engine.pinned_copy
    kw = {}
    if hb.reads_off is not None:
        kw = dict(reads_off=hb.reads_off, r_start=pc(hb.r_start), r_end=pc(hb.r_end), r_primary=pc(hb.r_primary), r_id=pc(hb.r_id))
    a, b = pc(hb.a), pc(hb.b)
    if rows8 and hb.n_sig and hb.a.dtype == np.int64 and np.abs(hb.a).max() < (1 << 31) and np.abs(hb.b).max() < (1 << 31):
        a, b = pc(hb.a.astype(np.int32)), pc(hb.b.astype(np.int32))
        r8 = engine.pinned_empty((hb.n_sig, 2), np.int32)
        r8[:, 0], r8[:, 1] = b, hb.read_id
        kw["rows8"] = r8
    return _abi.HostBatch(hb.segments, a, b, pc(hb.read_id), pc(hb.aux), n_chrom=hb.n_chrom, contig_len=hb.contig_len,
                          per_sig=bool(hb.c.flags & _abi.IN_PER_SIG), reads_sorted=bool(hb.c.flags & _abi.IN_READS_SORTED), **kw)


def _gate_first_engine(ctx, rows8=False):
    def run(hb):
        res = ctx.cluster_batch(_pinned_batch(hb, rows8=rows8), per_sig=True)
        assert ctx.lazy_info()[0], "the call did not take the gate-first form"
        return res
    return run


def test_gate_first_rows_identical_to_reference(ctx, monkeypatch):
    """csv_cluster_batch from page-locked columns copies only the position column and fetches the rows of the clusters that pass
    the size gate out of the caller's columns (k_lazy_zero / k_lazy_fetch): every golden case of the reference - the (0,0)
    sentinel look-alikes, INV / TRA chains, genotyping - gives the reference's rows through that form too"""
    monkeypatch.setenv("CSV_LAZY_MIN", "0")
    for fixture in ("small_cases.json.gz", "known_answers.json"):
        for case in load_json(fixture):
            st = store_from_json(case["store"])
            p = Params(**case["params"])
            want = {(t, c): r for t, c, r in case["rows"]}
            for rows8 in (False, True):                    # the fetch out of the columns / out of the interleaved {b, read_id} rows
                got, res, hb = rows_by_task(st, p, _gate_first_engine(ctx, rows8=rows8), tasks=list(want.keys()))
                for key in want:
                    assert_rows_equal(key[0], got[key], want[key], where="gate-first %s %s" % (case["name"], key))


def _gate_first_vs_oracle(ctx, st, p):
    hb = st.host_batch(st.tasks(), p)
    want = _oracle().cluster_batch(hb, per_sig=True).trimmed()
    got8 = ctx.cluster_batch(_pinned_batch(hb, rows8=True), per_sig=True)
    assert ctx.lazy_info()[0]
    assert_soa_equal(got8.trimmed(), want, store=st)
    phb = _pinned_batch(hb)
    got = ctx.cluster_batch(phb, per_sig=True)
    assert ctx.lazy_info()[0]
    assert_soa_equal(got.trimmed(), want, store=st)
    # a resident re-run of the same upload takes the device columns as the fetch left them (the host columns are not read again)
    ctx.run()
    assert_soa_equal(ctx.download(per_sig=True).trimmed(), want, store=st)
    with pytest.raises(engine.CsvError):
        ctx.validate()                                    # (the order check needs whole columns: a csv_batch_upload)
    return got.trimmed()


def test_gate_first_vs_oracle_every_tier(ctx, monkeypatch):
    from cutesv_amd.columns import SigStore
    monkeypatch.setenv("CSV_LAZY_MIN", "0")
    # all five types with genotyping, three presets
    for seed, p in ((101, Params(genotype=True)), (102, Params.ont(genotype=True)), (103, Params.hifi(genotype=True, min_support=3))):
        _gate_first_vs_oracle(ctx, synth.small_mixed(seed=seed, dup_frac=0.2, n_loci=200), p)
    # clusters of thousands of signatures (the LDS and the global-scratch tiers): fetched by every chain tile they cover
    st = synth.small_mixed(seed=77, n_sites=30, coverage=60, n_noise=30000, n_loci=3000, contig_len=400_000, dup_frac=0.3)
    got = _gate_first_vs_oracle(ctx, st, Params(max_cluster_bias_DEL=3000, max_cluster_bias_INS=3000, max_cluster_bias_DUP=20000,
                                                 max_cluster_bias_INV=20000, max_cluster_bias_TRA=5000, min_support=3, genotype=False))
    assert np.bincount(got["cluster_id"][got["cluster_id"] >= 0]).max() > 2048
    # a 12 000x pile-up next to 30x loci, genotyped
    base = synth.small_mixed(seed=5, coverage=30, n_contigs=2, contig_len=1_000_000)
    pile = synth.small_mixed(seed=6, coverage=12000, n_sites=3, n_contigs=2, contig_len=40_000, n_noise=0, n_loci=0, dup_frac=0.02)
    _gate_first_vs_oracle(ctx, synth.concat_stores(base, pile), Params(genotype=True, min_support=10, max_cluster_bias_DEL=200))
    # batch sizes on / next to the chain tile boundary; the last cluster is long
    for n in (2048, 4096, 4097, 6143, 64, 1):
        rng = np.random.default_rng(n)
        per = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
        pos, i = 1000, 0
        while i < n:
            m = min(int(rng.choice([1, 2, 5, 12, 20, 40, 70, 300])) if n - i > 400 else n - i, n - i)
            for j in range(m):
                per["DEL"].append((pos + int(rng.integers(0, 30)), int(rng.integers(40, 60)), "q%06d" % (i + j), "DEL", "1"))
            pos += 5000
            i += m
        _gate_first_vs_oracle(ctx, SigStore.from_tuple_lists(per), Params.ont(min_support=3))
    # thousands of small segments: chain tiles that span dozens of segments (the per-row source mapping)
    rng = np.random.default_rng(12)
    per = {"DEL": [], "INS": []}
    for c in range(700):
        ch = "ctg%05d" % c
        for site in range(int(rng.integers(1, 4))):
            pos = 1000 + site * 5000
            for r in range(int(rng.integers(3, 14))):
                per["DEL"].append((pos + int(rng.integers(-5, 5)), 300 + int(rng.integers(-3, 3)), "d%d_%d_%d" % (c, site, r), "DEL", ch))
                per["INS"].append((pos + 2000 + int(rng.integers(-5, 5)), 200 + int(rng.integers(-3, 3)), "i%d_%d_%d" % (c, site, r), "ACGT" * 50, "INS", ch))
    _gate_first_vs_oracle(ctx, SigStore.from_tuple_lists(per, []), Params.ont(min_support=3))
    # int64 columns (a length beyond 31 bits keeps the wide columns): the other instantiation of the fetch
    big = synth.small_mixed(seed=21, genotype=True)
    big.b[5] = 1 << 33
    hb = big.host_batch(big.tasks(), Params.ont(genotype=True))
    assert hb.a.dtype == np.int64
    _gate_first_vs_oracle(ctx, big, Params.ont(genotype=True))


def test_gate_first_zero_positions(ctx, monkeypatch):
    """the one chain predicate of a DEL / INS / DUP segment that reads a length: an element equal to the reference's [0, 0, '']
    sentinel (INDEL:62-64) - position 0 AND length 0 - starts a new cluster behind it and silences the cluster it ends.  The
    gate-first form fetches `b` of the rows at position 0 before the chain kernels run (k_lazy_zero)."""
    from cutesv_amd.columns import SigStore
    monkeypatch.setenv("CSV_LAZY_MIN", "0")
    per = {t: [] for t in ("DEL", "INS", "DUP", "INV", "TRA")}
    for ch, zero_len in (("1", 0), ("2", 7), ("3", 0)):
        n0 = 0
        for i in range(6):                                  # position 0: lengths 0 (the look-alike) or not
            per["DEL"].append((0, zero_len if i < 4 else 35, "z%s_%d" % (ch, i), "DEL", ch)); n0 += 1
            per["INS"].append((0, zero_len if i < 3 else 40, "y%s_%d" % (ch, i), "ACGT" * 10, "INS", ch))
        for i in range(12):
            per["DEL"].append((3 + i, 50 + i % 3, "a%s_%d" % (ch, i), "DEL", ch))
            per["INS"].append((2 + i, 60 + i % 2, "b%s_%d" % (ch, i), "ACGT" * 20, "INS", ch))
        for i in range(10):
            per["DEL"].append((5000 + i, 80, "c%s_%d" % (ch, i), "DEL", ch))
    st = SigStore.from_tuple_lists(per)
    for p in (Params.ont(min_support=3), Params(min_support=2, max_cluster_bias_DEL=10, max_cluster_bias_INS=10)):
        got = _gate_first_vs_oracle(ctx, st, p)
        monkeypatch.setenv("CSV_NO_LAZY", "1")
        hb = st.host_batch(st.tasks(), p)
        bulk = ctx.cluster_batch(_pinned_batch(hb), per_sig=True)
        assert not ctx.lazy_info()[0]
        assert_soa_equal(bulk.trimmed(), got, store=st)
        monkeypatch.delenv("CSV_NO_LAZY")


@pytest.mark.parametrize("genotype", [False, True])
def test_slim_results_equal_the_full_ones(ctx, genotype, monkeypatch):
    """ABI v7: CSV_OUT_NO_SUPPORT_LIST, CSV_OUT_COORD_I32 and NULL optional fields change what crosses PCIe, not a value: every
    array that is asked for equals the full result's, through k_publish (page-locked arrays) and through the copy path"""
    st = synth.small_mixed(seed=41, genotype=genotype, n_loci=150)
    p = Params.ont(genotype=genotype)
    pst = st.pinned()
    hb = pst.host_batch(pst.tasks(), p)
    assert hb.a.dtype == np.int32
    full = ctx.cluster_batch(hb).trimmed()
    forms = [dict(no_support=True), dict(coord32=True), dict(no_support=True, coord32=True, fields=("call_aux", "cipos", "cilen", "seq_pick")),
             dict(fields=()), dict(no_support=True, coord32=True, fields=("dr", "dv", "gl_idx"))]
    for kw in forms:
        for reuse in (True, False):                         # page-locked (published in place) / pageable (copied and unpacked)
            ctx._res_cache = None
            r = ctx.cluster_batch(hb, reuse=reuse, **kw)
            t = r.trimmed()
            assert r.n_calls == len(full["bp1"]) and r.n_support == len(full["support_sig"])
            for name in ("call_seg", "bp1", "bp2", "support") + tuple(kw.get("fields", _abi.OPTIONAL_CALL_FIELDS)):
                assert t[name] is not None and np.array_equal(t[name].astype(np.int64), full[name].astype(np.int64)), (kw, reuse, name)
                if kw.get("coord32") and name in _abi.COORD_FIELDS:
                    assert t[name].dtype == np.int32
            for name in _abi.OPTIONAL_CALL_FIELDS:
                if "fields" in kw and name not in kw["fields"]:
                    assert t[name] is None
            if kw.get("no_support"):
                assert t["support_off"] is None and t["support_sig"] is None
            else:
                assert np.array_equal(t["support_off"], full["support_off"]) and np.array_equal(t["support_sig"], full["support_sig"])
    # int32 coordinates need int32 columns
    wide = st.host_batch(st.tasks(), p)
    with pytest.raises(engine.CsvError):
        ctx.cluster_batch(wide, coord32=True)
    # resident: deliver slim into recycled page-locked arrays
    ctx.upload(hb); ctx.run()
    res = ctx.result_buffers(no_support=True, coord32=True, fields=("call_aux", "cipos", "cilen", "seq_pick", "dr", "gl_idx"))
    t = ctx.download(into=res).trimmed()
    for name in ("call_seg", "bp1", "bp2", "support", "call_aux", "cipos", "cilen", "seq_pick", "dr", "gl_idx"):
        assert np.array_equal(t[name].astype(np.int64), full[name].astype(np.int64)), name


def test_download_into_checks_the_batch_shape(ctx):
    """recycled result arrays that are too small for the uploaded batch's segments / signatures are refused (seg_status and the
    per-signature arrays have no capacity field of their own)"""
    small = synth.small_mixed(seed=3, n_contigs=2)
    large = synth.small_mixed(seed=4, n_contigs=4)
    p = Params.ont()
    ctx.upload(small.host_batch(small.tasks(), p), per_sig=True)
    res = ctx.result_buffers(per_sig=True)
    ctx.run(); ctx.download(per_sig=True, into=res)
    ctx.upload(large.host_batch(large.tasks(), p), per_sig=True)
    ctx.run()
    with pytest.raises(ValueError):
        ctx.download(per_sig=True, into=res)


def test_lazy_stage_equals_the_row_lists(ctx):
    """resolve.cluster_stage(lazy=True) on the HIP path: the same rows as the materialised stage, and the native emitter gives the
    same VCF text from the lazy objects' arrays (vcf.emit_stage) as from the result itself"""
    from cutesv_amd import resolve, vcf
    st = synth.small_mixed(seed=52, genotype=True).pinned()
    p = Params.ont(genotype=True)
    eager = resolve.cluster_stage(st, p, ctx=ctx)
    lazy = resolve.cluster_stage(st, p, ctx=ctx, lazy=True)
    again = resolve.cluster_stage(st, p, ctx=ctx, lazy=True)          # (the first result's private arrays survive the recycled ones)
    assert set(lazy) == set(eager) and sum(len(v) for v in eager.values()) > 100
    for ch in eager:
        assert lazy[ch] == eager[ch] and again[ch] == eager[ch]
        a = list(eager[ch]); a.sort(key=lambda x: int(x[2]))
        lazy[ch].sort(key=lambda x: int(x[2]))
        assert lazy[ch] == a and lazy[ch].backing() is not None
    # the rows read the context's recycled result in place (Context.lend): it is out of recycling while they live, back after
    held = next(iter(again.values())).backing().res
    assert held is not ctx._res_cache
    del lazy
    import gc
    gc.collect()
    third = resolve.cluster_stage(st, p, ctx=ctx, lazy=True)
    for ch in eager:
        assert again[ch] == eager[ch] and third[ch] == eager[ch]
    assert next(iter(again.values())).backing().res is held
    ref = {c: synth.reference_sequence(3_200_000, seed=9 + i) for i, c in enumerate(st.chroms)}
    hb = st.host_batch(st.tasks(), p)
    kw = dict(min_size=p.min_size, max_size=p.max_size, genotype=True)
    want, _ = vcf.emit_records(st, hb.segments, ctx.cluster_batch(hb), ref, **kw)
    got, _ = vcf.emit_stage(again, ref, **kw)
    assert got == want and got.count("\n") > 50
    slim, _ = vcf.emit_records(st, hb.segments, ctx.cluster_batch(hb, reuse=True, no_support=True, coord32=True,
                                                                  fields=("call_aux", "cipos", "cilen", "seq_pick", "dr", "gl_idx")), ref, **kw)
    assert slim == want


@pytest.mark.parametrize("block", [True, False])
@pytest.mark.parametrize("genotype", [False, True])
def test_pipelined_delivery_equals_the_serial_download(ctx, genotype, block):
    """csv_batch_publish_async / _wait: run k's result reaches the caller's page-locked arrays while run k + 1 computes (two result
    arenas on the device) - moved by the copy engine when the arrays sit back to back in one block (block=True: result_buffers'
    layout), written in place by k_publish on a stream of its own when they are scattered.  Every delivered result equals the
    synchronous download, whatever the interleaving; the state errors are errors"""
    st = synth.small_mixed(seed=61, genotype=genotype, n_loci=300).pinned()
    p = Params.ont(genotype=genotype)
    hb = st.host_batch(st.tasks(), p)
    want = _oracle().cluster_batch(hb, per_sig=False).trimmed()
    ctx.upload(hb, per_sig=False)
    ctx.run()
    a, b = ctx.result_buffers(block=block), ctx.result_buffers(block=block)
    with pytest.raises(engine.CsvError):
        ctx.publish_async(a)                               # (not before one synchronous download of the upload)
    assert_soa_equal(ctx.download().trimmed(), want, store=st)
    with pytest.raises(engine.CsvError):
        ctx.publish_wait()                                 # (nothing in flight)
    bufs = [a, b]
    ctx.run(); ctx.publish_async(bufs[0])
    for k in range(1, 12):
        ctx.run(); ctx.publish_async(bufs[k & 1])
        done = ctx.publish_wait()
        assert done is bufs[(k - 1) & 1]
        assert_soa_equal(done.trimmed(), want, store=st)
        for arr in done.arrays.values():                   # (the next delivery into these arrays must really write them)
            if arr is not None and arr.dtype != np.uint8:
                arr[...] = -7
    with pytest.raises(engine.CsvError):
        ctx.download()                                     # (a delivery is in flight)
    assert_soa_equal(ctx.publish_wait().trimmed(), want, store=st)
    # three in a row without waiting: the third is refused; slim results travel the same way
    slim = [ctx.result_buffers(no_support=True, coord32=True, fields=("cipos", "cilen", "dr", "gl_idx"), block=block) for _ in range(2)]
    ctx.run(); ctx.publish_async(slim[0]); ctx.run(); ctx.publish_async(slim[1])
    ctx.run()
    with pytest.raises(engine.CsvError):
        ctx.publish_async(a)
    for q in range(2):
        t = ctx.publish_wait().trimmed()
        for name in ("call_seg", "bp1", "bp2", "support", "cipos", "cilen", "dr", "gl_idx"):
            assert np.array_equal(t[name].astype(np.int64), want[name].astype(np.int64)), name
    # too small a result: the status of THAT delivery
    tiny = ctx.result_buffers(cap_calls=4, cap_support=4, block=block)
    ctx.run(); ctx.publish_async(tiny)
    with pytest.raises(engine.CsvError) as e:
        ctx.publish_wait()
    assert e.value.code == _abi.E_CAPACITY
    # pageable arrays cannot be written in place
    ctx.run()
    with pytest.raises(engine.CsvError):
        ctx.publish_async(ctx.result_buffers(pinned=False))
    assert_soa_equal(ctx.download().trimmed(), want, store=st)


def test_position_column_as_16_bit_gaps(ctx, monkeypatch):
    """CSV_IN_SIG_DELTA16 (ABI v8): the position column crosses the link as 16-bit gaps + the caller's escape list and is rebuilt
    on the device (k_unpack_a16) - bit for bit what the column itself gives: every golden-shaped workload, gaps that do not fit
    (> 65534, negative across segments, equal neighbours), 1 400 small segments (several anchors per tile), a subset of
    non-adjacent segments, segments that begin inside a source segment, one-shot (gate-first and bulk), resident, and with the
    reads table; a batch whose segments overlap falls back to the column itself."""
    monkeypatch.setenv("CSV_DELTA16_MIN", "0")
    monkeypatch.setenv("CSV_DELTA16_ESC", "0")           # (small test stores are sparse: take the gap form whatever the share of escapes)
    rng = np.random.default_rng(8)

    def both(pst, hb, st_plain, p, tasks=None):
        assert hb.c.flags & _abi.IN_SIG_DELTA16 and hb.a_delta is not None
        want = _oracle().cluster_batch(st_plain.host_batch(tasks or st_plain.tasks(), p), per_sig=True).trimmed()
        assert hb.rows8 is not None and hb.c.rows8
        for lazy_min in ("0", "1000000000"):
            monkeypatch.setenv("CSV_LAZY_MIN", lazy_min)
            for no_rows8 in (None, "1"):                  # the fetch out of the interleaved {b, read_id} array, or out of the two columns
                if no_rows8:
                    monkeypatch.setenv("CSV_NO_ROWS8", no_rows8)
                got = ctx.cluster_batch(hb, per_sig=True, reuse=True).trimmed()
                assert ctx.delta16_info() and ctx.lazy_info()[0] == (lazy_min == "0")
                assert_soa_equal(got, want, st_plain)
                monkeypatch.delenv("CSV_NO_ROWS8", raising=False)
        monkeypatch.delenv("CSV_LAZY_MIN")
        ctx.upload(hb, per_sig=True)
        assert ctx.delta16_info()
        ctx.run(); ctx.run()
        assert_soa_equal(ctx.download(per_sig=True).trimmed(), want, st_plain)
        monkeypatch.setenv("CSV_NO_DELTA16", "1")
        got = ctx.cluster_batch(hb, per_sig=True, reuse=True).trimmed()
        assert not ctx.delta16_info()
        assert_soa_equal(got, want, st_plain)
        monkeypatch.delenv("CSV_NO_DELTA16")
        return want

    for seed, kw, p in ((61, dict(), Params.ont(genotype=True)), (62, dict(n_sites=80, coverage=40, n_noise=3000), Params.hifi(genotype=True, min_support=3)),
                        (63, dict(n_sites=10, coverage=300, n_noise=50), Params.ont(min_support=3))):
        st = synth.small_mixed(seed=seed, **kw)
        pst = st.pinned()
        both(pst, pst.host_batch(pst.tasks(), p), st, p)
        sub = [t for i, t in enumerate(st.tasks()) if i % 3 != 1]          # non-adjacent segments
        both(pst, pst.host_batch(sub, p), st, p, tasks=sub)
    # gaps that do not fit, equal neighbours, positions near 2^31
    n = 6000
    pos = np.sort(np.concatenate([rng.integers(0, 2_000_000_000, 40), np.repeat(rng.integers(0, 2_000_000_000, 300), rng.integers(1, 40, 300))]))[:n]
    per = {"DEL": [(int(x), int(50 + (i % 7)), "r%d" % (i % 977), "DEL", "1") for i, x in enumerate(pos)],
           "INS": [(int(x) + 3, int(60 + (i % 5)), "q%d" % (i % 911), "A" * int(60 + (i % 5)), "INS", "1") for i, x in enumerate(pos)]}
    st = SigStore.from_tuple_lists(per)
    pst = st.pinned()
    assert len(pst.narrow["a_delta"][1]) > 30
    both(pst, pst.host_batch(pst.tasks(), Params.ont(min_support=3)), st, Params.ont(min_support=3))
    # thousands of small segments: many anchors per chain tile
    st = synth.small_mixed(seed=64, n_sites=30, coverage=20, n_contigs=24, contig_len=60_000, n_noise=2000)
    pst = st.pinned()
    p = Params.ont(min_support=3)
    hb = pst.host_batch(pst.tasks(), p)
    pieces = []
    for sg in hb.segments:                                               # every segment cut into pieces of ~40 rows at gaps wider than the bias
        b0, e0 = int(sg["sig_begin"]), int(sg["sig_end"])
        cuts = [b0]
        for i in range(b0 + 1, e0):
            if i - cuts[-1] >= 12 and sg["svtype"] in (_abi.DEL, _abi.INS, _abi.DUP) and st.a[i] - st.a[i - 1] > sg["max_cluster_bias"]:
                cuts.append(i)
        cuts.append(e0)
        for x, y in zip(cuts[:-1], cuts[1:]):
            q = sg.copy(); q["sig_begin"], q["sig_end"] = x, y
            pieces.append(q)
    segs = np.array(pieces, dtype=_abi.SEGMENT_DTYPE)
    assert len(segs) > 2 * len(hb.segments)
    nw = pst.narrow
    hb2 = _abi.HostBatch(segs, nw["a"], nw["b"], pst.read_id, pst.aux, n_chrom=len(pst.chroms), a_delta=nw["a_delta"])
    plain = _abi.HostBatch(segs, st.a, st.b, st.read_id, st.aux, n_chrom=len(st.chroms))
    want = _oracle().cluster_batch(plain, per_sig=True).trimmed()
    got = ctx.cluster_batch(hb2, per_sig=True, reuse=True).trimmed()
    assert ctx.delta16_info()
    assert_soa_equal(got, want, st)
    # overlapping segments: the library takes the column itself
    segs2 = np.concatenate([segs[:5], segs[:5]])
    hb3 = _abi.HostBatch(segs2, nw["a"], nw["b"], pst.read_id, pst.aux, n_chrom=len(pst.chroms), a_delta=nw["a_delta"])
    got = ctx.cluster_batch(hb3, per_sig=True, reuse=True).trimmed()
    assert not ctx.delta16_info()
    assert_soa_equal(got, _oracle().cluster_batch(_abi.HostBatch(segs2, st.a, st.b, st.read_id, st.aux, n_chrom=len(st.chroms)), per_sig=True).trimmed(), st)


@pytest.mark.parametrize("order", ["sorted", "extraction", "shuffled"])
def test_reads_table_as_16_bit_gaps_and_lengths(ctx, monkeypatch, order):
    """CSV_IN_READS_DELTA16 (ABI v8): the reads table's start column as 16-bit gaps (+ the escape rows: the first row of every sorted
    run, any gap that does not fit) and its end column as 16-bit lengths (+ the reads of 64 kb and more), rebuilt on the device
    before the reads are ordered - bit for bit what the columns themselves give, whatever the order of the table (a shuffled block
    is all escapes), with long reads, one-shot and resident, next to the signature column's own gap form"""
    monkeypatch.setenv("CSV_DELTA16_MIN", "0")
    monkeypatch.setenv("CSV_DELTA16_ESC", "0")
    monkeypatch.setenv("CSV_READS_GAP", "30000")
    import dataclasses
    rng = np.random.default_rng(17)
    for seed, kw, p in ((71, dict(n_sites=40, coverage=30), Params.ont(genotype=True, min_support=3)),
                        (72, dict(n_sites=25, coverage=60, contig_len=2_000_000, n_contigs=4, n_noise=500), Params.hifi(genotype=True, min_support=3)),
                        (73, dict(n_sites=30, coverage=25, n_contigs=5), Params(genotype=True, min_support=2, max_cluster_bias_INV=2000))):
        st = synth.small_mixed(seed=seed, genotype=True, **kw)
        # a share of very long reads: lengths that do not fit 16 bits
        long_ = rng.random(st.n_reads) < 0.07
        r_end = st.r_end.copy()
        r_end[long_] = st.r_start[long_] + rng.integers(65_535, 900_000, int(long_.sum()))
        st = dataclasses.replace(st, r_end=r_end)
        if order == "extraction":
            st, _ = synth.extraction_order(st, seed=seed, region=150_000, workers=5)
        elif order == "shuffled":
            perm = np.arange(st.n_reads)
            for c in range(len(st.chroms)):
                lo, hi = int(st.reads_off[c]), int(st.reads_off[c + 1])
                perm[lo:hi] = lo + rng.permutation(hi - lo)
            st = dataclasses.replace(st, r_start=st.r_start[perm], r_end=st.r_end[perm], r_primary=st.r_primary[perm], r_id=st.r_id[perm])
        pst = st.pinned()
        hb = pst.host_batch(pst.tasks(), p)
        assert hb.r_delta is not None and hb.r_len16 is not None and hb.c.flags & _abi.IN_READS_DELTA16
        assert len(hb.r_len16[1]) > 0                                        # some ends travel as escapes
        want = _oracle().cluster_batch(st.host_batch(st.tasks(), p), per_sig=True).trimmed()
        assert hb.r_idp is not None and hb.c.r_idp                       # ... and the id | primary << 31 word instead of two columns
        got = ctx.cluster_batch(hb, per_sig=True, reuse=True).trimmed()
        assert ctx.reads_delta_info() == 7
        assert_soa_equal(got, want, st)
        assert (got["dr"] > 0).any()
        ctx.upload(hb, per_sig=True)
        assert ctx.reads_delta_info() == 7
        ctx.run(); ctx.run()
        assert_soa_equal(ctx.download(per_sig=True).trimmed(), want, st)
        # only one of the two columns in its 16-bit form; neither
        for drop in ("r_delta", "r_len16"):
            hb1 = _abi.HostBatch(hb.segments, hb.a, hb.b, hb.read_id, hb.aux, n_chrom=hb.n_chrom, reads_off=hb.reads_off, r_start=hb.r_start, r_end=hb.r_end,
                                 r_primary=hb.r_primary, r_id=hb.r_id, contig_len=hb.contig_len,
                                 r_delta=None if drop == "r_delta" else hb.r_delta, r_len16=None if drop == "r_len16" else hb.r_len16)
            assert_soa_equal(ctx.cluster_batch(hb1, per_sig=True, reuse=True).trimmed(), want, st)
            assert ctx.reads_delta_info() == (2 if drop == "r_delta" else 1)
        monkeypatch.setenv("CSV_NO_DELTA16", "1")
        assert_soa_equal(ctx.cluster_batch(hb, per_sig=True, reuse=True).trimmed(), want, st)
        assert ctx.reads_delta_info() == 0
        monkeypatch.delenv("CSV_NO_DELTA16")
