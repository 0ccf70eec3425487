"""CPU tests of the host side: C-ABI surface, column store round trips, sharding, drop-in shims' argument handling."""
import ctypes as C
import json
import os
import pickle
import re
import subprocess
import sys

import numpy as np
import pytest

from cutesv_amd import synth, shard, _abi, _lib
from cutesv_amd.columns import SigStore, Params, TYPES
from helpers import load_json, store_from_json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "cutesv_hip.h")).read()
    declared = set(re.findall(r"\b(csv_[a-z_0-9]+)\s*\(", header))
    declared -= {"csv_ctx"}
    assert {"csv_cluster_batch", "csv_batch_upload", "csv_batch_run", "csv_batch_download", "csv_ctx_create",
            "csv_ctx_destroy", "csv_last_error", "csv_gl_index", "csv_abi_version"} <= declared
    L = _lib.lib()                                      # raises if the .so is missing or a symbol is absent
    for name in declared:
        assert hasattr(L, name), name
    assert {n for n, _, _ in _lib.SYMBOLS} == declared
    assert L.csv_abi_version() == _abi.ABI_VERSION
    # the pure host helper needs no GPU
    from cutesv_amd import genotype
    for c0, c1 in ((3, 1), (6, 2), (0, 5), (17, 9), (250, 31), (0, 400), (99, 2)):
        assert L.csv_gl_index(c0, c1) == genotype.gl_index(c0, c1)
    # the ctypes / numpy mirrors have exactly the layouts the library was compiled with
    from cutesv_amd import rebuild, vcf, rows, extract
    mirrors = [_abi.SEGMENT_DTYPE.itemsize, C.sizeof(_abi.BatchIn), C.sizeof(_abi.BatchOut), C.sizeof(_abi.RunStats),
               C.sizeof(rebuild.RebuildIn), C.sizeof(rebuild.RebuildOut), C.sizeof(vcf.VcfIn), C.sizeof(rows.RowsIn),
               C.sizeof(extract.CigarIn), C.sizeof(extract.CigarOut), C.sizeof(extract.SplitIn), C.sizeof(extract.SplitOut)]
    assert [L.csv_struct_size(i) for i in range(12)] == mirrors
    assert L.csv_struct_size(12) == -1


def test_missing_extension_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.ExtensionMissing):
        _lib.lib()


def test_store_roundtrips(tmp_path):
    case = load_json("small_cases.json.gz")[-1]          # real read names, real sequences
    st = store_from_json(case["store"])
    per, reads = st.tuple_lists()
    st2 = SigStore.from_tuple_lists(per, reads)
    for k in ("a", "b", "read_id", "aux", "reads_off", "r_start", "r_end", "r_primary", "r_id"):
        assert np.array_equal(getattr(st, k), getattr(st2, k)), k
    assert st.seg_index == st2.seg_index and st.names.names == st2.names.names
    st.save(str(tmp_path / "cols"))
    st3 = SigStore.load(str(tmp_path / "cols"))
    assert np.array_equal(st3.a, st.a) and st3.seg_index == st.seg_index and st3.sequence(st.seg_index[("INS", "chr1")][0]) == \
        st.sequence(st.seg_index[("INS", "chr1")][0])
    # unsorted input with duplicates is sorted / de-duplicated like the reference's rebuild step
    rng = np.random.default_rng(0)
    shuffled = {t: [per[t][i] for i in rng.permutation(len(per[t]))] + per[t][:5] for t in TYPES}
    st4 = SigStore.from_tuple_lists(shuffled, reads)
    assert np.array_equal(st4.a, st.a) and np.array_equal(st4.read_id, st.read_id)


def test_reference_workdir_reader(tmp_path):
    st = synth.small_mixed(seed=3, n_sites=6, n_noise=50, n_loci=5)
    per, reads = st.tuple_lists()
    index = {}
    d = str(tmp_path) + "/"
    for t in TYPES:
        index[t] = {}
        with open(d + t + ".pickle", "wb") as f:
            for ch in st.chroms:
                blk = [x for x in per[t] if x[-1] == ch]
                if blk:
                    index[t][ch] = f.tell()
                    pickle.dump(blk, f)
    index["reads"] = {}
    with open(d + "reads.pickle", "wb") as f:
        for ch in st.chroms:
            blk = [r for r in reads if r[-1] == ch]
            if blk:
                index["reads"][ch] = f.tell()
                pickle.dump(blk, f)
    with open(d + "sigindex.pickle", "wb") as f:
        pickle.dump(index, f)
    st2 = SigStore.from_reference_workdir(d)
    assert np.array_equal(st2.a, st.a) and np.array_equal(st2.b, st.b)
    for (t, ch), (b, e) in st.seg_index.items():         # aux is meaningful for INS / INV / TRA only
        assert st2.seg_index[(t, ch)] == (b, e)
        if t in ("INS", "INV", "TRA"):
            assert np.array_equal(st2.aux[b:e], st.aux[b:e]), (t, ch)
    assert [st2.names[i] for i in st2.read_id] == [st.names[i] for i in st.read_id]


def test_lpt_sharding_is_a_balanced_partition():
    st = synth.ont30(scale=0.02)
    for n in (1, 2, 4, 8):
        parts = shard.assign(st, n)
        flat = [c for p in parts for c in p]
        assert sorted(flat) == sorted({c for (_, c) in st.seg_index}) and len(flat) == len(set(flat))
        loads = [sum(shard.chromosome_cost(st, c) for c in p) for p in parts]
        assert max(loads) <= sum(loads) / n + max(shard.chromosome_cost(st, c) for c in flat)
        tasks = [t for r in range(n) for t in shard.tasks_of_rank(st, r, n)]
        assert sorted(tasks) == sorted(st.tasks())
    # chromosomes cut at gaps wider than max_cluster_bias: every signature in exactly one unit, and the
    # heaviest of 8 ranks within 3 % of the mean although chr1 alone is two thirds of a rank's share
    from cutesv_amd.columns import Params
    st = synth.ont30(scale=0.2)
    p = Params.ont()
    for n in (2, 4, 8):
        loads, seen = [], 0
        for units in shard.plan(st, n, p):
            segs, keys, _ = shard.rank_batch(st, p, units)
            assert len(set(keys)) == len(keys)
            loads.append(int((segs["sig_end"] - segs["sig_begin"]).sum()))
        assert sum(loads) == st.n_sig
        assert max(loads) <= 1.03 * sum(loads) / n, (n, loads)


def test_inv_dominated_chromosome_is_cut_per_strand_run():
    """INV rows are ordered (strand, pos) (main script :792): a coordinate cut is a sub-range of EVERY strand's run, the gap
    test holds at the exact index used, and the pieces' rows go back strand-major.  A store whose densest cuttable segment is
    INV, worlds 2 .. 8 with forced cuts, through the oracle engine: merged rows == unsharded rows, loads balanced."""
    import dataclasses
    from cutesv_amd.columns import Params, SigStore, NameTable
    from cutesv_amd import rows as rows_mod
    from oracle import oracle
    rng = np.random.default_rng(5)
    n_sites, per = 300, 15
    site = np.sort(rng.integers(10_000, 40_000_000, n_sites))
    a, b, rid, aux = [], [], [], []
    for strand in (0, 1):                                        # '++' rows first, then '--' (the rebuild's order)
        pos = np.repeat(site, per) + rng.integers(-40, 40, n_sites * per)
        o = np.argsort(pos, kind="stable")
        a.append(pos[o]); b.append((pos + 5000 + rng.integers(-40, 40, len(pos)))[o])
        rid.append((np.repeat(np.arange(n_sites), per) * per + np.tile(np.arange(per), n_sites))[o] + strand * 1_000_000)
        aux.append(np.full(len(pos), strand))
    n_inv = 2 * n_sites * per
    d_pos = np.sort(rng.integers(10_000, 40_000_000, 400))       # a sparse DEL segment on the same chromosome
    st = SigStore(chroms=["1"], a=np.concatenate([d_pos] + a).astype(np.int64), b=np.concatenate([np.full(400, 50)] + b).astype(np.int64),
                  read_id=np.concatenate([np.arange(400) + 5_000_000] + rid).astype(np.int32),
                  aux=np.concatenate([np.zeros(400)] + aux).astype(np.int32),
                  seg_index={("DEL", "1"): (0, 400), ("INV", "1"): (400, 400 + n_inv)}, names=NameTable(), strands=("++", "--"))
    p = Params.ont(min_support=5)

    def stage(units):
        hb, keys = shard.host_batch(st, p, units)
        per_seg = rows_mod.rows_by_segment(st, hb.segments, oracle.cluster_batch(hb))
        return {k: per_seg[i] for i, k in enumerate(keys)}
    full = shard.merge_rows([stage(shard.plan(st, 1, p)[0])])
    assert len(full["1"]) > 500
    for world in (2, 3, 8):
        plan = shard.plan(st, world, p, max_imbalance=0.0)
        assert any(u[2] > 1 for us in plan for u in us)
        loads = []
        parts = []
        for units in plan:
            segs, keys, _ = shard.rank_batch(st, p, units)
            loads.append(int((segs["sig_end"] - segs["sig_begin"]).sum()))
            parts.append(stage(units))
        assert sum(loads) == st.n_sig
        assert max(loads) <= 1.25 * st.n_sig / world, (world, loads)
        assert shard.merge_rows(parts) == full, world


def test_tra_window_status_skips_records_without_an_end():
    """fetch() also yields records that have no reference_end (an unmapped mate placed at its partner's position, flag 69);
    count_coverage never looks at their coordinates (`flag not in (0, 16)`, cuteSV_genotype.py:72-93) and neither may the
    chunked form: they only count towards the iteration total"""
    import types
    from cutesv_amd.tra_bam import window_status
    rec = lambda flag, s, e, q: types.SimpleNamespace(flag=flag, reference_start=s, reference_end=e, query_name=q)   # noqa: E731
    al = [rec(0, 100, 900, "a"), rec(69, 150, None, "u1"), rec(16, 120, 950, "b"), rec(256, 0, 2000, "sec"), rec(69, None, None, "u2"), rec(0, 90, 1000, "c")]
    names = set()
    assert window_status(al, 200, 800, names, up_bound=10, itround=500) == 0 and names == {"a", "b", "c"}
    names = set()
    assert window_status(al, 200, 800, names, up_bound=2, itround=500) == 1 and names == {"a", "b"}
    names = set()
    assert window_status(al, 200, 800, names, up_bound=10, itround=3, chunk=2) == -1 and names == {"a", "b"}     # 2 of 3 primary > 0.2


def test_shims_follow_the_reference_argument_contract():
    from cutesv_amd import resolve
    idx = {"DEL": {}, "INS": {}, "INV": {}, "DUP": {}, "TRA": {}}
    # chromosome absent from the index -> (chr, []) without touching any file or GPU (INDEL:44-45 etc.)
    assert resolve.run_del(("/nonexistent/", "7", "DEL", 10, 0.5, 200, 5, "bam", False, 500, 1.0, idx)) == ("7", [])
    assert resolve.run_ins(("/nonexistent/", "7", "INS", 10, 0.3, 100, 5, "bam", False, 500, 1.0, idx)) == ("7", [])
    assert resolve.run_inv(("/nonexistent/", "7", "INV", 10, 500, 30, "bam", False, 100000, 500, idx)) == ("7", [])
    assert resolve.run_dup(("/nonexistent/", "7", 10, 500, 30, "bam", False, 100000, 500, idx)) == ("7", [])
    assert resolve.run_tra(("/nonexistent/", "7", 10, 0.6, 50, "bam", False, 500, idx)) == ("7", [])
    assert resolve.run_tra(("/nonexistent/", "7", 10, 0.6, 50, "bam", True, 500, idx)) == ("7", [])


def test_bam_header_reference_lengths(tmp_path):
    # a BAM header is: magic, l_text, text, n_ref, then (l_name, name NUL, l_ref) per reference, in gzip members
    import gzip, struct
    from cutesv_amd.bam_header import reference_lengths
    refs = [("chr1", 248956422), ("chrUn_KI270442v1", 392061), ("MT", 16569)]
    text = b"@HD\tVN:1.6\tSO:coordinate\n"
    raw = b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(refs))
    for n, l in refs:
        raw += struct.pack("<i", len(n) + 1) + n.encode() + b"\0" + struct.pack("<i", l)
    p = tmp_path / "x.bam"
    with open(p, "wb") as f:                      # two gzip members, like consecutive BGZF blocks
        f.write(gzip.compress(raw[:30])); f.write(gzip.compress(raw[30:] + b"alignment records follow"))
    assert reference_lengths(str(p)) == dict(refs)
    (tmp_path / "y.bam").write_bytes(gzip.compress(b"CRAM...."))
    with pytest.raises(ValueError):
        reference_lengths(str(tmp_path / "y.bam"))


def test_two_rank_gloo_sharded_stage_matches_single_rank():
    """N > 1 plumbing on CPU: two gloo ranks take their LPT shard, cluster it (oracle engine stands in for
    the GPU here), exchange per-chromosome digests; rank 0 checks the union equals the unsharded run."""
    script = os.path.join(ROOT, "tests", "dist_shard_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARD-OK" in outs[0]


class _StubBam:
    """pysam.AlignmentFile stand-in over a store's reads table (the same stand-in the golden generator used)"""
    store = None

    def __init__(self, path):
        self.st = _StubBam.store

    def get_reference_length(self, chrom):
        return int(self.st.contig_len[self.st.chroms.index(chrom)])

    def fetch(self, chrom, s, e):
        import types
        st = self.st
        c = st.chroms.index(chrom)
        for i in range(int(st.reads_off[c]), int(st.reads_off[c + 1])):
            if st.r_start[i] >= e:
                break
            if st.r_end[i] > s:
                yield types.SimpleNamespace(flag=0 if st.r_primary[i] == 1 else 2048, reference_start=int(st.r_start[i]),
                                            reference_end=int(st.r_end[i]), query_name=st.names[st.r_id[i]])

    def close(self):
        pass


def test_bam_faithful_tra_genotyping_matches_reference(monkeypatch):
    """tra_bam.genotype_rows (the default of the run_tra drop-in under --genotype) reproduces the genotype fields the
    reference's call_gt computed for the same calls (tra_genotype.json.gz; pysam stubbed over the reads table)"""
    import sys
    import types
    from helpers import load_json, store_from_json
    from cutesv_amd import tra_bam
    stub = types.ModuleType("pysam")
    stub.AlignmentFile = _StubBam
    monkeypatch.setitem(sys.modules, "pysam", stub)
    n = 0
    for case in load_json("tra_genotype.json.gz"):
        st = store_from_json(case["store"])
        _StubBam.store = st
        p = case["params"]
        for t, c, rows in case["rows"]:
            if t != "TRA" or not rows:
                continue
            bare = [r[:6] + [".", "./.", ".,.,.", ".", ".", r[11]] for r in rows]
            got = tra_bam.genotype_rows(bare, "stub.bam", p["max_cluster_bias_TRA"], p["gt_round"])
            assert got == rows, (case["name"], c)
            n += len(rows)
    assert n > 20


def test_legacy_sigs_text_files_round_trip(tmp_path):
    """--write_old_sigs (main script :766-816): the text lines the reference prints per type, read back into the flat store"""
    from cutesv_amd.columns import SigStore
    st = synth.small_mixed(seed=21, genotype=True)
    per, reads = st.tuple_lists()
    fmt = {"DEL": lambda e: "%s\t%s\t%d\t%d\t%s\n" % (e[-2], e[-1], e[0], e[1], e[2]),
           "INS": lambda e: "%s\t%s\t%d\t%d\t%s\t%s\n" % (e[-2], e[-1], e[0], e[1], e[2], e[3]),
           "DUP": lambda e: "%s\t%s\t%d\t%d\t%s\n" % (e[-2], e[-1], e[0], e[1], e[2]),
           "INV": lambda e: "%s\t%s\t%s\t%d\t%d\t%s\n" % (e[-2], e[-1], e[0], e[1], e[2], e[3]),
           "TRA": lambda e: "%s\t%s\t%s\t%d\t%s\t%d\t%s\n" % (e[-2], e[-1], e[0], e[1], e[2], e[3], e[4])}
    for t, lst in per.items():
        with open(tmp_path / (t + ".sigs"), "w") as f:
            f.writelines(fmt[t](e) for e in lst)
    with open(tmp_path / "reads.sigs", "w") as f:
        f.writelines("%s\t%d\t%d\t%d\t%s\n" % (e[-1], e[0], e[1], e[2], e[3]) for e in reads)
    got = SigStore.from_sigs_dir(str(tmp_path))
    want = SigStore.from_tuple_lists(per, reads)
    assert got.seg_index == want.seg_index and got.chroms == want.chroms
    for k in ("a", "b", "read_id", "aux", "reads_off", "r_start", "r_end", "r_primary", "r_id"):
        assert np.array_equal(getattr(got, k), getattr(want, k)), k
    assert got.ins_seq == want.ins_seq and got.names.names == want.names.names


def test_vcf_text_is_the_same_on_one_thread_and_on_a_team(monkeypatch):
    """csv_vcf_emit formats slices of 1024 calls on a team of worker threads that outlives the call (asleep in between): the
    text and the SVID counters of a few thousand calls do not depend on the number of threads, nor on the team having been
    used, resized and used again (no GPU needed: the calls come from the oracle)"""
    from cutesv_amd import synth, vcf
    from oracle import oracle
    st = synth.ont30(seed=5, scale=0.12)
    p = Params.ont()
    hb = st.host_batch(st.tasks(), p)
    res = oracle.cluster_batch(hb, per_sig=False)
    assert res.n_calls > 2 * 1024

    def emit(threads):
        monkeypatch.setenv("CSV_VCF_THREADS", str(threads))
        out = vcf.emit_records(st, hb.segments, res, {}, min_size=p.min_size, max_size=p.max_size, genotype=False, ignore_sequence=True)
        return (out[0], list(out[1])) if isinstance(out, tuple) else out
    want = emit(1)
    for threads in (4, 4, 7, 2, 4):
        assert emit(threads) == want


def test_clip_join_is_the_python_slice_and_join():
    """_cols_native.clip_join (the ALT strings of INS calls for csv_vcf_in.ins_alt) == b"".join(seq[pick][:SVLEN].encode()),
    for the three shapes a store keeps its inserted sequences in (cuteSV_genotype.py:297-309)."""
    from cutesv_amd import _cols_native as cn
    rng = np.random.default_rng(5)
    table = ["".join("ACGTN"[k] for k in rng.integers(0, 5, int(m))) for m in rng.integers(0, 90, 300)]
    table[7] = "ACéGT中NN"                                  # not ASCII: sliced by code point
    picks = rng.integers(0, len(table), 1000).astype(np.int64)
    picks[:3] = 7
    lens = rng.integers(-5, 120, 1000).astype(np.int64)
    lens[:3] = (3, 6, 100)
    want = [table[p][:k].encode() for p, k in zip(picks.tolist(), lens.tolist())]
    for tab in (table, tuple(table), dict(enumerate(table))):
        took = np.full(1000, -1, np.int64)
        blob = cn.clip_join(tab, picks, lens, took)
        assert blob == b"".join(want) and took.tolist() == [len(w) for w in want]
    took = np.empty(1000, np.int64)
    blob = cn.clip_join(None, picks, lens, took)
    want = [("ACGT" * 40)[:max(0, k)].encode() for k in lens.tolist()]
    assert blob == b"".join(want) and took.tolist() == [len(w) for w in want]
    with pytest.raises(IndexError):
        cn.clip_join(table, np.array([len(table)], np.int64), np.array([1], np.int64), np.empty(1, np.int64))
    with pytest.raises(KeyError):
        cn.clip_join({1: "A"}, np.array([2], np.int64), np.array([1], np.int64), np.empty(1, np.int64))
    assert cn.clip_join(table, np.zeros(0, np.int64), np.zeros(0, np.int64), np.empty(0, np.int64)) == b""


def test_bench_reads_its_committed_counter_files():
    """bench.py attaches the PMC traffic and the VALU-issue figures of the dominant kernel from files committed under profiles/
    (collected by scripts/refresh_profiles.sh / profile_insts.sh on the same command): they must parse and make sense."""
    sys.path.insert(0, ROOT)
    import bench
    tr = bench.traffic_of("cfg3", 1.0)
    assert tr and tr["k_refine_indel_wave"] > 8_000_000 and all("." not in k and "[" not in k for k in tr)
    vi = bench.valu_issue_of("cfg3", 1.0, "k_refine_indel_wave", 13.6)
    assert vi and 0.3 < vi["valu_busy_frac"] < 1.05 and vi["valu_insts_per_launch"] > 1e6
    assert 0.05 < vi["wave_share_issuing_valu"] < 0.5
    assert bench.valu_issue_of("cfg3", 0.5, "k_refine_indel_wave", 13.6) is None        # (counters are of the full-size workload only)
    assert bench.valu_issue_of("cfg3", 1.0, "k_no_such_kernel", 1.0) is None
    # the one JSON line stays inside the 8 000-byte tail of stdout the driver keeps: the full objects of a run are compacted
    # to named numbers (every key of the contract survives), the detail goes to stderr / gpurun_out
    with open(os.path.join(ROOT, "profiles", "r06_bench_cfg3.json")) as f:
        full = json.loads(f.read().strip().splitlines()[-1])          # (`bench.py --full`: every object on the line)
    c = bench.compact(dict(full))
    assert c["host_to_host"]["ms"] == pytest.approx(full["host_to_host"]["ms"], rel=1e-3) and c["roofline"]["pcie"]["bytes_up_bulk"] > 0
    assert c["regions_ms"]["stage_wall_lazy_rows"] < c["regions_ms"]["stage_wall_rows"]
    assert len(json.dumps(c)) + 1 <= bench.LINE_LIMIT
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert c[k] == full[k] or k == "config", k
    assert c["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3) and c["roofline"]["bound"] == "hbm" and c["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert c["cpu_baseline"]["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-3) and c["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"]
    assert set(c["other_workloads"]) == set(full["other_workloads"]) and c["parity_vs_oracle"] is True
    # round 6: the stage under cuteSV's forked pool for cfg3 and cfg4, what the pool baseline is made of, the name of value's region
    assert c["value_region"] == "resident_delivered_pipelined"
    for wl in ("cfg3", "cfg4"):
        legs = c["mode1_stage"][wl]["legs"]
        assert [leg["workers"] for leg in legs] == [1, 8, 32]
        assert all(leg["rows_equal_reference_model"] and leg["wall_ms"] > 0 and leg["vs_reference_pool"] > 1 and leg["pool_alone_ms"] > 0 for leg in legs)
    assert all(c["cpu_baseline"][k] > 0 for k in ("wall_s", "wall_s_pool_fit", "pool_startup_s", "critical_path_s"))
    assert c["host_to_host"]["delta16"] is True and c["host_to_host"]["ms"] < 0.5
    r, w = os.pipe()
    saved = bench._OUT_FD
    try:
        bench._OUT_FD = w
        bench.emit(dict(full))
    finally:
        bench._OUT_FD = saved
        os.close(w)
    line = os.read(r, 1 << 20)
    os.close(r)
    d = json.loads(line)
    assert len(line) <= bench.LINE_LIMIT and d["value"] == full["value"] and d["roofline"]["kernel"] == full["roofline"]["kernel"]


def _same_store(x, y):
    assert x.chroms == y.chroms and x.seg_index == y.seg_index and tuple(x.strands) == tuple(y.strands)
    for k in ("a", "b", "read_id", "aux", "reads_off", "r_start", "r_end", "r_primary", "r_id"):
        u, v = getattr(x, k), getattr(y, k)
        assert (u is None) == (v is None), k
        if u is not None:
            assert np.array_equal(u, v), k
    assert list(x.names.names) == list(y.names.names)
    if isinstance(y.ins_seq, dict):
        assert not len(x.ins_seq) and not len(y.ins_seq)
    else:
        assert list(x.ins_seq) == list(y.ins_seq)


def test_task_pickles_walked_in_c_give_the_store_pickle_load_gives(tmp_path):
    """SigStore.from_task_pickles (the reference's <TYPE>.pickle / reads.pickle blocks walked in C out of the mapped files:
    INDEL:52-58, DUP:25-27, INV:42-44, TRA:36-38 read them with pickle.load) == from_task_lists(pickle.load(...)), for every
    type, every protocol pickle can have written, shared objects (memo references), x.5 positions, big and negative ints,
    booleans, non-ASCII text; and None for streams that need real unpickling."""
    import mmap
    from cutesv_amd.columns import SigStore, SpanList
    from cutesv_amd import _cols_native as cn
    rng = np.random.default_rng(11)
    names = ["read/%d/ccs" % i for i in range(300)] + ["ré中d%d" % i for i in range(5)]

    def lists(svtype, n):
        out = []
        chr1 = "chr1"                                         # ONE object: pickle writes memo references to it
        for i in range(n):
            nm = names[int(rng.integers(0, len(names)))]
            pos = int(rng.integers(0, 1 << 31)) + (0.5 if rng.random() < 0.2 else 0)
            if rng.random() < 0.05:
                pos = int(rng.integers(1 << 33, 1 << 40))     # LONG1
            ln = int(rng.integers(1, 70000))
            if svtype == "DEL":
                out.append((pos, ln, nm, "DEL", chr1))
            elif svtype == "INS":
                seq = "ACGT" * int(rng.integers(0, 80)) + ("é" if rng.random() < 0.02 else "")
                out.append((pos, ln, nm, seq, "INS", chr1))
            elif svtype == "DUP":
                out.append((int(pos), int(pos) + ln, nm, "DUP", chr1))
            elif svtype == "INV":
                out.append((("++", "--")[i & 1], int(pos), int(pos) + ln, nm, "INV", chr1))
            else:
                out.append(("ABCDE"[int(rng.integers(0, 5))], int(pos), "chr%d" % int(rng.integers(2, 6)), ln, nm, "TRA", chr1))
        return out

    reads = [(int(rng.integers(0, 1 << 28)), int(rng.integers(1 << 28, 1 << 29)), bool(i & 1) if i % 3 else int(i & 1),
              names[int(rng.integers(0, len(names)))], "chr1" if i % 4 else "chr2") for i in range(700)]
    # the walker says whether every string of a stream was ASCII (a string's len() is then its byte count and the store skips
    # the code-point pass over the sequences): both answers
    ascii_rows = [(10 + i, 50, "read%d" % (i % 7), "ACGT" * (i % 5), "INS", "chr1") for i in range(40)]
    for rows_, flag in ((ascii_rows, True), (ascii_rows[:20] + [(31, 50, "read1", "ACé", "INS", "chr1")] + ascii_rows[20:], False),
                        (ascii_rows[:20] + [(31, 50, "ré", "AC", "INS", "chr1")], False)):
        blob = pickle.dumps(rows_, protocol=4)
        t = cn.pickle_table(blob, 0, 6, (0, 1), (2, 3))
        assert t is not None and t[4] is flag
        got = SigStore.from_task_pickles("INS", "chr1", blob, 0)
        assert got.aux.tolist() == [len(r[3]) for r in rows_]
    for proto in (2, 3, 4, 5):
        for svtype in ("DEL", "INS", "DUP", "INV", "TRA"):
            for n in (0, 1, 3, 2500):                         # (APPEND form, one APPENDS batch, several batches and frames)
                sigs = lists(svtype, n)
                path, rpath = str(tmp_path / "s.pickle"), str(tmp_path / "r.pickle")
                with open(path, "wb") as f:
                    f.write(b"junk before the block")
                    off = f.tell()
                    pickle.dump(sigs, f, protocol=proto)
                    f.write(b"the next block")
                with open(rpath, "wb") as f:
                    pickle.dump(reads, f, protocol=proto)
                with open(path, "rb") as f, open(rpath, "rb") as g:
                    sm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                    rm = mmap.mmap(g.fileno(), 0, access=mmap.ACCESS_READ)
                with_reads = svtype != "TRA" and n > 1
                got = SigStore.from_task_pickles(svtype, "chr1", sm, off, rm if with_reads else None, 0 if with_reads else None)
                chroms = sorted({"chr1"} | {x[2] for x in sigs}) if svtype == "TRA" else None
                want = SigStore.from_task_lists(svtype, "chr1", sigs, reads if with_reads else [], chroms=chroms)
                assert got is not None, (proto, svtype, n)
                _same_store(got, want)
                if svtype == "INS" and n:
                    picks = rng.integers(0, n, 40).astype(np.int64)
                    clips = rng.integers(-3, 400, 40).astype(np.int64)
                    blob, took = got.ins_seq.join(picks, clips)
                    ref = [sigs[int(p)][3][:int(c)].encode() for p, c in zip(picks, clips)]
                    assert blob == b"".join(ref) and took.tolist() == [len(r) for r in ref]
                    assert got.ins_blob(picks=picks)[0] == want.ins_blob(picks=picks)[0]
                    assert np.array_equal(got.ins_blob(picks=picks)[1], want.ins_blob(picks=picks)[1])
                if n:
                    pk = rng.integers(0, len(got.names.names), 50)
                    assert got.names_blob(picks=pk)[0] == want.names_blob(picks=pk)[0]
    # streams the walker must hand back to pickle: a nested tuple, a list element, a shared tuple object, bytes, a wrong width
    t = (1, 2, "r", "DEL", "chr1")
    for bad in ([(1, 2, ("r",), "DEL", "chr1")], [[1, 2, "r", "DEL", "chr1"]], [t, t], [(1, 2, b"r", "DEL", "chr1")],
                [(1, 2, "r", "DEL")], [(None, 2, "r", "DEL", "chr1")], {"a": 1}, [(float("nan"), 2, "r", "DEL", "chr1")],
                [(1 << 70, 2, "r", "DEL", "chr1")]):
        assert cn.pickle_table(pickle.dumps(bad), 0, 5, (0, 1), (2,)) is None, bad
    # a truncated stream is never accepted: declined (its FRAME runs past the buffer: pickle.load then says what is wrong) or refused
    for proto in (2, 4):
        trunc = pickle.dumps([(i, 2, "r", "DEL", "chr1") for i in range(10)], protocol=proto)[:-9]
        try:
            got = cn.pickle_table(trunc, 0, 5, (0, 1), (2,))
        except ValueError:
            got = None
        assert got is None
    assert isinstance(SpanList(b"abcd", [0, 2], [2, 2])[-1], str) and SpanList(b"abcd", [0, 2], [2, 2])[0:2] == ["ab", "cd"]


def test_store_for_reads_a_reference_workdir_the_same_with_and_without_the_walker(tmp_path, monkeypatch):
    """resolve._store_for on the reference's file layout (helpers.write_reference_workdir): the C walk of the mapped pickles and
    the pickle.load path (CUTESV_AMD_UNPICKLE=1) build the same task stores, reads blocks included."""
    from cutesv_amd import resolve
    from helpers import write_reference_workdir
    for case in [c for c in load_json("small_cases.json.gz") if c["name"] in ("realnames_gt", "ont_gt", "hifi")]:
        st = store_from_json(case["store"])
        d = str(tmp_path / case["name"]) + "/"
        os.makedirs(d)
        idx = write_reference_workdir(st, d)
        for t in ("DEL", "INS", "INV", "DUP", "TRA"):
            for c in idx[t]:
                need_reads = t != "TRA"
                monkeypatch.delenv("CUTESV_AMD_UNPICKLE", raising=False)
                got = resolve._store_for(d, idx, t, c, need_reads)
                assert got.ins_seq is None or not isinstance(got.ins_seq, list)            # (the walker's spans, not pickle's objects)
                monkeypatch.setenv("CUTESV_AMD_UNPICKLE", "1")
                want = resolve._store_for(d, idx, t, c, need_reads)
                _same_store(got, want)


def test_random_workloads_sharded_over_2_to_8_ranks_merge_to_the_unsharded_rows():
    """random all-type workloads (genotyped or not, reads in extraction order or sorted, ONT / HiFi presets), cut into pieces for
    2, 3, 5 and 8 ranks with and without forced cuts, every rank's batch through the oracle engine: the merged rows are the
    unsharded rows (the N > 1 path the driver runs on hardware).  A 600-workload campaign of the same loop (43 k
    cut pieces) found nothing."""
    from cutesv_amd import rows as rows_mod
    from cutesv_amd.columns import Params
    from oracle import oracle
    cuts = 0
    for it in range(24):
        rng = np.random.default_rng(7000 + it)
        gt = bool(rng.integers(0, 2))
        st = synth.small_mixed(seed=9000 + it, n_sites=int(rng.integers(5, 400)), coverage=int(rng.choice([4, 9, 20, 45, 90])), genotype=gt,
                               n_contigs=int(rng.integers(2, 5)))
        if gt and rng.random() < 0.5:
            st, _ = synth.extraction_order(st, seed=it, region=int(rng.choice([100_000, 250_000])), workers=int(rng.integers(2, 9)))
        p = Params.ont(genotype=gt, min_support=int(rng.integers(2, 8))) if rng.random() < 0.5 else Params.hifi(genotype=gt, min_support=int(rng.integers(2, 5)))

        def stage(units):
            hb, keys = shard.host_batch(st, p, units)
            per_seg = rows_mod.rows_by_segment(st, hb.segments, oracle.cluster_batch(hb))
            return {k: per_seg[i] for i, k in enumerate(keys)}
        full = shard.merge_rows([stage(shard.plan(st, 1, p, genotype=gt)[0])])
        for world in (2, 3, 5, 8):
            plan = shard.plan(st, world, p, genotype=gt, max_imbalance=float(rng.choice([0.0, 0.03])))
            cuts += sum(1 for us in plan for u in us if u[2] > 1)
            assert shard.merge_rows([stage(u) for u in plan]) == full, (it, world)
    assert cuts > 500


def test_pickle_walker_accepts_nothing_that_pickle_rejects():
    """differential fuzz of _cols_native.pickle_table against pickle.loads: single-byte corruptions of a task's pickled list in
    protocols 2 / 4 / 5.  The walker may decline (None: resolve._store_for then lets pickle.load read the block), it must never
    ACCEPT a stream that pickle itself rejects - a FRAME that runs past the buffer, a string payload that is not UTF-8 (r04: the
    walker ignored frame lengths and decoded strings late or never)."""
    import random
    import signal
    from cutesv_amd import _cols_native as cn
    rng = random.Random(11)
    rows = [(1000 + i * 7, 40 + i % 13, ("read_%d_é" % i) if i % 50 == 0 else "read_%d" % i, "ACGT" * (i % 9 + 1), "INS", "1") for i in range(200)]

    def on_alarm(*_):
        raise TimeoutError()
    old = signal.signal(signal.SIGALRM, on_alarm)
    both = walker_only = 0
    try:
        for proto in (2, 4, 5):
            blob = pickle.dumps(rows, protocol=proto)
            for _ in range(1500):
                b = bytearray(blob)
                b[rng.randrange(len(b))] = rng.randrange(256)
                b = bytes(b)
                try:
                    ok_w = cn.pickle_table(b, 0, 6, (0, 1), (2, 3, 4, 5)) is not None
                except ValueError:
                    ok_w = False
                if not ok_w:
                    continue                                   # (declined: nothing to compare)
                signal.alarm(5)
                try:
                    ok_p = isinstance(pickle.loads(b), list)
                except BaseException:                          # noqa: BLE001  (whatever pickle raises on garbage, a time-out included)
                    ok_p = False
                finally:
                    signal.alarm(0)
                both += ok_p
                walker_only += not ok_p
                if ok_p:                                       # accepted by both: the SAME table (the row-at-a-time path of the walker
                    want = pickle.loads(b)                     # hands anything unusual back to its opcode loop - and must then agree)
                    t = cn.pickle_table(b, 0, 6, (0, 1), (2, 3, 4, 5))
                    assert t[0] == len(want)
                    assert np.frombuffer(t[2][0], np.int64).tolist() == [int(r[0]) for r in want]
                    assert np.frombuffer(t[2][1], np.int64).tolist() == [int(r[1]) for r in want]
                    for k, (o, l) in zip((2, 3, 4, 5), t[3]):
                        o, l = np.frombuffer(o, np.int64), np.frombuffer(l, np.int32)
                        assert [b[int(x):int(x) + int(y)].decode("utf-8", "surrogatepass") for x, y in zip(o, l)] == [r[k] for r in want]
    finally:
        signal.signal(signal.SIGALRM, old)
    assert walker_only == 0 and both > 500, (walker_only, both)


def test_pickle_walker_row_at_a_time_path_and_its_edges():
    """pickle_table reads a row - MARK, fields, TUPLE - in one go and hands anything else back to its opcode loop: the table must
    not depend on which of the two read a row.  Streams built by hand around the hand-over points: the last rows of a buffer (the
    fast path wants 16 bytes of slack), a FRAME between two fields, BINPUT / LONG_BINPUT memo numbering (protocol 2), a name that
    is a memo reference, LONG1 / BININT / BINFLOAT fields, a row of the wrong width in the middle, and every value of the end
    hint (none, exact, short, far too long, beyond the buffer) - same table, byte for byte."""
    import struct
    from cutesv_amd import _cols_native as cn

    def table(blob, width=5, ints=(0, 1), strs=(2, 3, 4), off=0, hint=None):
        t = cn.pickle_table(blob, off, width, ints, strs) if hint is None else cn.pickle_table(blob, off, width, ints, strs, hint)
        if t is None:
            return None
        return (t[0], t[1], [np.frombuffer(x, np.int64).tolist() for x in t[2]],
                [[blob[int(o):int(o) + int(n)] for o, n in zip(np.frombuffer(a, np.int64), np.frombuffer(b, np.int32))] for a, b in t[3]], t[4])

    def expect(rows, blob, strs=(2, 3, 4)):
        return (len(rows), len(blob), [[int(r[0]) for r in rows], [int(r[1]) for r in rows]], [[r[k].encode("utf-8", "surrogatepass") for r in rows] for k in strs], True)

    rng = np.random.default_rng(5)
    chr1, kind = "chr1", "DEL"
    rows = []
    for i in range(5000):
        pos = int(rng.integers(0, 1 << 31))
        if i % 7 == 0:
            pos = pos + 0.5                                       # BINFLOAT
        if i % 11 == 0:
            pos = int(rng.integers(1 << 33, 1 << 45))             # LONG1
        rows.append((pos, int(rng.integers(-70000, 70000)), "read%d" % (i // 2), kind, chr1))
    for i in range(0, 5000, 2):                                   # the same str OBJECT twice: the second row's name is a memo reference
        rows[i + 1] = rows[i + 1][:2] + (rows[i][2],) + rows[i + 1][3:]
    for proto in (2, 3, 4, 5):
        blob = pickle.dumps(rows, protocol=proto)
        want = expect(rows, blob)
        assert table(blob) == want
        for hint in (len(blob), len(blob) // 3, 10, len(blob) * 50, -5, 0):
            assert table(blob, hint=hint) == want, (proto, hint)
        pad = b"x" * 40
        assert table(pad + blob + pad, off=40, hint=40 + len(blob))[2] == want[2]
        # a list element that is not one of our rows, far into the stream: declined whichever path meets it
        for bad in ((1, 2, "r", kind), (1, 2, 3, kind, chr1), (1, "x", "r", kind, chr1), (None, 2, "r", kind, chr1), (1, 2, ("r",), kind, chr1)):
            assert table(pickle.dumps(rows[:3000] + [bad] + rows[3000:], protocol=proto)) is None, (proto, bad)
    # by hand (protocol 2 opcodes): ] q0 ( rows... e .   with a FRAME between two fields, LONG_BINPUT, and a buffer that ends right
    # behind STOP - the last rows of every buffer are read by the opcode loop
    head = b"\x80\x02]q\x00("                                     # memo 0: the list; then the batch's MARK
    first = b"(K\x07K\x08X\x02\x00\x00\x00r0q\x01X\x03\x00\x00\x00DELq\x02X\x01\x00\x00\x001q\x03tq\x04"      # memo 1 the name, 2 "DEL", 3 "1", 4 the tuple
    stream, names, memo = head + first, [b"r0"], 5
    for i in range(1, 300):
        nm = b"name%d" % i
        stream += b"(" + b"J" + struct.pack("<i", 100 + i) + (b"\x95" + struct.pack("<Q", 0) if i % 5 == 0 else b"") + b"M" + struct.pack("<H", i) + \
                  b"X" + struct.pack("<I", len(nm)) + nm + ((b"r" + struct.pack("<I", memo)) if i % 3 == 0 else (b"q" + bytes([memo]))) + b"h\x02h\x03t"
        memo += 1
        if memo < 250 and i % 2:
            stream += b"q" + bytes([memo])                        # the tuple memoized too (every other row)
            memo += 1
        names.append(nm)
        if memo >= 250:
            break
    stream += b"e."
    got = table(stream, strs=(2,))
    ref = pickle.loads(stream)
    assert got is not None and got[0] == len(ref) == len(names)
    assert got[2] == [[r[0] for r in ref], [r[1] for r in ref]] and got[3] == [[r[2].encode() for r in ref]]
    assert all(r[3] == "DEL" and r[4] == "1" for r in ref)
    for cut in range(1, 40):                                      # truncated anywhere near the end: refused or declined, never a table
        try:
            assert cn.pickle_table(stream[:-cut], 0, 5, (0, 1), (2,)) is None
        except ValueError:
            pass


def test_walked_reads_block_round_trips_through_shared_memory(tmp_path):
    """columns.WalkedReads - a chromosome's reads block as the workers of a pool hand it round (broker.Client.reads_put / reads_get):
    laid out in a buffer and read back it is the same block, with the chromosome span kept once when every row carries the same
    one (29 bytes a row) and per row otherwise; a task store built from a block that went through the buffer is the store built
    from the walk itself, with and without the reads-near-a-window filter."""
    from cutesv_amd import _cols_native as cn
    from cutesv_amd.columns import WalkedReads
    rng = np.random.default_rng(9)
    names = ["read%d" % i for i in range(300)]
    sigs = [(int(p_), 50 + i % 7, names[int(rng.integers(0, 300))], "DEL", "chr1") for i, p_ in enumerate(np.sort(rng.integers(1000, 900_000, 400)))]
    for n_chr in (1, 2):
        c = ["chr1", "chr2"]
        reads = [(int(s_), int(s_) + int(rng.integers(100, 30_000)), bool(i % 5), names[int(rng.integers(0, 300))], c[(i % 4 == 0) * (n_chr - 1)])
                 for i, s_ in enumerate(np.sort(rng.integers(0, 1_000_000, 3000)))]
        rblob, sblob = pickle.dumps(reads, protocol=4), pickle.dumps(sigs, protocol=4)
        wr = WalkedReads.from_table(cn.pickle_table(rblob, 0, 5, (0, 1, 2), (3, 4)))
        assert wr.n == 3000 and wr.one_chr == (n_chr == 1) and len(wr.chr_off) == (1 if n_chr == 1 else 3000)
        assert wr.nbytes() < 3000 * (30 if n_chr == 1 else 42) + 64 * 8
        buf = bytearray(wr.nbytes())
        wr.write_into(buf)
        back = WalkedReads.from_buffer(bytes(buf))
        assert back.n == wr.n and back.one_chr == wr.one_chr
        fd = os.memfd_create("walked", 0)                         # what reads_put does: the same bytes, written column by column
        try:
            os.ftruncate(fd, wr.nbytes())
            wr.write_fd(fd)
            assert os.pread(fd, wr.nbytes() + 1, 0) == bytes(buf)
        finally:
            os.close(fd)
        for k, dt in WalkedReads.FIELDS:
            assert getattr(back, k).dtype == dt and np.array_equal(getattr(back, k), getattr(wr, k)), k
        sel = np.array([0, 5, 2999])
        o, l = back.chr_spans(sel)
        assert [rblob[int(x):int(x) + int(y)].decode() for x, y in zip(o, l)] == [reads[int(i)][4] for i in sel]
        assert [len(x) for x in back.chr_spans()] == [3000, 3000]

        class Shelf:                                             # what broker.Client offers: the block comes back out of a buffer
            def __init__(self):
                self.d = {}

            def reads_get(self, key):
                return WalkedReads.from_buffer(self.d[key]) if key in self.d else None

            def reads_put(self, key, w):
                b = bytearray(w.nbytes())
                w.write_into(b)
                self.d[key] = bytes(b)
        for margin in (None, 1000):
            shelf = Shelf()
            direct = SigStore.from_task_pickles("DEL", "chr1", sblob, 0, rblob, 0, gt_margin=margin)
            first = SigStore.from_task_pickles("DEL", "chr1", sblob, 0, rblob, 0, gt_margin=margin, reads_cache=shelf, reads_key="k")
            again = SigStore.from_task_pickles("DEL", "chr1", sblob, 0, rblob, 0, gt_margin=margin, reads_cache=shelf, reads_key="k")
            assert list(shelf.d) == ["k"]
            _same_store(first, direct)
            _same_store(again, direct)
            want = SigStore.from_task_lists("DEL", "chr1", sigs, reads)
            if margin is None:
                _same_store(direct, want)
            else:
                assert 0 < direct.n_reads < want.n_reads


def test_block_end_is_the_next_offset_of_the_index():
    """resolve._block_end: the walker's size hint - where a chromosome's pickled block ends in its file (the blocks lie back to
    back, main script :817-857): the next larger offset of the index, the file's end for the last block, None when the index does
    not hold integers (the walker then doubles its buffers as before)"""
    from cutesv_amd import resolve
    idx = {"2": 700, "1": 0, "X": 1900, "10": 1200}                # (in file order: 1, 2, 10, X - not in key order)
    assert [resolve._block_end(idx, c, 2500) for c in ("1", "2", "10", "X")] == [700, 1200, 1900, 2500]
    assert resolve._block_end({"1": 0}, "1", 0) == 0 and resolve._block_end(idx, "nope", 10) is None
    assert resolve._block_end({"1": "zero"}, "1", 10) is None and resolve._block_end({"1": 0, "2": None}, "1", 10) is None


def test_reads_near_is_its_numpy_statement():
    """columns._reads_near (one C pass, `_cols_native.reads_near`) against the rule written out in numpy - the form it replaced:
    a flag per 2^shift-bp bin of the union of [x - margin, x + margin] over the task's coordinates, a read stays iff a flagged bin
    lies in [start, end]; None (keep every read) for empty inputs and for coordinates outside [0, 2^40)."""
    from cutesv_amd.columns import _reads_near

    def statement(pos1, pos2, r_start, r_end, margin, shift=10):
        n = len(pos1)
        if n == 0 or len(r_start) == 0:
            return None
        xs = pos1 if pos2 is None else np.concatenate([pos1, pos2])
        hi = int(max(int(xs.max()), int(r_end.max()))) + margin + (2 << shift)
        if not (int(xs.min()) >= 0 and int(r_start.min()) >= 0) or hi >= (1 << 40):
            return None
        nb = (hi >> shift) + 2
        d = np.zeros(nb + 1, np.int64)
        np.add.at(d, np.maximum(xs - margin, 0) >> shift, 1)
        np.add.at(d, ((xs + margin) >> shift) + 1, -1)
        cum = np.zeros(nb + 1, np.int64)
        np.cumsum(np.cumsum(d[:-1]) > 0, out=cum[1:])
        rb0 = np.minimum(r_start >> shift, nb - 1)
        rb1 = np.minimum(np.maximum(r_end, r_start) >> shift, nb - 1)
        return (cum[rb1 + 1] - cum[rb0]) > 0

    rng = np.random.default_rng(3)
    kept = []
    for it in range(300):
        span = (1 << 32) if it % 60 == 59 else int(rng.choice([5_000, 200_000, 3_000_000]))
        n, nr = int(rng.integers(1, 400)), int(rng.integers(1, 3000))
        pos1 = rng.integers(0, span, n)
        pos2 = None if it % 3 else pos1 + rng.integers(0, 50_000, n)
        r_start = rng.integers(0, span, nr)
        r_end = r_start + rng.integers(-10, 40_000, nr)           # (an end in front of its start: counted as the start)
        margin = int(rng.choice([0, 1, 50, 1000, 5000]))
        shift = int(rng.choice([0, 4, 10, 13])) if span < (1 << 30) else 10
        got, want = _reads_near(pos1, pos2, r_start, r_end, margin, shift), statement(pos1, pos2, r_start, r_end, margin, shift)
        assert (got is None) == (want is None)
        if want is not None:
            assert got.dtype == np.bool_ and np.array_equal(got, want), it
            kept.append(got.mean())
    assert min(kept) < 0.05 and max(kept) > 0.95
    z = np.zeros(0, np.int64)
    one = np.array([5], np.int64)
    assert _reads_near(z, None, one, one, 10) is None and _reads_near(one, None, z, z, 10) is None
    assert _reads_near(np.array([-1]), None, one, one, 10) is None and _reads_near(one, None, np.array([-3]), one, 10) is None
    assert _reads_near(np.array([1 << 40]), None, one, one, 10) is None and _reads_near(one, None, one, np.array([(1 << 40) - 5]), 10) is None
    got = _reads_near(one, None, np.array([0, 5000]), np.array([10, 6000]), 100)
    assert got.tolist() == [True, False]
    got[0] = False                                                # (writable: from_task_pickles keeps one read of a block it would empty)


def test_rebuild_staging_fill_equals_the_concatenated_columns(monkeypatch):
    """rebuild.rebuild_to_device_batch writes the per-type rows once, in the ABI's widths, into the context's staging columns
    (threads, chunks of 2^19 rows): the same columns as the per-type conversions + np.concatenate they replaced"""
    from cutesv_amd import rebuild, engine
    from cutesv_amd.columns import TYPES
    monkeypatch.setattr(engine, "pinned_empty", lambda n, dt: np.empty(n, dt))
    rng = np.random.default_rng(1)
    chroms = ["2", "10", "1", "X"]
    per = {}
    for t, m in (("DEL", 1_300_000), ("INS", 700_001), ("TRA", 5)):
        per[t] = dict(chrom=rng.integers(0, 4, m), a=rng.integers(0, 1 << 40, m), b=rng.integers(0, 1 << 20, m),
                      read_id=rng.integers(0, 1 << 30, m).astype(np.int32), aux=rng.integers(0, 1000, m))
    seen = {}

    class Stop(Exception):
        pass

    def capture(ctx, seg, a, b, rid, aux, major, nodedup, **kw):
        seen.update(seg=seg.copy(), a=a.copy(), b=b.copy(), rid=rid.copy(), aux=aux.copy(), src_row_out=kw.get("src_row_out"))
        raise Stop()
    monkeypatch.setattr(rebuild, "rebuild_columns", capture)

    class Ctx:
        pass
    with pytest.raises(Stop):
        rebuild.rebuild_to_device_batch(Ctx(), chroms, per, None)
    order = sorted(range(len(chroms)), key=lambda i: chroms[i])
    crank = np.zeros(len(chroms), np.int64)
    crank[order] = np.arange(len(chroms))
    want = {k: [] for k in ("seg", "a", "b", "rid", "aux")}
    for ti, t in enumerate(TYPES):
        if t in per:
            d = per[t]
            want["seg"].append((ti * len(chroms) + crank[d["chrom"]]).astype(np.int32))
            want["a"].append(d["a"]); want["b"].append(d["b"]); want["rid"].append(d["read_id"]); want["aux"].append(d["aux"].astype(np.int32))
    for k, v in want.items():
        assert np.array_equal(seen[k], np.concatenate(v)) and seen[k].dtype == np.concatenate(v).dtype, k
    assert seen["src_row_out"] is not None and len(seen["src_row_out"]) == len(seen["a"])
    per["DEL"]["chrom"][7] = 4                                         # a chromosome index outside the table is an error, not a clip
    with pytest.raises(ValueError):
        rebuild.rebuild_to_device_batch(Ctx(), chroms, per, None)


@pytest.mark.parametrize("kw", [dict(narrow_support=True), dict(narrow_support=True, coord32=True), dict(),
                                dict(no_support=True, coord32=True, fields=("call_aux", "cipos", "cilen", "seq_pick", "dr", "gl_idx")),
                                dict(narrow_support=True, fields=())])
def test_block_result_arrays_sit_back_to_back(kw):
    """HostResult(block=...): the per-call arrays and the support list are carved out of ONE buffer, widest elements first, with
    no gap between neighbours and every array aligned to its element - what csv_batch_publish_async (include/cutesv_hip.h) needs to
    hand the result to the copy engine as one run of arrays; seg_status and the per-signature arrays stay outside"""
    bufs = []

    def block(nbytes):
        bufs.append(bytearray(nbytes))
        return bufs[-1]
    for cap_calls, cap_support in ((1, 1), (77, 501), (4096, 12345)):
        r = _abi.HostResult(1000, cap_calls, cap_support, n_seg=3, block=block, **kw)
        inside = sorted((a.__array_interface__["data"][0], a.nbytes, name) for name, a in r.arrays.items()
                        if a is not None and name not in ("seg_status", "cluster_id", "allele_id"))
        base = np.frombuffer(bufs[-1], np.uint8).__array_interface__["data"][0]
        assert inside[0][0] == base
        for (ad, nb, name), (ad2, _, name2) in zip(inside, inside[1:]):
            assert ad + nb == ad2, (name, name2)                   # exactly adjacent: one run for the copy engine
        assert inside[-1][0] + inside[-1][1] - base == sum(nb for _, nb, _ in inside) <= len(bufs[-1])
        for ad, _, name in inside:
            assert ad % r.arrays[name].dtype.itemsize == 0, name
        want = _abi.HostResult(1000, cap_calls, cap_support, n_seg=3, **kw)
        assert {k: (None if v is None else (v.dtype, v.shape)) for k, v in r.arrays.items()} == \
               {k: (None if v is None else (v.dtype, v.shape)) for k, v in want.arrays.items()}
        for name, a in r.arrays.items():                          # the struct carries the carved addresses
            if a is not None and name != "support_sig":
                assert getattr(r.c, name) == a.__array_interface__["data"][0], name
