"""The drop-in under the reference's own execution model (main script :1113-1199): a forked multiprocessing.Pool, one
map_async(run_X, [tuple]) per (chromosome, type), the five callables of cutesv_amd.resolve reading the reference's pickles.

CPU part (this file's unmarked tests): everything of that path that is not a kernel - the GPU broker's protocol, shared
regions, request merging and result slicing, the workers' side, device round-robin - with the C oracle behind the broker's
serving loop (tests/broker_oracle.py; the product's broker has the HIP library behind it and nothing else).
GPU part (`-m gpu`): the same stage in a FRESH interpreter (the pytest process already holds a HIP context), with the real
broker and with one context per worker, 1 / 4 / 16 workers on one device, against the reference's digests."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from cutesv_amd import _abi, broker, resolve, synth
from cutesv_amd.columns import Params
from helpers import load_json, store_from_json, assert_rows_equal, assert_soa_equal

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture()
def oracle_broker(monkeypatch):
    """the broker's serving loop over the oracle, named like resolve.warm_up() would name it for this process"""
    procs = []

    def start(devices=(0,), env=None):
        prefix = "cutesv_amd-test-%d-%d" % (os.getpid(), time.monotonic_ns())
        monkeypatch.setenv("CUTESV_AMD_BROKER_NAME", prefix)
        e = dict(os.environ, **(env or {}))
        e["PYTHONPATH"] = ROOT + os.pathsep + e.get("PYTHONPATH", "")
        for d in devices:
            name = broker.socket_name(os.getpid(), d)
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "broker_oracle.py"), "--name", name, "--device", str(d),
                                           "--watch-pid", str(os.getpid()), "--linger", "20"], env=e))
            t_end = time.monotonic() + 60
            while broker._try_connect(name) is None:
                assert procs[-1].poll() is None, "the test broker died"
                assert time.monotonic() < t_end
                time.sleep(0.01)
        return prefix
    yield start
    for p in procs:
        p.terminate()
    for p in procs:
        p.wait(timeout=10)


def _oracle_res(hb):
    from oracle import oracle
    return oracle.cluster_batch(hb, per_sig=False)


FIELDS = ("call_seg", "call_aux", "bp1", "bp2", "support", "cipos", "cilen", "search_pos", "seq_pick", "dr", "dv", "gl_idx", "support_off", "support_sig")


def _same(got, want, fields=FIELDS):
    for f in fields:
        if got[f] is not None:
            assert np.array_equal(np.asarray(got[f]).astype(np.int64), np.asarray(want[f]).astype(np.int64)), f


def test_requests_through_the_broker_equal_direct_calls(oracle_broker):
    oracle_broker()
    st = synth.small_mixed(seed=11, n_sites=30)
    p = Params.ont(genotype=True, min_support=3)
    with broker.Client.connect(0, owner_pid=os.getpid(), spawn=False) as cl:
        # a whole multi-segment batch (every field, int64 support list, per-signature outputs)
        hb = st.host_batch(st.tasks(), p)
        got = cl.cluster_batch(hb, per_sig=True).trimmed()
        from oracle import oracle
        want = oracle.cluster_batch(hb, per_sig=True).trimmed()
        assert_soa_equal(got, want, st)
        # one task at a time out of the genome's columns: the client sends the segment's rows and its chromosome's reads only,
        # and moves the signature indices of the result back
        for task in st.tasks():
            hb1 = st.host_batch([task], p)
            got = cl.cluster_batch(hb1, reuse=True, fields=resolve.ROW_FIELDS).trimmed()
            want = _oracle_res(hb1).trimmed()
            assert got["call_cluster"] is None and got["search_pos"] is None
            _same(got, want)
        info = cl.info()
        assert info["calls"] == 1 + len(st.tasks()) and info["engine"].startswith("oracle")
        # capacity negotiation travels: a result that is too small comes back with the sizes needed
        hb = st.host_batch(st.tasks(), p)
        got = cl.cluster_batch(hb, cap_calls=1, cap_support=1).trimmed()
        _same(got, _oracle_res(hb).trimmed())
        # an error of the library reaches the worker as the same exception (TRA genotyping without reference lengths)
        hbt = st.host_batch([t for t in st.tasks() if t[0] == "TRA"][:1], Params())
        hbt.segments["genotype"] = 1
        from cutesv_amd.engine import CsvError
        with pytest.raises(CsvError):
            cl.cluster_batch(hbt)
        # a request whose pointers leave its region is refused, not executed
        reg = cl.region
        import struct
        cin = _abi.BatchIn.from_buffer_copy(bytes(hb.c))
        cin.a = reg.base + reg.size + 4096
        data = cl._request(broker.K_CALL, struct.pack("<Q", reg.base) + bytes(cin) + bytes(_abi.BatchOut()))
        assert broker.REPLY.unpack(data[:broker.REPLY.size])[0] == _abi.E_INVALID


def test_waiting_requests_are_merged_into_one_batch(oracle_broker):
    oracle_broker(env={"CUTESV_AMD_BROKER_GATHER_MS": "300"})
    st = synth.small_mixed(seed=12, n_sites=40)
    p = Params.hifi(genotype=True, min_support=3)
    tasks = [t for t in st.tasks() if t[0] != "TRA"]
    out, errs = {}, []

    def work(task):
        try:
            with broker.Client.connect(0, owner_pid=os.getpid(), spawn=False) as cl:
                barrier.wait()
                hb1 = st.host_batch([task], p)
                r = cl.cluster_batch(hb1, reuse=True, fields=resolve.ROW_FIELDS)
                out[task] = ({k: (None if v is None else np.array(v)) for k, v in r.trimmed().items() if k != "n_clusters"}, r.n_clusters)
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    barrier = threading.Barrier(len(tasks))
    th = [threading.Thread(target=work, args=(t,)) for t in tasks]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for task in tasks:
        want = _oracle_res(st.host_batch([task], p)).trimmed()
        _same(out[task][0], want)
    with broker.Client.connect(0, owner_pid=os.getpid(), spawn=False) as cl:
        info = cl.info()
    assert info["merged_calls"] >= len(tasks) - 1 and max(info["engine_batch_sizes"]) >= len(tasks) - 1, info
    assert info["engine_calls"] < len(tasks)
    assert any(n == -1 for _, n in out.values())          # (cluster counts are a batch-level figure: documented as unknown)


def _golden_case(name):
    case = next(c for c in load_json("small_cases.json.gz") if c["name"] == name)
    return store_from_json(case["store"]), Params(**case["params"]), case


@pytest.mark.parametrize("name,threads", [("ont_gt", 4), ("realnames_gt", 3), ("hifi", 1)])
def test_phase3_pool_through_the_broker_returns_the_reference_rows(oracle_broker, tmp_path, name, threads, monkeypatch):
    oracle_broker()
    monkeypatch.setenv("CUTESV_AMD_TRA_GT", "off")       # (the reference's TRA rows of these cases were made with action=False)
    st, p, case = _golden_case(name)
    wd = str(tmp_path) + "/"
    idx = st.write_reference_workdir(wd)
    results = resolve.main_ctrl_phase3(wd, idx, p, threads)
    want_by_chr = {}
    for t in ("DEL", "INS", "INV", "DUP", "TRA"):
        for tt, c, rows in case["rows"]:
            if tt == t and rows:
                want_by_chr.setdefault(c, []).extend([(t, r) for r in rows])
    assert {c for c, r in results.items() if r} == set(want_by_chr)
    for c, want in want_by_chr.items():
        assert len(results[c]) == len(want)
        for g, (t, w) in zip(results[c], want):
            assert_rows_equal(t, [g], [w], where="pool %s" % c)
    if p.genotype:
        # the tasks of a chromosome shared its walked reads block through the broker (the first one left it there)
        with broker.Client.connect(0, owner_pid=os.getpid(), spawn=False) as cl:
            info = cl.info()
        assert info.get("blocks", 0) >= 1 and info.get("block_hits", 0) >= 1, info
    # the restatement's five callables on the same files, under the same harness (bench.py's mode1_stage baseline)
    from oracle import py_restatement as pr
    ref = resolve.main_ctrl_phase3(wd, idx, p, threads, fns=pr.REF_FNS)
    for c, want in want_by_chr.items():
        for g, (t, w) in zip(ref[c], want):
            assert_rows_equal(t, [g], [w], where="restatement pool %s" % c)


def test_a_failing_task_does_not_hang_the_pool(oracle_broker, tmp_path, monkeypatch):
    oracle_broker()
    monkeypatch.setenv("CUTESV_AMD_TRA_GT", "off")
    st, p, case = _golden_case("ont_gt")
    wd = str(tmp_path) + "/"
    idx = st.write_reference_workdir(wd)
    bad = dict(idx, DEL=dict(idx["DEL"]))
    victim = next(iter(bad["DEL"]))
    bad["DEL"][victim] = os.path.getsize(wd + "DEL.pickle") + 77        # an offset behind the file: that task raises in its worker
    errors = []
    results = resolve.main_ctrl_phase3(wd, bad, p, 3, on_error=errors.append)
    assert len(errors) == 1
    good = resolve.main_ctrl_phase3(wd, idx, p, 3)
    want_del = next(rows for t, c, rows in case["rows"] if t == "DEL" and c == victim)
    assert sum(len(v) for v in good.values()) - sum(len(v) for v in results.values()) == len(want_del) > 0


def _device_of(_):
    time.sleep(0.05)
    return resolve.device_index(), os.getpid()


def test_pool_workers_share_the_devices_round_robin(monkeypatch):
    import multiprocessing as mp
    monkeypatch.delenv("CUTESV_AMD_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    monkeypatch.setenv("CUTESV_AMD_DEVICES", "4")
    with mp.get_context("fork").Pool(8) as pool:
        got = pool.map(_device_of, range(64), chunksize=1)
    by_pid = {}
    for d, pid in got:
        by_pid.setdefault(pid, set()).add(d)
    assert all(len(v) == 1 for v in by_pid.values())                 # a worker keeps its device
    assert {next(iter(v)) for v in by_pid.values()} == {0, 1, 2, 3}
    monkeypatch.setenv("CUTESV_AMD_DEVICES", "2,5")
    with mp.get_context("fork").Pool(4) as pool:
        assert {d for d, _ in pool.map(_device_of, range(32), chunksize=1)} == {2, 5}
    monkeypatch.setenv("CUTESV_AMD_DEVICE", "3")
    assert resolve.device_index() == 3


def test_two_brokers_two_devices(oracle_broker, tmp_path, monkeypatch):
    """CUTESV_AMD_DEVICES=2: the pool's workers split over the brokers of device 0 and 1; the rows do not change"""
    oracle_broker(devices=(0, 1))
    monkeypatch.setenv("CUTESV_AMD_DEVICES", "2")
    monkeypatch.setenv("CUTESV_AMD_TRA_GT", "off")
    st, p, case = _golden_case("ont_gt")
    wd = str(tmp_path) + "/"
    idx = st.write_reference_workdir(wd)
    results = resolve.main_ctrl_phase3(wd, idx, p, 4)
    n_want = sum(len(rows) for _, _, rows in case["rows"])
    assert sum(len(v) for v in results.values()) == n_want
    calls = []
    for d in (0, 1):
        with broker.Client.connect(d, owner_pid=os.getpid(), spawn=False) as cl:
            calls.append(cl.info()["calls"])
    assert all(c > 0 for c in calls), calls


# ================================================================================================ GPU: a fresh interpreter per stage
STAGE = os.path.join(HERE, "pool_stage_main.py")


def _run_stage(tmp_path, cfg, threads, mode, extra_env=None):
    out = str(tmp_path / ("stage_%s_%d_%s.json" % (cfg, threads, mode)))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **(extra_env or {}))
    for k in ("CUTESV_AMD_BROKER_NAME", "CUTESV_AMD_DEVICE", "CUTESV_AMD_DEVICES"):
        env.pop(k, None)
    subprocess.run([sys.executable, STAGE, "--cfg", cfg, "--threads", str(threads), "--mode", mode, "--out", out, "--work", str(tmp_path)],
                   check=True, env=env, timeout=900)
    with open(out) as f:
        return json.load(f)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["broker", "warm", "direct"])
@pytest.mark.parametrize("threads", [1, 4, 16])
@pytest.mark.parametrize("cfg", ["cfg3_s025", "cfg4_s002"])
def test_forked_pool_stage_on_one_device(tmp_path, cfg, threads, mode):
    """The literal phase-3 block with cutesv_amd.resolve.run_* under a forked Pool(T) on ONE device, in a fresh interpreter:
    broker started by the first worker / by warm_up() in the parent / one HIP context per worker (created after fork).
    Checks per (type, chromosome) digests of the reference's rows, that the parent never touched HIP, where the contexts
    lived, and that a failing task is swallowed like the reference swallows it."""
    r = _run_stage(tmp_path, cfg, threads, mode)
    d = load_json("digests.json")[cfg]
    assert r["parent_loaded_hip_library"] is False and r["parent_has_context"] is False
    assert r["errors"] == 1 and r["bad_task_rows"] == 0                 # the injected failing task, and only it
    got = r["digests"]
    for key, (n, h) in d["segments"].items():
        if n:
            assert got[key] == [n, h], key
    assert sum(1 for k, v in got.items() if v[0]) == sum(1 for v in d["segments"].values() if v[0])
    if mode == "direct":
        assert len(r["worker_context_pids"]) >= 1 and r["parent_pid"] not in r["worker_context_pids"]
        assert all(k == "Context" for k in r["worker_engine_kinds"])
    else:
        assert all(k == "Client" for k in r["worker_engine_kinds"])
        assert r["broker"]["engine"] == "libcutesv_hip.so" and r["broker"]["pid"] not in r["worker_pids"] + [r["parent_pid"]]
        assert r["broker"]["calls"] >= r["tasks"] - 1
    assert r["leftover_brokers"] == 0


class _OracleCtx:
    """stands in for the engine where only csv_cluster_batch's result matters (the task-store side is host code)"""

    def cluster_batch(self, hb, reuse=False, **kw):
        from oracle import oracle
        return oracle.cluster_batch(hb, per_sig=False)


def test_a_task_keeps_only_the_reads_that_can_reach_its_windows(tmp_path, monkeypatch):
    """SigStore.from_task_pickles(gt_margin=...) drops the reads of the chromosome's block that cannot cover any genotyping
    window of the task (columns._reads_near): the rows - DR, GT, PL, GQ, QUAL of every call, all types - must equal the rows
    from the whole block, and most of the block must be gone on a genome-like layout"""
    from cutesv_amd.columns import SigStore
    monkeypatch.setattr(resolve, "_ctx", _OracleCtx())
    monkeypatch.setattr(resolve, "_ctx_pid", None)
    monkeypatch.setenv("CUTESV_AMD_TRA_GT", "off")
    kept_share = []
    for seed, kw, p in ((201, dict(n_sites=40, coverage=30, contig_len=3_000_000, n_contigs=2, n_noise=40), Params.ont(genotype=True, min_support=3)),
                        (202, dict(n_sites=25, coverage=20, contig_len=2_000_000, n_contigs=3, n_noise=200, n_loci=10), Params.hifi(genotype=True, min_support=3)),
                        (203, dict(n_sites=60, coverage=12, contig_len=800_000, n_contigs=2), Params(genotype=True, min_support=2, max_cluster_bias_DEL=50, max_cluster_bias_INV=2000))):
        st = synth.small_mixed(seed=seed, genotype=True, **kw)
        wd = str(tmp_path / ("w%d" % seed)) + "/"
        os.makedirs(wd)
        idx = st.write_reference_workdir(wd)

        def stage():
            out = {}
            for t, c in st.tasks():
                if t == "DEL":
                    r = resolve.run_del((wd, c, "DEL", p.min_support, p.diff_ratio_merging_DEL, p.max_cluster_bias_DEL, min(p.min_support, 5), "bam", True, p.gt_round, p.remain_reads_ratio, idx))
                elif t == "INS":
                    r = resolve.run_ins((wd, c, "INS", p.min_support, p.diff_ratio_merging_INS, p.max_cluster_bias_INS, min(p.min_support, 5), "bam", True, p.gt_round, p.remain_reads_ratio, idx))
                elif t == "INV":
                    r = resolve.run_inv((wd, c, "INV", p.min_support, p.max_cluster_bias_INV, p.min_size, "bam", True, p.max_size, p.gt_round, idx))
                elif t == "DUP":
                    r = resolve.run_dup((wd, c, p.min_support, p.max_cluster_bias_DUP, p.min_size, "bam", True, p.max_size, p.gt_round, idx))
                else:
                    continue
                out[(t, c)] = [helpers_canonical(t, x) for x in r[1]]
            return out
        near = stage()
        monkeypatch.setenv("CUTESV_AMD_ALL_READS", "1")
        whole = stage()
        monkeypatch.delenv("CUTESV_AMD_ALL_READS")
        assert near == whole and sum(len(v) for v in near.values()) > 20
        assert any(row[7] not in (".", "0") for rows in near.values() for row in rows if row[1] in ("DEL", "INS"))      # some DR > 0: reads do cover
        for t, c in st.tasks():
            if t in ("DEL", "INS"):
                m = resolve._mapped(wd + t + ".pickle"), resolve._mapped(wd + "reads.pickle")
                a = SigStore.from_task_pickles(t, c, m[0], idx[t][c], m[1], idx["reads"][c], gt_margin=1000)
                b = SigStore.from_task_pickles(t, c, m[0], idx[t][c], m[1], idx["reads"][c])
                kept_share.append(a.n_reads / max(1, b.n_reads))
    assert min(kept_share) < 0.5, kept_share


def helpers_canonical(t, row):
    from helpers import canonical_row
    return canonical_row(t, row)


def test_a_worker_survives_its_brokers_death(oracle_broker, tmp_path, monkeypatch):
    """the broker is killed between two tasks of one worker: the next task raises once at most and the worker reconnects (here: to a
    broker started again under the same name) instead of failing every later task on a dead socket"""
    import signal
    monkeypatch.setenv("CUTESV_AMD_TRA_GT", "off")
    monkeypatch.setenv("CUTESV_AMD_BROKER", "1")
    monkeypatch.setattr(resolve, "_ctx", None)
    monkeypatch.setattr(broker, "spawn", lambda name, device, watch_pid, linger=None, log=None, prealloc=None: subprocess.Popen(
        [sys.executable, os.path.join(HERE, "broker_oracle.py"), "--name", name, "--device", str(device), "--watch-pid", str(watch_pid), "--linger", "5"],
        env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))))
    oracle_broker()
    st, p, case = _golden_case("ont_gt")
    wd = str(tmp_path) + "/"
    idx = st.write_reference_workdir(wd)
    c = next(iter(idx["DEL"]))
    args = (wd, c, "DEL", p.min_support, p.diff_ratio_merging_DEL, p.max_cluster_bias_DEL, min(p.min_support, 5), "bam", p.genotype, p.gt_round, p.remain_reads_ratio, idx)
    want = resolve.run_del(args)
    with broker.Client.connect(0, owner_pid=os.getpid(), spawn=False) as cl:
        pid = cl.info()["pid"]
    os.kill(pid, signal.SIGKILL)
    t_end = time.monotonic() + 10
    while broker._try_connect(broker.socket_name(os.getpid(), 0)) is not None and time.monotonic() < t_end:
        time.sleep(0.02)
    again = resolve.run_del(args)                       # the dead socket is noticed, a new broker is started, the task is redone
    assert again == want and len(want[1]) > 0
    with broker.Client.connect(0, owner_pid=os.getpid(), spawn=False) as cl:
        assert cl.info()["pid"] != pid
        cl.shutdown()
    resolve._ctx.close()
    resolve._ctx = None


def test_warm_up_starts_the_brokers_and_shut_down_stops_them(tmp_path, monkeypatch):
    """resolve.warm_up() in the pool's parent: one broker per device, named after this process, found by the forked workers through
    the inherited CUTESV_AMD_BROKER_NAME; the CPU-side state the workers would build in their first task (extension modules, the
    cal_GL table) is in place before the fork; shut_down() stops the brokers"""
    from cutesv_amd import genotype
    monkeypatch.setenv("CUTESV_AMD_TRA_GT", "off")
    monkeypatch.setenv("CUTESV_AMD_BROKER", "1")
    monkeypatch.setenv("CUTESV_AMD_DEVICES", "2")
    monkeypatch.delenv("CUTESV_AMD_BROKER_NAME", raising=False)
    monkeypatch.setattr(resolve, "_ctx", None)
    monkeypatch.setattr(broker, "spawn", lambda name, device, watch_pid, linger=None, log=None, prealloc=None: subprocess.Popen(
        [sys.executable, os.path.join(HERE, "broker_oracle.py"), "--name", name, "--device", str(device), "--watch-pid", str(watch_pid), "--linger", "5"],
        env=dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))))
    names = resolve.warm_up()
    try:
        assert len(names) == 2 and os.environ["CUTESV_AMD_BROKER_NAME"].endswith("-%d" % os.getpid())
        assert len(genotype._table) >= 5151                               # the whole cal_GL table, inherited by the workers
        t_end = time.monotonic() + 30
        while any(broker._try_connect(n) is None for n in names):
            assert time.monotonic() < t_end
            time.sleep(0.02)
        st, p, case = _golden_case("ont_gt")
        wd = str(tmp_path) + "/"
        idx = st.write_reference_workdir(wd)
        results = resolve.main_ctrl_phase3(wd, idx, p, 4)
        assert sum(len(v) for v in results.values()) == sum(len(rows) for _, _, rows in case["rows"])
        used = []
        for d in (0, 1):
            with broker.Client.connect(d, owner_pid=os.getpid(), spawn=False) as cl:
                used.append(cl.info()["calls"])
        assert all(u > 0 for u in used), used
    finally:
        resolve.shut_down()
    assert all(broker._try_connect(n) is None for n in names)
    assert "CUTESV_AMD_BROKER_NAME" not in os.environ


def test_merged_batches_stay_inside_preallocated_staging():
    """Broker._groups: waiting single-segment requests are laid side by side - at most max_batch of them, and, behind a broker whose
    staging was page-locked on purpose (resolve.warm_up's prealloc), never more signatures / reads-table rows than that staging
    holds: the batch is cut instead of a few hundred MB of page-locked memory growing in the middle of a stage.  A staging block
    that grew out of the first request of a cold broker is no limit (it keeps growing)"""
    import types
    from cutesv_amd import broker as bk

    def req(n_sig, n_reads=0):
        seg = np.zeros(1, _abi.SEGMENT_DTYPE)
        seg["genotype"], seg["chrom"] = (1 if n_reads else 0), 0
        off = np.array([0, n_reads], np.int64)
        cin = _abi.BatchIn(n_seg=1, n_chrom=1, seg=seg.ctypes.data, n_sig=n_sig, n_reads=n_reads, reads_off=off.ctypes.data if n_reads else None)
        return types.SimpleNamespace(cin=cin, _keep=(seg, off))
    b = bk.Broker.__new__(bk.Broker)
    b.max_batch, b.prealloc, b._stage = 4, (1000, 5000), dict(n=1000, r=5000, k=4)
    sizes = lambda groups: [[(int(p.cin.n_sig), int(p.cin.n_reads)) for p in g] for g in groups]     # noqa: E731
    pend = [req(300), req(300), req(300), req(300), req(50), req(50), req(50), req(50), req(50)]
    assert sizes(b._groups(pend)) == [[(300, 0)] * 3, [(300, 0), (50, 0), (50, 0), (50, 0)], [(50, 0)] * 2]      # signatures, then max_batch
    pend = [req(10, 3000), req(10, 3000), req(10, 1000), req(2000, 100)]
    assert sizes(b._groups(pend)) == [[(10, 3000)], [(10, 3000), (10, 1000)], [(2000, 100)]]                 # reads rows; an oversize request alone
    b.prealloc = None                                                                                       # a cold broker: only max_batch
    assert [len(g) for g in b._groups([req(300) for _ in range(9)])] == [4, 4, 1]
