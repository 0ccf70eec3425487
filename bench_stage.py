"""bench.py's `mode1_stage` leg (also runnable by itself): the clustering stage the way cuteSV runs it.

The reference's phase 3 (main script :1113-1199) is a forked `Pool(processes=threads)` with one `map_async(run_X, [tuple])`
per (chromosome, type); its workers read their task from `<TYPE>.pickle` / `reads.pickle`.  This leg writes a workload as
exactly those files (outside every timed region) and times, from `Pool(...)` to the merged `results` dict:

  * the drop-in - `cutesv_amd.resolve.run_*` under that pool, T workers sharing ONE GPU through its broker
    (`cutesv_amd/broker.py`); `warm`: the broker was started ahead of the stage (`resolve.warm_up()`, as a cuteSV run would do
    at start-up: the HIP runtime's start overlaps the extraction phase); `cold`: the first worker starts it inside the
    timed region; `direct`: one HIP context per worker, created after fork (CUTESV_AMD_BROKER=0);
  * the reference's execution model - `oracle/py_restatement.py`'s five callables under the same pool, the same T, reading
    the SAME pickles (both sides pay for `pickle`).

    python bench_stage.py --workload cfg4 --workers 1,8,32
"""
import argparse
import hashlib
import json
import gc
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cutesv_amd import resolve, broker                 # noqa: E402

READS_FIELD = {"DEL": 12, "INS": 12, "DUP": 10, "INV": 11}


def rows_digest(results):
    """order-independent over chromosomes, order-dependent inside one; DUP / TRA read lists as sets (the reference builds them
    from Python sets: DUP:82, TRA:182)"""
    h = hashlib.sha256()
    n = 0
    for ch in sorted(results):
        for r in results[ch]:
            r = list(r)
            k = READS_FIELD.get(r[1], 11)
            if r[1] == "DUP" or r[1] not in READS_FIELD:
                r[k] = ",".join(sorted(r[k].split(",")))
            h.update("\t".join(r).encode())
            h.update(b"\n")
            n += 1
    return n, h.hexdigest()


def _stage(wd, idx, params, T, fns=None):
    if fns is None:
        # a timed stage starts with an empty broker: the walked reads blocks its workers share (the second task of a chromosome
        # finds what the first one left) must be ITS OWN, not those of the repetition before
        try:
            for d in _devices():
                with broker.Client.connect(d, owner_pid=os.getpid(), spawn=False) as cl:
                    cl.reads_flush()
        except broker.BrokerError:
            pass
    # every timed stage - either side's - starts from a collected heap: the rows of the stage before (350 000 lists and strings) make
    # the parent's collector passes, which run while it unpickles the workers' rows, 40-50 ms longer for whichever stage comes second
    gc.collect()
    t0 = time.perf_counter()
    res = resolve.main_ctrl_phase3(wd, idx, params, T, fns=fns)
    return time.perf_counter() - t0, res


def _devices():
    return resolve.device_list() or [resolve.device_index()]


def _broker_info(shutdown=False):
    """report of the first device's broker (None if any device's broker is missing); shutdown=True stops them all"""
    first = None
    try:
        for d in _devices():
            with broker.Client.connect(d, owner_pid=os.getpid(), spawn=False) as cl:
                info = cl.info()
                first = first or info
                if first is not info:
                    for k in ("calls", "batches", "merged_calls", "busy_s", "maps", "stage_in_s", "engine_s", "slice_out_s", "blocks", "block_hits"):
                        first[k] = first.get(k, 0) + info.get(k, 0)
                if shutdown:
                    cl.shutdown()
        return first
    except broker.BrokerError:
        return None


def _wait_gone(timeout=20.0):
    t_end = time.monotonic() + timeout
    for d in _devices():
        name = broker.socket_name(os.getpid(), d)
        while broker._try_connect(name) is not None and time.monotonic() < t_end:
            time.sleep(0.02)


def mode1_stage(name, store, params, workers=(1, 8, 32), reference=True, cold_and_direct_at=8, reps=2, keep_dir=None, log=None, cols_leg=True, devices=1):
    """-> dict for the bench line.  Must run before this process holds any HIP state (the pools fork).
    devices > 1: the pool's workers share that many GPUs (CUTESV_AMD_DEVICES), one broker each."""
    if devices and int(devices) > 1:
        os.environ["CUTESV_AMD_DEVICES"] = str(int(devices))
    from oracle import py_restatement as pr                       # (the CPU baseline of this leg)
    say = log or (lambda s: sys.stderr.write("[mode1_stage %s] %s\n" % (name, s)))
    wd = (keep_dir or tempfile.mkdtemp(prefix="cutesv_amd_stage_")) + "/"
    os.makedirs(wd, exist_ok=True)
    out = {"workload": name, "signatures": int(store.n_sig), "reads": int(store.n_reads), "devices": len(_devices()),
           "region": "Pool(processes=T) -> one map_async(run_X, [tuple]) per (chr, type) on the reference's pickles -> merged results dict "
                     "(main script :1113-1199); pickles written outside the timed region"}
    try:
        t0 = time.perf_counter()
        idx = store.write_reference_workdir(wd)
        out["tasks"] = sum(len(idx[t]) for t in ("DEL", "INS", "INV", "DUP", "TRA"))
        out["write_workdir_s"] = round(time.perf_counter() - t0, 2)
        out["pickle_bytes"] = sum(os.path.getsize(os.path.join(wd, f)) for f in os.listdir(wd))
        os.environ.setdefault("CUTESV_AMD_TRA_GT", "off")         # (the synthetic work dirs have no BAM to re-open, TRA:258-309)
        n_sig = out["signatures"]
        legs = []
        os.environ["CUTESV_AMD_BROKER"] = "1"
        t0 = time.perf_counter()
        resolve.warm_up()
        # a worker's first request waits for the broker's context anyway; make "warm" mean warm
        deadline = time.monotonic() + 180
        info = None
        while info is None and time.monotonic() < deadline:
            info = _broker_info()
            if info is None:
                time.sleep(0.02)
        out["broker_start_s"] = round(time.perf_counter() - t0, 3)
        out["broker_engine_start_s"] = None if info is None else info.get("engine_start_s")
        ref_digest = None
        for T in workers:
            walls, res = [], None
            for _ in range(reps):
                res = None                                # (the rows of the repetition before are gone before this one is timed)
                dt, res = _stage(wd, idx, params, T)
                walls.append(dt)
            dg = rows_digest(res)
            res = None
            leg = {"workers": T, "wall_ms": round(min(walls) * 1e3, 2), "wall_ms_all": [round(w * 1e3, 1) for w in walls], "rows": dg[0],
                   "signatures_per_s": round(n_sig / min(walls)),
                   # the same pool with tasks that do nothing: what neither side of the comparison can get under
                   "pool_alone_ms": round(min(pr.pool_startup_seconds(T, out["tasks"]) for _ in range(2)) * 1e3, 1)}
            say("drop-in warm broker T=%d: %s ms" % (T, leg["wall_ms_all"]))
            if reference:
                dtr, ref = _stage(wd, idx, params, T, fns=pr.REF_FNS)
                rd = rows_digest(ref)
                ref = None
                ref_digest = ref_digest or rd
                leg.update(reference_pool_wall_ms=round(dtr * 1e3, 1), vs_reference_pool=round(dtr / min(walls), 1), rows_equal_reference_model=(rd == dg))
                say("reference model T=%d: %.1f ms (x%.1f), rows equal: %s" % (T, dtr * 1e3, dtr / min(walls), rd == dg))
            legs.append(leg)
        if cols_leg:
            # the same pool, the same callables, the same argument tuples - on the flat column directory the build's own
            # rebuild step leaves next to the pickles (SURVEY 8(b) "Input files": <work_dir>/cutesv_amd.cols, SigStore.save);
            # a worker then maps its columns instead of walking a pickle.  The reference side of the ratio is unchanged.
            t0 = time.perf_counter()
            store.save(wd + "cutesv_amd.cols")
            out["write_cols_s"] = round(time.perf_counter() - t0, 2)
            for leg in legs:
                T = leg["workers"]
                walls, res = [], None
                for _ in range(reps):
                    res = None
                    dt, res = _stage(wd, idx, params, T)
                    walls.append(dt)
                leg["cols_wall_ms"] = round(min(walls) * 1e3, 2)
                leg["cols_wall_ms_all"] = [round(w * 1e3, 1) for w in walls]
                leg["cols_rows_equal"] = rows_digest(res) == ((leg["rows"], ref_digest[1]) if ref_digest else rows_digest(res))
                res = None
                if "reference_pool_wall_ms" in leg:
                    leg["cols_vs_reference_pool"] = round(leg["reference_pool_wall_ms"] / leg["cols_wall_ms"], 1)
                say("drop-in on cutesv_amd.cols T=%d: %s ms" % (T, [round(w * 1e3, 1) for w in walls]))
            shutil.rmtree(wd + "cutesv_amd.cols", ignore_errors=True)
            resolve._stores.clear()
        info = _broker_info(shutdown=True)
        _wait_gone()
        resolve.shut_down()
        out["legs"] = legs
        out["broker"] = None if info is None else {k: info.get(k) for k in ("calls", "batches", "merged_calls", "max_batch", "busy_s", "stage_in_s", "engine_s", "slice_out_s", "maps", "blocks", "block_hits", "stage_grows", "bus", "engine")}
        if cold_and_direct_at:
            T = cold_and_direct_at
            os.environ.pop("CUTESV_AMD_BROKER_NAME", None)
            dt, res = _stage(wd, idx, params, T)                   # cold: the first worker starts the broker
            out["cold_broker"] = {"workers": T, "wall_ms": round(dt * 1e3, 1), "rows": rows_digest(res)[0]}
            say("drop-in cold broker T=%d: %.1f ms" % (T, dt * 1e3))
            _broker_info(shutdown=True)
            _wait_gone()
            os.environ["CUTESV_AMD_BROKER"] = "0"                  # direct: a HIP context per worker, created after fork
            walls = []
            for _ in range(2):
                dt, res = _stage(wd, idx, params, T)
                walls.append(dt)
            out["context_per_worker"] = {"workers": T, "wall_ms": round(min(walls) * 1e3, 1), "wall_ms_all": [round(w * 1e3, 1) for w in walls],
                                         "rows": rows_digest(res)[0]}
            say("drop-in context per worker T=%d: %s ms" % (T, out["context_per_worker"]["wall_ms_all"]))
    finally:
        os.environ.pop("CUTESV_AMD_DEVICES", None) if devices and int(devices) > 1 else None
        os.environ.pop("CUTESV_AMD_BROKER", None)
        os.environ.pop("CUTESV_AMD_BROKER_NAME", None)
        if keep_dir is None:
            shutil.rmtree(wd, ignore_errors=True)
    return out


def compact(m):
    """what fits the driver's line: {workers, wall_ms, signatures_per_s, vs_reference_pool} per T"""
    if not isinstance(m, dict) or "legs" not in m:
        return m
    c = {"tasks": m.get("tasks"), "legs": [{k: leg.get(k) for k in ("workers", "wall_ms", "signatures_per_s", "reference_pool_wall_ms", "vs_reference_pool",
                                                                       "rows_equal_reference_model", "cols_wall_ms", "cols_vs_reference_pool", "pool_alone_ms") if k in leg} for leg in m["legs"]]}
    for k in ("cold_broker", "context_per_worker"):
        if m.get(k):
            c[k] = {"workers": m[k]["workers"], "wall_ms": m[k]["wall_ms"]}
    c["broker_start_s"] = m.get("broker_start_s")
    if m.get("broker"):
        c["merged_calls"] = m["broker"].get("merged_calls")
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--workers", default="1,8,32")
    ap.add_argument("--no-reference", action="store_true")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--keep-dir", default=None)
    ap.add_argument("--devices", type=int, default=1, help="GPUs the pool's workers share (one broker each)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-broker and context-per-worker legs")
    a = ap.parse_args()
    import bench
    store, params, wl = bench.make_workload(a.workload, a.scale, 0)
    m = mode1_stage(a.workload, store, params, workers=tuple(int(x) for x in a.workers.split(",")), reference=not a.no_reference,
                    reps=a.reps, keep_dir=a.keep_dir, devices=a.devices, cold_and_direct_at=0 if a.no_cold else 8)
    print(json.dumps(m))


if __name__ == "__main__":
    main()
