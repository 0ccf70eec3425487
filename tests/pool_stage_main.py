"""Run in a FRESH interpreter by tests/test_pool_dropin.py (`-m gpu`) and scripts: the reference's phase 3
(main script :1113-1199) with cutesv_amd.resolve's five callables under a forked Pool, on the reference's pickle layout.

    python tests/pool_stage_main.py --cfg cfg3_s025 --threads 4 --mode broker|warm|direct --out result.json --work DIR

Writes what the test asserts on: per-(type, chromosome) digests of the rows, whether this (parent) process ever loaded the
HIP runtime, which engine each worker held and in which process, the broker's own report, the swallowed error count.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from cutesv_amd import resolve, synth, broker          # noqa: E402
from cutesv_amd.columns import Params                  # noqa: E402

PROBE_DIR = None


def _probe(fn, args):
    out = fn(args)
    with open(os.path.join(PROBE_DIR, "w%d" % os.getpid()), "w") as f:      # (one small file per worker, rewritten per task)
        json.dump(dict(pid=os.getpid(), kind=type(resolve._ctx).__name__, ctx_pid=resolve._ctx_pid,
                       hip=hip_loaded()), f)
    return out


def run_del(a):
    return _probe(resolve.run_del, a)


def run_ins(a):
    return _probe(resolve.run_ins, a)


def run_inv(a):
    return _probe(resolve.run_inv, a)


def run_dup(a):
    return _probe(resolve.run_dup, a)


def run_tra(a):
    return _probe(resolve.run_tra, a)


def hip_loaded():
    with open("/proc/self/maps") as f:
        m = f.read()
    return "libamdhip64" in m or "libcutesv_hip" in m


def workload(cfg):
    import helpers
    d = helpers.load_json("digests.json")[cfg]
    st = {"cfg3_s025": lambda: synth.ont30(scale=0.25), "cfg4_s002": lambda: synth.hifi30_gt(scale=0.02),
          "cfg5_s002": lambda: synth.ont90_all(scale=0.02)}[cfg]()
    return st, Params(**d["params"])


def main():
    global PROBE_DIR
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", default="cfg3_s025")
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--mode", default="broker", choices=["broker", "warm", "direct"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--work", required=True)
    a = ap.parse_args()
    import helpers
    st, p = workload(a.cfg)
    wd = os.path.join(a.work, "wd_%s_%d_%s" % (a.cfg, a.threads, a.mode)) + "/"
    os.makedirs(wd, exist_ok=True)
    PROBE_DIR = os.path.join(wd, "probe")
    os.makedirs(PROBE_DIR, exist_ok=True)
    idx = st.write_reference_workdir(wd)
    n_tasks = sum(len(idx[t]) for t in ("DEL", "INS", "INV", "DUP", "TRA"))
    idx["DEL"]["zz_bad"] = os.path.getsize(wd + "DEL.pickle") + 1234       # a task that raises inside its worker
    os.environ["CUTESV_AMD_TRA_GT"] = "off"
    os.environ["CUTESV_AMD_BROKER"] = "0" if a.mode == "direct" else "1"
    t0 = time.perf_counter()
    if a.mode == "warm":
        resolve.warm_up()
    t_warm = time.perf_counter() - t0
    errors = []
    t0 = time.perf_counter()
    results = resolve.main_ctrl_phase3(wd, idx, p, a.threads, fns=dict(DEL=run_del, INS=run_ins, INV=run_inv, DUP=run_dup, TRA=run_tra),
                                       on_error=errors.append)
    wall = time.perf_counter() - t0
    # rows back into (type, chromosome) groups
    per = {}
    for ch, rows in results.items():
        for r in rows:
            t = r[1] if r[1] in ("DEL", "INS", "DUP", "INV") else "TRA"
            per.setdefault("%s:%s" % (t, ch), []).append(r)
    digests = {k: [len(v), helpers.digest(k.split(":")[0], v)] for k, v in per.items()}
    probes = []
    for fn in os.listdir(PROBE_DIR):
        with open(os.path.join(PROBE_DIR, fn)) as f:
            probes.append(json.load(f))
    info = None
    if a.mode != "direct":
        with broker.Client.connect(0, owner_pid=os.getpid(), spawn=False) as cl:
            info = cl.info()
            cl.shutdown()
        t_end = time.monotonic() + 20
        while broker._try_connect(broker.socket_name(os.getpid(), 0)) is not None and time.monotonic() < t_end:
            time.sleep(0.05)
    leftover = 0 if broker._try_connect(broker.socket_name(os.getpid(), 0)) is None else 1
    out = dict(cfg=a.cfg, threads=a.threads, mode=a.mode, tasks=n_tasks, signatures=st.n_sig, wall_s=wall, warm_up_s=t_warm,
               digests=digests, errors=len(errors), error_text=[repr(e)[:200] for e in errors], bad_task_rows=len(results.get("zz_bad", [])),
               parent_pid=os.getpid(), parent_loaded_hip_library=hip_loaded(), parent_has_context=resolve._ctx is not None,
               worker_pids=[q["pid"] for q in probes], worker_engine_kinds=[q["kind"] for q in probes],
               worker_context_pids=[q["ctx_pid"] for q in probes if q["kind"] == "Context"],
               workers_loaded_hip=[q["hip"] for q in probes], broker=info, leftover_brokers=leftover)
    with open(a.out, "w") as f:
        json.dump(out, f)
    print(json.dumps({k: v for k, v in out.items() if k not in ("digests",)}))


if __name__ == "__main__":
    main()
