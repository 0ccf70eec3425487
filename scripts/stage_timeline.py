"""Where a mode-1 stage's wall time goes: the reference's phase 3 with the drop-in under a forked Pool(T), every task's phases
(pickle -> columns, engine call, rows) on one clock (CUTESV_AMD_TIMELINE), next to the same pool running no-op tasks.

    python scripts/stage_timeline.py --workload cfg3 --workers 8,32 [--cols]
"""
import argparse
import glob
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                   # noqa: E402
import bench_stage                             # noqa: E402
from cutesv_amd import resolve                 # noqa: E402
from oracle import py_restatement as pr        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--workers", default="8,32")
    ap.add_argument("--cols", action="store_true", help="also write the flat cutesv_amd.cols directory (the build's own input format)")
    ap.add_argument("--all", action="store_true", help="every task of the last repetition, worker by worker, and the broker's counters for that stage")
    a = ap.parse_args()
    store, params, _ = bench.make_workload(a.workload, a.scale, 0)
    wd = tempfile.mkdtemp(prefix="cutesv_amd_tl_") + "/"
    idx = store.write_reference_workdir(wd)
    if a.cols:
        store.save(wd + "cutesv_amd.cols")
    os.environ["CUTESV_AMD_TRA_GT"] = "off"
    os.environ["CUTESV_AMD_BROKER"] = "1"
    resolve.warm_up()
    while bench_stage._broker_info() is None:
        time.sleep(0.02)
    n_tasks = sum(len(idx[t]) for t in ("DEL", "INS", "INV", "DUP", "TRA"))
    for T in [int(x) for x in a.workers.split(",")]:
        print("== T=%d: no-op pool of %d tasks %.1f ms" % (T, n_tasks, min(pr.pool_startup_seconds(T, n_tasks) for _ in range(3)) * 1e3))
        for rep in range(2):
            if rep == 0:
                # the first repetition of every T is the stage a cuteSV run has: nothing on the broker's shelf; the second one finds
                # every walked reads block there (what the tasks cost when nobody has to walk a block)
                try:
                    from cutesv_amd import broker
                    for d in bench_stage._devices():
                        with broker.Client.connect(d, owner_pid=os.getpid(), spawn=False) as cl:
                            cl.reads_flush()
                except Exception:                          # noqa: BLE001  (no broker to flush: the stage will say so itself)
                    pass
            tl = tempfile.mkdtemp(prefix="tl_")
            os.environ["CUTESV_AMD_TIMELINE"] = tl
            b0 = bench_stage._broker_info() or {}
            t0 = time.time()
            res = resolve.main_ctrl_phase3(wd, idx, params, T)
            t1 = time.time()
            del os.environ["CUTESV_AMD_TIMELINE"]
            rows = []
            for fn in glob.glob(tl + "/*.tl"):
                with open(fn) as f:
                    for line in f:
                        x = line.split()
                        rows.append((x[0], x[1], int(x[2]), int(x[3])) + tuple(float(v) - t0 for v in x[4:]) + (os.path.basename(fn),))
            rows.sort(key=lambda r: r[4])
            store_ms = sum(r[5] - r[4] for r in rows) * 1e3
            call_ms = sum(r[7] - r[6] for r in rows) * 1e3
            rows_ms = sum(r[8] - r[7] for r in rows) * 1e3
            print("   [%s] wall %.1f ms | first task starts %.1f, last ends %.1f | sum over tasks: store %.1f, call %.1f, rows %.1f ms | workers used %d"
                  % ("empty shelf" if rep == 0 else "blocks shared", (t1 - t0) * 1e3, rows[0][4] * 1e3, max(r[8] for r in rows) * 1e3, store_ms, call_ms, rows_ms, len({r[9] for r in rows})))
            big = sorted(rows, key=lambda r: -(r[8] - r[4]))[:4]
            for r in big:
                print("      %s %s n=%d reads=%d: start %.1f store %.1f call %.1f rows %.1f ms" % (r[0], r[1], r[2], r[3], r[4] * 1e3, (r[5] - r[4]) * 1e3, (r[7] - r[6]) * 1e3, (r[8] - r[7]) * 1e3))
            if a.all:
                b1 = bench_stage._broker_info() or {}
                print("      broker, this stage: %s" % {k: round(b1.get(k, 0) - b0.get(k, 0), 4) for k in ("calls", "batches", "merged_calls", "busy_s", "stage_in_s", "engine_s", "slice_out_s", "block_hits", "stage_grows")})
                for w in (sorted({r[9] for r in rows}) if rep == 1 else ()):
                    mine = [r for r in rows if r[9] == w]
                    print("      %s: %s" % (w, "  ".join("%s%s %.1f [s %.1f c %.1f r %.1f] -> %.1f" % (r[0][0], r[1], r[4] * 1e3, (r[5] - r[4]) * 1e3, (r[7] - r[6]) * 1e3,
                                                                                                        (r[8] - r[7]) * 1e3, r[8] * 1e3) for r in mine)))
    bench_stage._broker_info(shutdown=True)


if __name__ == "__main__":
    main()
