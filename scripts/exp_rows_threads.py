"""row builder: text in T threads + objects in one pass; rows_ms per thread count (CSV_ROWS_THREADS)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cutesv_amd import engine, rows as rows_mod
store, params, _ = bench.make_workload(sys.argv[1] if len(sys.argv) > 1 else "cfg3", 1.0, 0)
hb = store.host_batch(store.tasks(), params)
ctx = engine.Context(0)
res = ctx.cluster_batch(hb)
ref = None
for T in (1, 4, 8, 16, 32, 64):
    os.environ["CSV_ROWS_THREADS"] = str(T)
    ts = []
    for _ in range(5):
        rows = seg = None
        t0 = time.perf_counter(); rows, seg = rows_mod.materialise(store, hb.segments, res); ts.append((time.perf_counter() - t0) * 1e3)
    if ref is None: ref = rows
    assert rows == ref
    print("threads %2d: rows_ms min %.2f median %.2f" % (T, min(ts), sorted(ts)[2]), flush=True)
