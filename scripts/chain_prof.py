import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cutesv_amd import synth, engine, rebuild
from cutesv_amd.columns import TYPES
store, params, _ = bench.make_workload("cfg3", 1.0, 0)
per = synth.unsorted_rows(store, seed=1, dup_frac=0.05)
ctx = engine.Context(0)
def seg_of(t, ci, beg, end):
    rec = store.segment(t, store.chroms[ci], params).copy(); rec["sig_begin"], rec["sig_end"] = beg, end; return rec
for rep in range(3):
    t0=time.perf_counter()
    batch, tasks, src_row = rebuild.rebuild_to_device_batch(ctx, store.chroms, per, seg_of)
    t1=time.perf_counter()
    r2 = ctx.cluster_batch(batch)
    t2=time.perf_counter()
    print("rebuild_to_device_batch %.2f ms, cluster_batch(device columns) %.2f ms" % ((t1-t0)*1e3,(t2-t1)*1e3))
# pieces of rebuild_to_device_batch: the staging fill (numpy, page-locked destination), the library call, the segment records
chroms = store.chroms
live = [(ti, t) for ti, t in enumerate(TYPES) if t in per and len(per[t]["a"])]
n_rows = sum(len(per[t]["a"]) for _, t in live)
for rep in range(3):
    t0 = time.perf_counter()
    cat = rebuild._staging(ctx, n_rows)
    lo = 0
    for ti, t in live:
        d = per[t]; hi = lo + len(d["a"])
        np.take(np.arange(len(chroms), dtype=np.int32), np.asarray(d["chrom"]), out=cat["seg"][lo:hi], mode="clip")
        np.copyto(cat["a"][lo:hi], d["a"], casting="unsafe"); np.copyto(cat["b"][lo:hi], d["b"], casting="unsafe")
        np.copyto(cat["rid"][lo:hi], d["read_id"], casting="unsafe"); np.copyto(cat["aux"][lo:hi], d["aux"], casting="unsafe")
        lo = hi
    t1 = time.perf_counter()
    major = np.zeros(len(TYPES) * len(chroms), np.uint8)
    r = rebuild.rebuild_columns(ctx, cat["seg"], cat["a"], cat["b"], cat["rid"], cat["aux"], major, None, keep_on_device=True)
    t2 = time.perf_counter()
    segs = [seg_of(t, 0, 0, 1) for _ in range(48)]
    t3 = time.perf_counter()
    print("staging fill %.2f ms (%d rows, %.0f MB), rebuild_columns %.2f ms (device %.2f ms), 48 segment records %.2f ms; input dtypes %s" %
          ((t1 - t0) * 1e3, n_rows, n_rows * 28 / 1e6, (t2 - t1) * 1e3, r["ms_device"], (t3 - t2) * 1e3, {k: str(v.dtype) for k, v in per["DEL"].items() if hasattr(v, "dtype")}))
