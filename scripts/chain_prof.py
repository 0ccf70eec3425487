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
# pieces
chroms=store.chroms
t0=time.perf_counter()
order = sorted(range(len(chroms)), key=lambda i: chroms[i]); crank=np.zeros(len(chroms),np.int64); crank[order]=np.arange(len(chroms))
cols = {k: [] for k in ("seg","a","b","rid","aux")}
for ti,t in enumerate(TYPES):
    if t not in per or len(per[t]["a"])==0: continue
    d=per[t]
    cols["seg"].append(ti*len(chroms)+crank[np.asarray(d["chrom"],np.int64)])
    cols["a"].append(np.asarray(d["a"],np.int64)); cols["b"].append(np.asarray(d["b"],np.int64))
    cols["rid"].append(np.asarray(d["read_id"],np.int64)); cols["aux"].append(np.asarray(d["aux"],np.int64))
t1=time.perf_counter()
cat={k: np.concatenate(v) for k,v in cols.items()}
t2=time.perf_counter()
major=np.zeros(len(TYPES)*len(chroms),np.uint8)
r=rebuild.rebuild_columns(ctx, cat["seg"],cat["a"],cat["b"],cat["rid"],cat["aux"],major,None,keep_on_device=True)
t3=time.perf_counter()
print("per-type numpy %.2f ms, concatenate %.2f ms, rebuild_columns %.2f ms (device %.2f ms) dtypes %s" % ((t1-t0)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,r["ms_device"], {k:str(v.dtype) for k,v in per["DEL"].items() if hasattr(v,'dtype')}))
