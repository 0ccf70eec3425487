"""GPU box: time the GPU rebuild (csv_rebuild_signatures) on a shuffled cfg-3 genome against numpy's lexsort."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cutesv_amd import synth, engine, rebuild

st = synth.ont30()
ctx = engine.Context(0)
rng = np.random.default_rng(1)
per = {}
for (t, ch), (b, e) in st.seg_index.items():
    d = per.setdefault(t, dict(chrom=[], a=[], b=[], read_id=[], aux=[]))
    d["chrom"].append(np.full(e - b, st.chroms.index(ch))); d["a"].append(st.a[b:e]); d["b"].append(st.b[b:e])
    d["read_id"].append(st.read_id[b:e]); d["aux"].append(st.aux[b:e] if t == "INS" else np.zeros(e - b, np.int32))
for t, d in per.items():
    cols = {k: np.concatenate(v) for k, v in d.items()}
    perm = rng.permutation(len(cols["a"]))
    per[t] = {k: v[perm] for k, v in cols.items()}
for it in range(3):
    t0 = time.perf_counter()
    got, info = rebuild.store_from_unsorted(ctx, st.chroms, per)
    dt = time.perf_counter() - t0
    print("rebuild: %d rows, %d passes, device %.3f ms, wall (H2D + sort + D2H + python) %.1f ms" % (got.n_sig, info["n_passes"], info["ms_device"], dt * 1e3))
# segments come out in the reference's order (chromosome NAMES sorted as strings); compare per segment
for key, (b, e) in st.seg_index.items():
    gb, ge = got.seg_index[key]
    assert ge - gb == e - b and np.array_equal(got.a[gb:ge], st.a[b:e]) and np.array_equal(got.b[gb:ge], st.b[b:e]) \
        and np.array_equal(got.read_id[gb:ge], st.read_id[b:e]), key
print("rebuilt store == original store, segment by segment")
t0 = time.perf_counter()
for t, d in per.items():
    o = np.lexsort((d["read_id"], d["b"], d["a"], d["chrom"]))
dt = time.perf_counter() - t0
print("numpy lexsort of the same rows (single thread): %.1f ms" % (dt * 1e3))
