"""experiment: a batch with thousands of small segments (a reference with thousands of alt / decoy contigs): per-stage times"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cutesv_amd import engine
from cutesv_amd.columns import Params, SigStore
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
rng = np.random.default_rng(5)
per = {"DEL": [], "INS": []}
for c in range(nc):
    ch = "ctg%05d" % c
    for site in range(3):
        pos = 1000 + site * 5000
        for r in range(12):
            per["DEL"].append((pos + int(rng.integers(-5, 5)), 300 + int(rng.integers(-3, 3)), "rd%d_%d_%d" % (c, site, r), "DEL", ch))
            per["INS"].append((pos + 2000 + int(rng.integers(-5, 5)), 200 + int(rng.integers(-3, 3)), "ri%d_%d_%d" % (c, site, r), "ACGT" * 50, "INS", ch))
st = SigStore.from_tuple_lists(per, [])
p = Params.ont()
hb = st.host_batch(st.tasks(), p)
print("segments %d signatures %d" % (len(hb.segments), hb.n_sig), flush=True)
ctx = engine.Context(0)
import time
res = ctx.cluster_batch(hb, reuse=True)
for name, b in (("pageable", hb), ("pinned", st.pinned().host_batch(st.tasks(), p))):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); res = ctx.cluster_batch(b, reuse=True); ts.append((time.perf_counter() - t0) * 1e3)
    print("calls", res.n_calls, "one-shot ms (%s)" % name, ["%.2f" % t for t in ts])
ctx.upload(hb, per_sig=False)
for _ in range(5): ctx.run()
ctx.sync()
acc = [list(ctx.run(stats=True).ms_stage) for _ in range(20)]
med = np.median(np.array(acc), axis=0) * 1e3
print(" ".join("%s=%.1f" % (n.replace("k_", ""), v) for n, v in zip(engine.stage_names(), med) if v > 0.5))
