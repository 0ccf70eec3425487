"""per-stage event times of the resident step, nothing else (for timing experiments with ablated libraries):
   [CUTESV_AMD_LIB=build/lib_x.so] python scripts/stage_times.py [workload] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from cutesv_amd import engine, _abi
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
store, params, _ = bench.make_workload(wl, 1.0, 0)
hb = store.pinned().host_batch(store.tasks(), params)
ctx = engine.Context(0)
ctx.upload(hb, per_sig=False)
ctx.option(1, int(os.environ.get('REUSE_READS', '0')))
for _ in range(10):
    ctx.run()
ctx.sync()
import time
t0 = time.perf_counter()
for _ in range(steps):
    ctx.run()
ctx.sync()
dt = (time.perf_counter() - t0) / steps
acc = []
for _ in range(50):
    acc.append(list(ctx.run(stats=True).ms_stage))
med = np.median(np.array(acc), axis=0) * 1e3
names = engine.stage_names()
print("%-10s us/step %.2f  " % (os.environ.get("LABEL", ""), dt * 1e6) + " ".join("%s=%.1f" % (n.replace("k_", ""), v) for n, v in zip(names, med) if v > 0.5))
