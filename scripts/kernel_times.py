#!/usr/bin/env python3
"""Per-kernel HIP-event times of one workload, nothing else (A/B of kernel variants on one box):
    [CUTESV_AMD_LIB=build/lib_x.so] python scripts/kernel_times.py [cfg3] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from cutesv_amd import engine, _abi                       # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
store, params, _ = bench.make_workload(wl, 1.0, 0)
pst = store.pinned()
phb = pst.host_batch(pst.tasks(), params)
ctx = engine.Context(0)
ctx.upload(phb, per_sig=False)
ctx.option(1, 0)
names = engine.stage_names()
for _ in range(5):
    ctx.run(stats=True)
acc = np.zeros(_abi.N_STAGES)
for _ in range(steps):
    acc += np.array(list(ctx.run(stats=True).ms_stage))
pk = bench.per_kernel_us(acc / steps, names)
ctx.run(); ctx.download()
for _ in range(4):
    ctx.run()
ctx.sync()
import time
t0 = time.perf_counter()
for _ in range(steps):
    ctx.run()
ctx.sync()
ko = (time.perf_counter() - t0) / steps * 1e3
print("%-12s kernel-only %.4f ms  " % (os.path.basename(os.environ.get("CUTESV_AMD_LIB", "default")), ko) +
      " ".join("%s=%.1f" % (k[2:], v) for k, v in pk.items() if k.startswith("k_") and isinstance(v, float) and v > 0.3))
ctx.close()
