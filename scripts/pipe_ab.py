#!/usr/bin/env python3
"""Where a step of the pipelined resident loop goes (GPU box): python scripts/pipe_ab.py [cfg3] [steps]
host time inside csv_batch_run / csv_batch_publish_async / csv_batch_publish_wait per step, next to the serial loop and the
kernel-only loop; under `rocprofv3 --kernel-trace` scripts/pipe_timeline.py prints the kernels of the last steps."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from cutesv_amd import engine                             # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
store, params, name = bench.make_workload(wl, 1.0, 0)
pst = store.pinned()
phb = pst.host_batch(pst.tasks(), params)
ctx = engine.Context(0)
ctx.upload(phb)
ctx.run()
probe = ctx.download()
c32 = phb.a.dtype.itemsize == 4
bufs = [ctx.result_buffers(cap_calls=probe.n_calls + 64, cap_support=probe.n_support + 64, coord32=c32) for _ in range(2)]
for _ in range(5):
    ctx.run(); ctx.download(into=bufs[0])
t0 = time.perf_counter()
for _ in range(steps):
    ctx.run()
ctx.sync()
print("%s kernel-only       %.1f us/step" % (wl, (time.perf_counter() - t0) / steps * 1e6))
t0 = time.perf_counter()
for _ in range(steps):
    ctx.run(); ctx.download(into=bufs[0])
print("%s serial delivered  %.1f us/step" % (wl, (time.perf_counter() - t0) / steps * 1e6))
pc = time.perf_counter
for rep in range(2):
    tr = tp = tw = 0.0
    ctx.run(); ctx.publish_async(bufs[0])
    t0 = pc()
    for k in range(1, steps):
        a = pc(); ctx.run(); b = pc(); ctx.publish_async(bufs[k & 1]); c = pc(); ctx.publish_wait(); d = pc()
        tr += b - a; tp += c - b; tw += d - c
    ctx.publish_wait()
    dt = pc() - t0
    print("%s pipelined         %.1f us/step: host in run %.1f, publish_async %.1f, publish_wait %.1f" %
          (wl, dt / steps * 1e6, tr / steps * 1e6, tp / steps * 1e6, tw / steps * 1e6), flush=True)
ctx.close()
