"""experiment: one genome as ONE resident batch vs the same genome as G resident part-batches (whole chromosomes, balanced
by signatures) on G contexts of the same device, run concurrently.  Prints us per genome."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from cutesv_amd import engine, shard
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
store, params, _ = bench.make_workload(wl, 1.0, 0)
pst = store.pinned()


def timeit(ctxs, steps=300):
    for _ in range(20):
        for c in ctxs: c.run()
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c in ctxs: c.run()
    for c in ctxs: c.sync()
    return (time.perf_counter() - t0) / steps * 1e6


for G in (1, 2, 3, 4):
    ctxs = []
    for g in range(G):
        tasks = shard.tasks_of_rank(pst, g, G, genotype=params.genotype) if G > 1 else pst.tasks()
        c = engine.Context(0)
        c.upload(pst.host_batch(tasks, params), per_sig=False)
        ctxs.append(c)
    print("G=%d: %.1f us per genome" % (G, timeit(ctxs)), flush=True)
    for c in ctxs: c.close()
