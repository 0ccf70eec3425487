#!/usr/bin/env python3
"""Where k_genotype's wavefronts spend their cycles (a -DCSV_GT_PROF build of the library):
    CUTESV_AMD_LIB=build/lib_prof.so python scripts/gt_prof.py cfg5"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from cutesv_amd import engine                             # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
store, params, _ = bench.make_workload(wl, 1.0, 0)
pst = store.pinned()
ctx = engine.Context(0)
ctx.upload(pst.host_batch(pst.tasks(), params), per_sig=False)
for _ in range(4):
    ctx.run()
    ctx.sync()
