#!/usr/bin/env python3
"""A/B of the one-shot boundary call on the GPU box: python scripts/oneshot_ab.py [cfg3 cfg4 cfg5 ...] [only=label,label]
csv_cluster_batch from page-locked columns to page-locked results: bulk upload vs the gate-first form, full vs slim results."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                              # noqa: E402
from cutesv_amd import engine                             # noqa: E402

SLIM = dict(no_support=True, coord32=True, fields=("call_aux", "cipos", "cilen", "seq_pick", "dr", "gl_idx"))
ctx = engine.Context(0)
ONLY = [x[5:].split(",") for x in sys.argv[1:] if x.startswith("only=")]
ONLY = ONLY[0] if ONLY else None
for wl in ([x for x in sys.argv[1:] if not x.startswith("only=")] or ["cfg3"]):
    store, params, name = bench.make_workload(wl, 1.0, 0)
    pst = store.pinned()
    phb = pst.host_batch(pst.tasks(), params)
    n = phb.n_sig
    for label, env, kw in (("bulk/full/no-delta16", {"CSV_NO_LAZY": "1", "CSV_NO_DELTA16": "1"}, {}), ("bulk/full", {"CSV_NO_LAZY": "1"}, {}),
                           ("gate-first/full/no-delta16", {"CSV_NO_DELTA16": "1"}, {}), ("gate-first/full", {}, {}),
                           ("gate-first/slim/no-delta16", {"CSV_NO_DELTA16": "1"}, SLIM), ("gate-first/slim", {}, SLIM),
                           ("gate-first/full/no-peek", {"CSV_NO_PEEK": "1"}, {}), ("gate-first/full/no-rows8", {"CSV_NO_ROWS8": "1"}, {}),
                           ("gate-first/slim/no-rows8", {"CSV_NO_ROWS8": "1"}, SLIM), ("gate-first/full/no-reads-overlap", {"CSV_NO_READS_OVERLAP": "1"}, {}),
                           ("gate-first/slim", {}, SLIM), ("gate-first/full", {}, {})):
        if ONLY is not None and label not in ONLY:
            continue
        for k in ("CSV_NO_LAZY", "CSV_COPY_STREAM", "CSV_NO_DELTA16", "CSV_NO_PEEK", "CSV_NO_ROWS8", "CSV_NO_READS_OVERLAP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ctx._res_cache = None
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            r = ctx.cluster_batch(phb, reuse=True, **kw)
            ts.append((time.perf_counter() - t0) * 1e3)
        lazy, saved = ctx.lazy_info()
        ts = ts[2:]
        print("%s %-28s min %.3f med %.3f ms  (%.2e sig/s)  lazy=%d delta16=%d bytes_not_sent=%.1f MB calls=%d" %
              (wl, label, min(ts), float(np.median(ts)), n / (min(ts) * 1e-3), lazy, ctx.delta16_info(), saved / 1e6, r.n_calls), flush=True)
ctx.close()
