#!/usr/bin/env python3
"""Kernels of the last steps of a pipelined loop from a rocprofv3 --kernel-trace database:
    python scripts/pipe_timeline.py <results.db> [n_publishes]
start offset, duration, queue, name - the run of step k + 1 should sit beside the k_publish of step k."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
q = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
print("# kernels columns:", cols)
ev = [(r[1], r[2], r[3] if q else 0, r[0][:48]) for r in cur.execute("select name, start, end%s from kernels" % ((", " + q) if q else ""))]
ev.sort()
pubs = [i for i, e in enumerate(ev) if "k_publish" in e[3]]
lo = pubs[-n - 1] if len(pubs) > n else 0
t0 = ev[lo][0]
for e in ev[lo:]:
    print("%9.2f us  +%8.2f  q%-3s %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
