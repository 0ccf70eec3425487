#!/usr/bin/env python3
"""Timeline of the kernels of a few consecutive steps from a rocprofv3 kernel-trace database: start offset, duration, stream.

    python scripts/rocprof_timeline.py <results.db> [first kernel name substring] [step index] [steps]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "k_chain_count"
which = int(sys.argv[3]) if len(sys.argv) > 3 else 30
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = list(cur.execute("select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")))
idx = [i for i, r in enumerate(rows) if first in r[0]]
if len(idx) <= which + steps:
    which = max(0, len(idx) - steps - 1)
lo, hi = idx[which], idx[which + steps]
t0 = rows[lo][1]
for r in rows[lo:hi]:
    print("%9.2f us  +%7.2f  %s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, ("q%s" % r[3]) if qcol else "", r[0][:70]))
