#!/usr/bin/env python3
"""Kernels and copies of the last steps of a pipelined loop from a rocprofv3 --kernel-trace [--memory-copy-trace] database:
    python scripts/pipe_timeline.py <results.db> [n_deliveries]
start offset, duration, queue, name - the run of step k + 1 should sit beside the delivery of step k."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
ev = [(r[1], r[2], "q%s" % r[3], r[0][:56]) for r in cur.execute("select name, start, end, queue_id from kernels")]
mc = [v for v in views if "memory_cop" in v and "rocpd_" not in v]
if mc:
    for r in cur.execute("select name, start, end, size from %s" % mc[0]):
        ev.append((r[1], r[2], "dma", "%s %d B" % (str(r[0])[:32], r[3])))
ev.sort()
marks = [i for i, e in enumerate(ev) if "k_publish" in e[3]]
lo = marks[-n - 1] if len(marks) > n else 0
while lo > 0 and "k_chain_count" not in ev[lo][3]:
    lo -= 1
t0 = ev[lo][0]
for e in ev[lo:]:
    print("%9.2f us  +%8.2f  %-4s %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
