#!/usr/bin/env python3
"""Kernels AND memory copies of the last few one-shot calls from a rocprofv3 --kernel-trace --memory-copy-trace database:
    python scripts/rocprof_oneshot_timeline.py <results.db> [calls]
start offset from the call's first event, duration, kind, name / bytes."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
ncalls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
ev = [(r[1], r[2], "K", r[0][:60]) for r in cur.execute("select name, start, end from kernels")]
mc = [v for v in views if "memory_cop" in v and "rocpd_" not in v]
if mc:
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % mc[0])]
    print("# %s columns: %s" % (mc[0], cols))
    size = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
    name = "name" if "name" in cols else cols[0]
    for r in cur.execute("select %s, start, end%s from %s" % (name, (", " + size) if size else "", mc[0])):
        ev.append((r[1], r[2], "C", "%s %s B" % (str(r[0])[:40], r[3] if size else "?")))
ev.sort()
heads = [i for i, e in enumerate(ev) if e[2] == "K" and "k_chain_count" in e[3]]
# a call = everything from the copies before its k_chain_count to the k_publish behind it
for h in heads[-ncalls:]:
    lo = h
    while lo > 0 and ev[lo - 1][0] > ev[h][0] - 600_000 and not (ev[lo - 1][2] == "K" and "k_publish" in ev[lo - 1][3]):
        lo -= 1
    hi = h
    while hi + 1 < len(ev) and not (ev[hi][2] == "K" and "k_publish" in ev[hi][3]):
        hi += 1
    t0 = ev[lo][0]
    print("---- call: %.1f us from first event to the end of k_publish" % ((ev[hi][1] - t0) / 1e3))
    for e in ev[lo:hi + 1]:
        print("%9.2f us  +%8.2f  %s  %s" % ((e[0] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
