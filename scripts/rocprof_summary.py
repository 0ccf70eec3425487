#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (ROCm 7.2 default output) into the text summary committed under profiles/.

    python scripts/rocprof_summary.py gpurun_out/prof/kt/kt_results.db > profiles/r01_kernel_trace_cfg3.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary (from %s)" % sys.argv[1])
print("# %-62s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print("%-64s %8d %14.3f %12.3f %8.2f" % (name[:64], calls, total, avg, pct))
try:
    import statistics
    groups = {}
    for r in cur.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count, duration from kernels"):
        groups.setdefault(tuple(r[:8]), []).append(r[8] / 1000.0)
    rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]) / len(kv[1]))
    # (median: the bench command also runs a few steps with the caches flushed, whose launches sit in the average)
    print("\n# per launch shape: kernel grid wg lds scratch vgpr agpr sgpr calls avg_us median_us min_us max_us")
    for k, d in rows:
        print("%-64s %9d %5d %7d %7d %4d %4d %4d %6d %10.3f %10.3f %10.3f %10.3f" % ((k[0][:64],) + tuple(k[1:]) + (len(d), sum(d) / len(d), statistics.median(d), min(d), max(d))))
except Exception as e:          # view layout differs between versions
    print("# (per-launch view unavailable: %s)" % e)
try:
    rows = list(cur.execute("select name, counter_name, count(*), avg(value) from counters_collection group by name, counter_name"))
    if rows:
        print("\n# PMC: kernel counter launches avg_value")
        for r in rows:
            print("%-64s %-24s %6d %18.1f" % (r[0][:64], r[1], r[2], r[3]))
except Exception as e:
    pass
