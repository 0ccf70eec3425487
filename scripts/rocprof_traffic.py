#!/usr/bin/env python3
"""PMC passes (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs) -> per-kernel HBM traffic per launch.

    python scripts/rocprof_traffic.py <fetch.db> <write.db> <out.json> > profiles/<tag>_pmc.txt

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes: the counters are in KiB;
on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read, so it is doubled.  That factor
is re-checked here on k_chain_count, whose only bulk traffic is one pass over the 8-byte position column
(2 x FETCH must equal 8 B x signatures); WRITE_SIZE is taken as is (re-checked on k_chain_apply, which writes
8 B per signature + 8 B per cluster).  For the gather-heavy kernels the doubled value is an upper bound.
"""
import json
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"k_refine<(\d+), (\d+)", name)
    if m:
        return "k_refine_block" if m.group(1) == "256" else ("k_refine_mid" if m.group(2) == "256" else "k_refine_wave")
    m = re.search(r"csv::(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, avg in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                                    "group by kernel_name", (counter,)):
        k = short(name)                                    # (the two k_genotype instantiations run once per step each: summed)
        out[k] = (max(n, out[k][0]), avg + out[k][1]) if k in out else (n, avg)
    return out


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
print("# per launch, averaged over the launches of the run; KiB as reported, bytes after the gfx950 FETCH x2 correction")
print("# %-24s %8s %14s %14s %16s" % ("kernel", "launches", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "traffic_bytes"))
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, (0, 0))[1] + write.get(k, (0, 0))[1])):
    f = fetch.get(k, (0, 0.0)); w = write.get(k, (0, 0.0))
    b = int(2 * f[1] * 1024 + w[1] * 1024)
    res[k] = b
    print("%-26s %8d %14.1f %14.1f %16d" % (k, max(f[0], w[0]), f[1], w[1], b))
with open(sys.argv[3], "w") as fh:
    json.dump(res, fh, indent=1, sort_keys=True)
