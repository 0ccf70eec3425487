#!/usr/bin/env python3
"""known byte counts (build/pmc_calib) + the two rocprofv3 PMC databases -> per-pattern counter factors.

    python scripts/micro/pmc_calib_report.py known.json fetch.db write.db out.json

factor = counter bytes (KiB * 1024, as reported) / bytes the pattern really moves at the granularity that fits best;
rocprof_traffic.py divides a pipeline kernel's counters by the factor of its access pattern."""
import json
import sqlite3
import sys

known = json.load(open(sys.argv[1]))


def per_launch(db, counter):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    order = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else "rowid")
    out = {}
    for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name=? order by %s" % order, (counter,)):
        k = name.split("(")[0].split("::")[-1].strip()
        out.setdefault(k, []).append(val * 1024.0)
    return out


fetch, write = per_launch(sys.argv[2], "FETCH_SIZE"), per_launch(sys.argv[3], "WRITE_SIZE")
res = {}


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


print("# %-34s %14s %14s   known bytes -> ratio counter / known" % ("pattern", "FETCH_B", "WRITE_B"))
for k, kn in known.items():
    subs = {"": kn}
    if any(isinstance(v, dict) for v in kn.values()):
        subs = {s: dict(v, read_exact=kn.get("read_exact")) for s, v in kn.items() if isinstance(v, dict)}
    n_sub = len(subs)
    for i, (sname, kv) in enumerate(subs.items()):
        f = med(fetch.get(k, [])[i::n_sub]); w = med(write.get(k, [])[i::n_sub])
        ent = {"fetch_counter_bytes": f, "write_counter_bytes": w}
        line = "%-36s %14s %14s  " % (k + ("." + sname if sname else ""), "%.0f" % f if f is not None else "-", "%.0f" % w if w is not None else "-")
        for kk, vv in kv.items():
            if vv is None or not (kk.startswith("read_") or kk.startswith("write_")):
                continue
            c = f if kk.startswith("read_") else w
            if c is None or vv == 0:
                continue
            ent["ratio_" + kk] = c / vv
            ent[kk] = vv
            line += " %s=%d (x%.3f)" % (kk, vv, c / vv)
        res[k + ("." + sname if sname else "")] = ent
        print(line)
json.dump(res, open(sys.argv[4], "w"), indent=1, sort_keys=True)
