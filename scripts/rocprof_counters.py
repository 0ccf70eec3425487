#!/usr/bin/env python3
"""rocprofv3 PMC passes (any counters, one or more rocpd databases) -> one table, per kernel and launch.

    python scripts/rocprof_counters.py [--json out.json] <pass1.db> [<pass2.db> ...] > profiles/<tag>_insts.txt

Per-launch averages of every counter found; the int64-column instantiations are kept apart like rocprof_traffic.py does.
Derived columns when their inputs are there: valu_per_wave = SQ_INSTS_VALU / SQ_WAVES;  valu_busy = SQ_ACTIVE_INST_VALU /
SQ_BUSY_CYCLES / 4 (four SIMDs per CU share one busy-cycle count: the fraction of the kernel's CU-cycles in which a SIMD
issued a vector instruction - a wave64 VALU instruction occupies its SIMD for 4 cycles)."""
import sqlite3
import sys

from rocprof_traffic import short


def main():
    table, counters, launches = {}, [], {}
    argv = sys.argv[1:]
    out_json = None
    if argv and argv[0] == "--json":
        out_json, argv = argv[1], argv[2:]
    for db in argv:
        cur = sqlite3.connect(db).cursor()
        for name, cname, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                               "group by kernel_name, counter_name"):
            k = short(name)
            if cname not in counters:
                counters.append(cname)
            row = table.setdefault(k, {})
            row[cname] = row.get(cname, 0.0) + avg           # (the two passes of k_genotype: summed)
            launches[k] = max(launches.get(k, 0), n)
    der = []
    if "SQ_INSTS_VALU" in counters and "SQ_WAVES" in counters:
        der.append("valu_per_wave")
    if "SQ_ACTIVE_INST_VALU" in counters and "SQ_BUSY_CYCLES" in counters:
        der.append("valu_busy")
    print("# per launch, averaged over the launches of the run")
    print("# %-26s %8s " % ("kernel", "launches") + " ".join("%18s" % c for c in counters + der))
    key = "SQ_INSTS_VALU" if "SQ_INSTS_VALU" in counters else counters[0]
    for k in sorted(table, key=lambda k: -table[k].get(key, 0.0)):
        r = table[k]
        d = []
        if "valu_per_wave" in der:
            d.append(r.get("SQ_INSTS_VALU", 0.0) / r["SQ_WAVES"] if r.get("SQ_WAVES") else 0.0)
        if "valu_busy" in der:
            # SQ_ACTIVE_INST_VALU: quad-cycles (4 clocks) in which a wavefront's VALU instruction occupies its SIMD, summed over
            # the chip; SQ_BUSY_CYCLES: clocks an SQ is busy, summed over the chip's 32 shader engines (8 XCDs x 4) - so
            # SQ_BUSY_CYCLES / 32 is the kernel's duration in clocks (checked against the trace: 1 006 854 / 32 = 31 464 clocks
            # = 13.1 us at 2.4 GHz for a 13.1 us kernel).  Busy share of the 1024 SIMDs = 4 * ACTIVE / (1024 * BUSY / 32)
            # = ACTIVE / (8 * BUSY): a pure counter ratio, no clock assumed.  (r04 divided by 4 instead of 8: the column read
            # 1.49 for a kernel at 0.74.)
            d.append(r.get("SQ_ACTIVE_INST_VALU", 0.0) / r["SQ_BUSY_CYCLES"] / 8.0 if r.get("SQ_BUSY_CYCLES") else 0.0)
        print("%-28s %8d " % (k, launches[k]) + " ".join("%18.1f" % r.get(c, 0.0) for c in counters) + " " + " ".join("%18.3f" % x for x in d))
    if out_json:
        import json
        with open(out_json, "w") as fh:
            json.dump({k: dict(table[k], launches=launches[k]) for k in table}, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
