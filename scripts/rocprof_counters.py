#!/usr/bin/env python3
"""one rocprofv3 --pmc pass (rocpd database) -> per-kernel averages per launch of every counter collected

    python scripts/rocprof_counters.py <pmc.db> [<pmc2.db> ...] > profiles/<tag>_insts.txt
"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"k_refine<(\d+), (\d+)", name)
    if m:
        return "k_refine_block" if m.group(1) == "256" else ("k_refine_mid" if m.group(2) == "256" else "k_refine_wave")
    m = re.search(r"k_genotype<(\d+)", name)
    if m:
        return "k_genotype_%s" % m.group(1)
    m = re.search(r"csv::(k_[a-z_0-9]+)", name)
    return m.group(1) if m else name[:40]


tab, names, launches = {}, [], {}
for db in sys.argv[1:]:
    cur = sqlite3.connect(db).cursor()
    for kname, cname, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"):
        k = short(kname)
        if not k.startswith("k_"):
            continue
        tab.setdefault(k, {})[cname] = avg
        launches[k] = n
        if cname not in names:
            names.append(cname)
print("# per launch, averaged over the launches of the run")
print("# %-22s %8s " % ("kernel", "launches") + " ".join("%16s" % c for c in names))
for k in sorted(tab, key=lambda k: -tab[k].get(names[0], 0)):
    print("%-24s %8d " % (k, launches[k]) + " ".join("%16.1f" % tab[k].get(c, float("nan")) for c in names))
