#!/usr/bin/env python3
"""PMC passes (FETCH_SIZE, WRITE_SIZE: separate rocprofv3 runs) -> per-kernel L2 <-> fabric traffic per launch.

    python scripts/rocprof_traffic.py <fetch.db> <write.db> <out.json> > profiles/<tag>_pmc.txt

What the counters mean on gfx950, CALIBRATED in round 4 on known byte counts (scripts/micro/pmc_calib.hip, results in
profiles/r04_pmc_calibration.{json,txt}): every pattern the pipeline uses was run from cold L2s -
  * FETCH_SIZE (KiB) x 1024 x 2 = the bytes of the distinct 128-byte lines the kernel touches: coalesced 16 B and 4 B per
    lane streams (x0.501 of the exact bytes), the 32-byte-stride pairs of k_chain_count (x0.501), and the gathers of
    ~20-element runs from four columns of k_refine_indel_wave (x0.501 of the 128-byte-line count, 0.713 of the 64-byte
    count: the request is a whole line).  So 2 x FETCH is exact at line granularity for every read pattern here.
  * WRITE_SIZE (KiB) x 1024 = the bytes written at 32-byte granularity: exact for 4 / 16 B per lane streams, for 64-byte
    records (scattered or dense: x1.000), for 8-byte item words (x1.000), and for 8 B per lane runs at 8-byte alignment it is
    the count of distinct 32-byte sectors (x1.000 of that, 1.141 of the exact bytes).  Partial-line stores cause no fetch.
Both count requests that the 256 MiB Infinity Cache answers (MI355X_MICROARCH.md): "traffic" is L2 <-> fabric traffic, an
upper bound of HBM traffic - a kernel whose working set is Infinity-Cache resident can show more than 8 TB/s here.

Round-3 bug fixed here: the two instantiations of a kernel template over the column width (k_chain_count<true> / <false>,
k_refine_indel_wave<true> / <false>, ...) are ALTERNATIVES - a step runs one of them - but were summed like the two passes
of k_genotype, which doubled the reported traffic of exactly those kernels (r03: k_refine_indel_wave 57.9 MB, really 29).
They are now kept apart ("k_x" = the int32-column form bench.py times, "k_x[int64]" the other).
"""
import json
import re
import sqlite3
import sys


def short(name):
    wide = ""
    m = re.search(r"csv::(k_chain_count|k_refine_indel_wave|k_reads_runs|k_reads_plan|k_reads_gather|k_genotype_tra)<(true|false)>", name)
    if m:
        return m.group(1) + ("" if m.group(2) == "true" else "[int64]")
    m = re.search(r"csv::k_genotype<(\d+), (\d+), (true|false), (true|false)>", name)
    if m:       # first / second pass of one step: summed; column width kept apart
        return "k_genotype" + ("" if m.group(4) == "true" else "[int64]")
    m = re.search(r"k_refine<(\d+), (\d+)", name)
    if m:
        return "k_refine_block" if m.group(1) == "256" else ("k_refine_mid" if m.group(2) == "256" else "k_refine_wave")
    m = re.search(r"csv::(k_[a-z_0-9]+)", name)
    return (m.group(1) if m else name) + wide


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, avg in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                                    "group by kernel_name", (counter,)):
        k = short(name)                                    # (the two k_genotype passes run once per step each: summed)
        out[k] = (max(n, out[k][0]), avg + out[k][1]) if k in out else (n, avg)
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {}
    print("# per launch, averaged over the launches of the run.  read_B = FETCH_SIZE KiB x 1024 x 2 (bytes of the 128-byte lines fetched),")
    print("# write_B = WRITE_SIZE KiB x 1024 (32-byte sectors written); calibration: profiles/r04_pmc_calibration.txt")
    print("# %-30s %8s %14s %14s %14s %14s %16s" % ("kernel", "launches", "FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "read_B", "write_B", "traffic_bytes"))
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, (0, 0))[1] + write.get(k, (0, 0))[1])):
        f = fetch.get(k, (0, 0.0)); w = write.get(k, (0, 0.0))
        rb, wb = int(2 * f[1] * 1024), int(w[1] * 1024)
        res[k] = rb + wb
        res[k + ".read"] = rb
        res[k + ".write"] = wb
        print("%-32s %8d %14.1f %14.1f %14d %14d %16d" % (k, max(f[0], w[0]), f[1], w[1], rb, wb, rb + wb))
    with open(sys.argv[3], "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
