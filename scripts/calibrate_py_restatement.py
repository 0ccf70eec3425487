#!/usr/bin/env python3
"""Build-container only: time the REAL reference (imported from /root/reference with a pysam stub) against
oracle/py_restatement.py on the same workload, so the stand-in's fidelity as a timing proxy is on record.

    python scripts/calibrate_py_restatement.py [scale]        # default cfg3 at scale 0.25, 1 process
"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg                       # stubs pysam, imports the reference
from cutesv_amd import synth
from cutesv_amd.columns import Params
from oracle import py_restatement as pr

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
record = {"where": "build container (the reference cannot travel to the GPU box)", "scale": scale, "processes": 1, "cases": []}
for name, st, p in (("cfg3 ONT INS+DEL", synth.ont30(scale=scale), Params.ont()),
                    ("cfg4 HiFi --genotype", synth.hifi30_gt(scale=scale * 0.2), Params.hifi(genotype=True, min_support=3))):
    with tempfile.TemporaryDirectory() as d:
        d += "/"
        idx = mg.write_reference_workdir(st, d)
        t0 = time.perf_counter()
        n_ref = 0
        for t, ch in st.tasks():
            fn = {"DEL": mg.R_INDEL.run_del, "INS": mg.R_INDEL.run_ins}[t]
            bias = p.max_cluster_bias_DEL if t == "DEL" else p.max_cluster_bias_INS
            ratio = p.diff_ratio_merging_DEL if t == "DEL" else p.diff_ratio_merging_INS
            n_ref += len(fn((d, ch, t, p.min_support, ratio, bias, min(p.min_support, 5), "bam", p.genotype, p.gt_round,
                             p.remain_reads_ratio, idx))[1])
        t_ref = time.perf_counter() - t0
    tl = pr.tasks_from_store(st, p)
    t0 = time.perf_counter()
    res = pr.run_pool(tl, 1)
    t_py = time.perf_counter() - t0
    n_py = sum(len(r[1]) for r in res)
    print("%-22s sigs=%d reads=%d | reference %.2f s (%d rows, incl. its pickle.load) | py_restatement %.2f s (%d rows) | ratio restatement/reference = %.2f"
          % (name, st.n_sig, st.n_reads, t_ref, n_ref, t_py, n_py, t_py / t_ref))
    record["cases"].append({"workload": name, "signatures": int(st.n_sig), "reads": int(st.n_reads), "reference_s": round(t_ref, 3), "reference_rows": n_ref,
                            "py_restatement_s": round(t_py, 3), "py_restatement_rows": n_py, "ratio_restatement_over_reference": round(t_py / t_ref, 3)})
import json
record["note"] = ("ratio < 1: oracle/py_restatement.py is FASTER than the reference on the same input, so a speedup quoted against it "
                  "understates the speedup against cuteSV itself")
with open(os.path.join(ROOT, "profiles", "calibration_py_restatement.json"), "w") as f:
    json.dump(record, f, indent=1)
