#!/usr/bin/env python3
"""Randomized campaign of the forked-pool path on the GPU box (not part of the test suite):
    python scripts/stress_pool.py [n_iterations] [first_seed]
random flags x random workload shapes, written as the reference's work directory, through the reference's phase-3 block with
the drop-in's callables under a forked Pool(T) (T random, ONE broker for the whole campaign: merged requests, shared walked
reads blocks, the reads filter), rows compared with the C oracle's rows of the same store.  Exits non-zero on a mismatch."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cutesv_amd import synth, resolve, broker, rows as rows_mod      # noqa: E402
from cutesv_amd.columns import Params, TYPES                          # noqa: E402
from oracle import oracle                                             # noqa: E402
from helpers import canonical_row                                     # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
os.environ["CUTESV_AMD_BROKER"] = "1"
os.environ["CUTESV_AMD_TRA_GT"] = "off"
resolve.warm_up()
bad = []
t0 = time.time()
for it in range(n_it):
    rng = np.random.default_rng(seed0 + it)
    gt = bool(rng.integers(0, 2))
    p = Params(min_support=int(rng.integers(1, 12)), min_size=int(rng.choice([0, 30, 500])), max_size=int(rng.choice([-1, 2000, 100000])), genotype=gt,
               max_cluster_bias_INS=int(rng.choice([20, 100, 1000, 5000])), diff_ratio_merging_INS=float(rng.choice([0.0, 0.3, 0.9, 2.0])),
               max_cluster_bias_DEL=int(rng.choice([20, 200, 1000, 5000])), diff_ratio_merging_DEL=float(rng.choice([0.0, 0.2, 0.5, 1.5])),
               max_cluster_bias_INV=int(rng.choice([10, 500, 5000])), max_cluster_bias_DUP=int(rng.choice([10, 500, 5000])),
               max_cluster_bias_TRA=int(rng.choice([5, 50, 2000])), diff_ratio_filtering_TRA=float(rng.choice([0.2, 0.6, 1.0])),
               remain_reads_ratio=float(rng.choice([0.3, 0.7, 1.0, 1.5])))
    st = synth.small_mixed(seed=seed0 + it, n_sites=int(rng.integers(5, 60)), coverage=int(rng.choice([4, 9, 20, 45, 90])), dup_frac=float(rng.choice([0.0, 0.1, 0.6])),
                           n_noise=int(rng.integers(0, 4000)), n_loci=int(rng.integers(0, 300)), contig_len=int(rng.choice([300_000, 2_000_000])),
                           pos_sigma=float(rng.choice([1.0, 12.0, 60.0])), len_sigma=float(rng.choice([0.003, 0.04, 0.2])), n_contigs=int(rng.integers(2, 6)))
    wd = tempfile.mkdtemp(prefix="csv_pool_") + "/"
    try:
        idx = st.write_reference_workdir(wd)
        if rng.integers(0, 3) == 0:
            st.save(wd + "cutesv_amd.cols")               # a third of the runs on the flat column directory
        resolve._stores.clear()
        T = int(rng.integers(1, 7))
        got = resolve.main_ctrl_phase3(wd, idx, p, T)
        tasks = st.tasks()
        hb = st.host_batch(tasks, p)
        per_seg = rows_mod.rows_by_segment(st, hb.segments, oracle.cluster_batch(hb, per_sig=False))
        want = {}
        for t in TYPES:
            for k, (tt, ch) in enumerate(tasks):
                if tt == t:
                    want.setdefault(ch, []).extend(per_seg[k])

        def canon(rows):
            return [canonical_row(r[1] if r[1] in ("DEL", "INS", "DUP", "INV") else "TRA", r) for r in rows]
        for ch in set(want) | set(got):
            a, b = canon(got.get(ch, [])), canon(want.get(ch, []))
            if a != b:
                raise AssertionError("chromosome %s: %d rows, the oracle has %d; first difference %r" % (ch, len(a), len(b), next(((x, y) for x, y in zip(a, b) if x != y), None)))
    except Exception as e:                                # noqa: BLE001
        bad.append((seed0 + it, repr(e)[:300]))
    finally:
        shutil.rmtree(wd, ignore_errors=True)
info = None
try:
    with broker.Client.connect(0, owner_pid=os.getpid(), spawn=False) as cl:
        info = cl.info()
        cl.shutdown()
except broker.BrokerError:
    pass
print("%d pool stages in %.1f s, %d failures; broker: %s" % (n_it, time.time() - t0, len(bad), {k: info.get(k) for k in ("calls", "batches", "merged_calls", "blocks", "block_hits")} if info else None))
for b in bad[:20]:
    print(b)
sys.exit(1 if bad else 0)
