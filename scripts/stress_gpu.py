#!/usr/bin/env python3
"""Long randomized parity campaign on the GPU box (not part of the test suite):
    python scripts/stress_gpu.py [n_iterations] [first_seed] [big]
random flags x random workload shapes (incl. TRA genotyping), every SoA field of the HIP path bit-exact against the
oracle.  Prints the failing seeds, exits non-zero if there are any."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cutesv_amd import synth, engine                      # noqa: E402
from cutesv_amd.columns import Params                     # noqa: E402
from oracle import oracle                                 # noqa: E402
from helpers import assert_soa_equal                      # noqa: E402

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"          # deep coverage / wide bias: the LDS and workgroup tiers
ctx = engine.Context(0)
bad = []
skipped = 0
t0 = time.time()
for it in range(n_it):
    rng = np.random.default_rng(seed0 + it)
    gt = bool(rng.integers(0, 2))
    p = Params(min_support=int(rng.integers(1, 14)), min_size=int(rng.choice([0, 30, 500])),
               max_size=int(rng.choice([-1, 2000, 100000])), genotype=gt, genotype_tra=bool(gt and rng.integers(0, 2)),
               gt_round=int(rng.choice([500, 40, 7])),
               max_cluster_bias_INS=int(rng.choice([0, 20, 100, 1000, 5000])), diff_ratio_merging_INS=float(rng.choice([0.0, 0.1, 0.3, 0.9, 2.0])),
               max_cluster_bias_DEL=int(rng.choice([0, 20, 200, 1000, 5000])), diff_ratio_merging_DEL=float(rng.choice([0.0, 0.2, 0.5, 1.5])),
               max_cluster_bias_INV=int(rng.choice([10, 500, 5000])), max_cluster_bias_DUP=int(rng.choice([10, 500, 5000])),
               max_cluster_bias_TRA=int(rng.choice([5, 50, 2000])), diff_ratio_filtering_TRA=float(rng.choice([0.2, 0.6, 1.0])),
               remain_reads_ratio=float(rng.choice([0.3, 0.7, 1.0, 1.5])))
    st = synth.small_mixed(seed=seed0 + it, n_sites=int(rng.integers(3, 12) if BIG else rng.integers(5, 80)), coverage=int(rng.choice([200, 400, 900]) if BIG else rng.choice([4, 9, 14, 20, 45, 90, 150])),
                           dup_frac=float(rng.choice([0.0, 0.1, 0.6])), n_noise=int(rng.integers(0, 6000)),
                           n_loci=int(rng.integers(0, 400)), contig_len=int(rng.choice([300_000, 2_000_000])),
                           pos_sigma=float(rng.choice([1.0, 12.0, 60.0])), len_sigma=float(rng.choice([0.003, 0.04, 0.2])),
                           n_contigs=int(rng.integers(2, 6)))
    tasks = st.tasks()
    if rng.integers(0, 3) == 0:
        tasks = [t for i, t in enumerate(tasks) if rng.integers(0, 2)] or tasks[:1]
    # reads table: sorted, in extraction order (whole runs move), or shuffled (general sort); TRA genotyping walks in
    # stable start order, so the oracle gets the same table
    mode = int(rng.integers(0, 3))
    if st.reads_off is not None and mode == 1:
        os.environ["CSV_READS_GAP"] = "30000"
        st, _ = synth.extraction_order(st, seed=it, region=int(rng.choice([100_000, 250_000])), workers=int(rng.integers(2, 9)))
    elif st.reads_off is not None and mode == 2:
        import dataclasses
        perm = np.arange(st.n_reads)
        for c in range(len(st.chroms)):
            lo, hi = int(st.reads_off[c]), int(st.reads_off[c + 1])
            perm[lo:hi] = lo + rng.permutation(hi - lo)
        st = dataclasses.replace(st, r_start=st.r_start[perm], r_end=st.r_end[perm], r_primary=st.r_primary[perm], r_id=st.r_id[perm])
    # (r06) the position column as 16-bit gaps for two pinned runs in three - whatever the share of escapes and the batch size -
    # and the same-run tier peek of the one-shot calls on or off
    os.environ["CSV_DELTA16_ESC"] = "0"
    os.environ["CSV_DELTA16_MIN"] = "0" if rng.integers(0, 3) else "1000000000"
    if rng.integers(0, 3) == 0:
        os.environ["CSV_NO_ROWS8"] = "1"                  # the gate-first fetch out of the two columns instead of the interleaved rows
    else:
        os.environ.pop("CSV_NO_ROWS8", None)
    if rng.integers(0, 4) == 0:
        os.environ["CSV_NO_PEEK"] = "1"
    else:
        os.environ.pop("CSV_NO_PEEK", None)
    if rng.integers(0, 2) == 0:
        st = st.pinned()                                  # page-locked columns, int32 twins of the positions / lengths (half of the runs)
    hb = st.host_batch(tasks, p)
    # one-shot calls from page-locked columns: the gate-first form (only the positions travel in bulk) for two runs in three
    os.environ["CSV_LAZY_MIN"] = "0" if rng.integers(0, 3) else "1000000000"
    try:
        want = oracle.cluster_batch(hb, per_sig=True).trimmed()
        if rng.integers(0, 2):                            # the one-shot call, results copied out or published in place ...
            got = ctx.cluster_batch(hb, per_sig=True, reuse=bool(rng.integers(0, 2))).trimmed()
            if rng.integers(0, 4) == 0 and hb.a.dtype == np.int32:       # ... and a slim result of the same batch (ABI v7)
                fields = tuple(f for f in ("call_aux", "cipos", "cilen", "seq_pick", "dr", "dv", "gl_idx", "search_pos", "call_cluster") if rng.integers(0, 2))
                slim = ctx.cluster_batch(hb, reuse=bool(rng.integers(0, 2)), no_support=bool(rng.integers(0, 2)), coord32=bool(rng.integers(0, 2)), fields=fields).trimmed()
                for name in ("call_seg", "bp1", "bp2", "support") + fields:
                    assert np.array_equal(slim[name].astype(np.int64), got[name].astype(np.int64)), "slim " + name
        else:                                             # ... or upload / run (twice: tiers on demand, idempotence) / download
            ctx.upload(hb, per_sig=True)
            ctx.run()
            if rng.integers(0, 2):
                ctx.sync()                                # (the first run's answers are on the host when the next is planned)
            ctx.run(); ctx.run()
            got = ctx.download(per_sig=True).trimmed()
            if rng.integers(0, 4) == 0 and not os.environ.get("CSV_STRESS_NO_PIPE"):      # ... and the pipelined delivery of two more runs
                ctx.upload(hb, per_sig=False)
                ctx.run(); ctx.download()
                blk = bool(rng.integers(0, 2))                # (one page-locked block: the copy engine delivers; scattered arrays: k_publish in place)
                bufs = [ctx.result_buffers(cap_calls=len(got["bp1"]) + 8, cap_support=len(got["support_sig"]) + 8, block=blk) for _ in range(2)]
                ctx.run(); ctx.publish_async(bufs[0]); ctx.run(); ctx.publish_async(bufs[1])
                for _ in range(2):
                    d = ctx.publish_wait().trimmed()
                    for name in ("call_seg", "bp1", "bp2", "support", "cipos", "cilen", "seq_pick", "dr", "dv", "gl_idx", "support_off", "support_sig"):
                        assert np.array_equal(np.asarray(d[name]).astype(np.int64), np.asarray(got[name]).astype(np.int64)), "pipelined " + name
        assert_soa_equal(got, want, store=st, set_order_segments=())
    except Exception as e:                                # noqa: BLE001
        bad.append((seed0 + it, repr(e)[:200]))
print("%d iterations in %.1f s, %d failures, %d over the cover-set limit" % (n_it, time.time() - t0, len(bad), skipped))
for b in bad[:20]:
    print(b)
sys.exit(1 if bad else 0)
