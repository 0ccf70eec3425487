"""GPU debugging aid: run a few small batches stage by stage (CSV_DEBUG=1 names the failing kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cutesv_amd import synth, engine, _abi
from cutesv_amd.columns import Params
from oracle import oracle

st = synth.small_mixed(seed=2026, n_sites=24)
ctx = engine.Context(0)
FIELDS = ("call_seg", "call_cluster", "bp1", "bp2", "support", "cipos", "cilen", "search_pos", "seq_pick",
          "dr", "dv", "gl_idx", "support_off", "support_sig", "cluster_id", "allele_id")
for name, types, gt in (("DEL", ("DEL",), False), ("INS", ("INS",), False), ("DUP", ("DUP",), False), ("INV", ("INV",), False),
                        ("TRA", ("TRA",), False), ("ALL", ("DEL", "INS", "INV", "DUP", "TRA"), False),
                        ("ALL+GT", ("DEL", "INS", "INV", "DUP", "TRA"), True)):
    print("==== case", name, flush=True)
    p = Params.ont(genotype=gt)
    hb = st.host_batch(st.tasks(types=types), p)
    got = ctx.cluster_batch(hb, per_sig=True).trimmed()
    want = oracle.cluster_batch(hb, per_sig=True).trimmed()
    print("  clusters", got["n_clusters"], want["n_clusters"], "calls", len(got["bp1"]), len(want["bp1"]), flush=True)
    for k in FIELDS:
        ok = len(got[k]) == len(want[k]) and np.array_equal(got[k], want[k])
        if not ok:
            bad = np.flatnonzero(np.asarray(got[k][:min(len(got[k]), len(want[k]))]) != np.asarray(want[k][:min(len(got[k]), len(want[k]))]))[:8]
            print("  MISMATCH", k, "lens", len(got[k]), len(want[k]), "first bad", bad, [ (int(got[k][i]), int(want[k][i])) for i in bad], flush=True)
print("done")
