"""debug aid: the deep pile-up batch over and over, interleaved with other batches, every field against the oracle"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from cutesv_amd import synth, engine
from cutesv_amd.columns import Params
from oracle import oracle
from helpers import SOA_FIELDS
base = synth.small_mixed(seed=5, coverage=30, n_contigs=2, contig_len=1_000_000)
pile = synth.small_mixed(seed=6, coverage=12000, n_sites=3, n_contigs=2, contig_len=40_000, n_noise=0, n_loci=0, dup_frac=0.02)
st = synth.concat_stores(base, pile)
p = Params(genotype=True, min_support=10, max_cluster_bias_DEL=200)
hb = st.host_batch(st.tasks(), p)
want = oracle.cluster_batch(hb, per_sig=True).trimmed()
small = [synth.small_mixed(seed=s, genotype=bool(s % 2)) for s in range(4)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
DUMP = os.environ.get("CSV_DUMP_ITEMS")
def load_dump():
    raw = open(DUMP, "rb").read()
    ni, nbig, ntiny, ncalls = np.frombuffer(raw, np.int32, 4)
    o = 16; out = {"n": ni, "big": nbig, "tiny": ntiny, "calls": ncalls}
    for name, w, dt in (("rec", 16, np.int32), ("cnt", 8, np.int64), ("nslots", 4, np.int32), ("small", 16, np.int32), ("lbig", 4, np.int32), ("ltiny", 16, np.int32)):
        a = np.frombuffer(raw, dt, ni * w // np.dtype(dt).itemsize, o); o += ni * w
        out[name] = a.reshape(ni, -1) if w == 16 else a
    return out
good_dump = None
ctx = engine.Context(0)
for it in range(n):
    if it % 3 == 0:
        s2 = small[it % 4]
        ctx.cluster_batch(s2.host_batch(s2.tasks(), Params.ont(genotype=s2.reads_off is not None)), per_sig=bool(it % 2))
    if it % 50 == 49:
        ctx.close(); ctx = engine.Context(0)          # fresh arena
    got = ctx.cluster_batch(hb, per_sig=True).trimmed()
    diff = [f for f in SOA_FIELDS + ("support_sig", "cluster_id", "allele_id", "seg_status") if not np.array_equal(got[f], want[f])]
    if DUMP and not diff and good_dump is None:
        good_dump = load_dump()
    if DUMP and diff and good_dump is not None:
        d = load_dump()
        print("   dump: n", d["n"], good_dump["n"], "big", d["big"], good_dump["big"], "tiny", d["tiny"], good_dump["tiny"], "calls", d["calls"], good_dump["calls"])
        for name in ("rec", "cnt", "nslots"):
            if d["n"] == good_dump["n"]:
                x = np.flatnonzero((d[name] != good_dump[name]).reshape(d["n"], -1).any(axis=1))
                print("   dump diff", name, x[:8].tolist(), "bad", d[name][x[:4]].tolist(), "good", good_dump[name][x[:4]].tolist())
                for j in x[:3]:
                    sm = np.flatnonzero(d["small"][:d["n"] - d["big"] - d["tiny"], 0] == j); bg = np.flatnonzero(d["lbig"][:d["big"]] == j); ty = np.flatnonzero(d["ltiny"][:d["tiny"], 0] == j)
                    print("      item", j, "rec", d["rec"][j].tolist(), "type", st.tasks()[d["rec"][j][1]], "in small at", sm.tolist(), "big at", bg.tolist(), "tiny at", ty.tolist(),
                          "| good lists: small", np.flatnonzero(good_dump["small"][:good_dump["n"] - good_dump["big"] - good_dump["tiny"], 0] == j).tolist(), "big", np.flatnonzero(good_dump["lbig"][:good_dump["big"]] == j).tolist())
    if diff:
        bad += 1
        if len(got["bp1"]) != len(want["bp1"]):
            g = set(zip(got["call_seg"].tolist(), got["bp1"].tolist(), got["bp2"].tolist(), got["support"].tolist(), got["call_cluster"].tolist()))
            w = set(zip(want["call_seg"].tolist(), want["bp1"].tolist(), want["bp2"].tolist(), want["support"].tolist(), want["call_cluster"].tolist()))
            import collections
            cg = collections.Counter(zip(got["call_seg"].tolist(), got["call_cluster"].tolist(), got["bp1"].tolist()))
            cw = collections.Counter(zip(want["call_seg"].tolist(), want["call_cluster"].tolist(), want["bp1"].tolist()))
            for key in cg:
                if cg[key] != cw.get(key, 0):
                    idx = [i for i in range(len(got["bp1"])) if (got["call_seg"][i], got["call_cluster"][i], got["bp1"][i]) == key]
                    m = np.flatnonzero(want["cluster_id"] == key[1])
                    print("   duplicated call", key, "task", st.tasks()[key[0]], "x%d (oracle x%d) at call indices %s; cluster size %d; support_off %s; dr %s gl %s" % (
                        cg[key], cw.get(key, 0), idx, len(m), got["support_off"][idx[0]:idx[-1] + 2].tolist(), got["dr"][idx].tolist(), got["gl_idx"][idx].tolist()), flush=True)
                    for i in idx:
                        so = got["support_off"]
                        print("      call", i, "bp1", got["bp1"][i], "bp2", got["bp2"][i], "support", got["support"][i], "sigs", (got["support_sig"][so[i]:so[i + 1]] - m[0]).tolist() if len(m) else None, flush=True)
            for k, b1, b2, sp, cl in sorted(g - w)[:2]:
                m = np.flatnonzero(want["cluster_id"] == cl)
                print("   cluster", cl, "size", len(m), "a range", st.a[m].min() if len(m) else None, st.a[m].max() if len(m) else None, "b min", st.b[m].min() if len(m) else None, "seg_status", got["seg_status"].tolist(), flush=True)
        f = diff[0]
        idx = np.flatnonzero(np.asarray(got[f]) != np.asarray(want[f]))[:6] if len(got[f]) == len(want[f]) else "len"
        print("iteration %d: fields %s differ; %s at %s: got %s want %s" % (it, diff, f, idx, np.asarray(got[f])[idx] if not isinstance(idx, str) else len(got[f]),
                                                                          np.asarray(want[f])[idx] if not isinstance(idx, str) else len(want[f])), flush=True)
print("%d iterations, %d bad" % (n, bad))
