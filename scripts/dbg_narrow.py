import sys, numpy as np
sys.path.insert(0, '.')
from cutesv_amd import synth, engine, _abi
from cutesv_amd.columns import Params
from oracle import oracle
st = synth.small_mixed(seed=21, genotype=True)
p = Params.ont(genotype=True)
wide = st.host_batch(st.tasks(), p)
pst = st.pinned()
narrow = pst.host_batch(pst.tasks(), p)
with engine.Context(0) as ctx:
    w = ctx.cluster_batch(wide, per_sig=True).trimmed()
    n = ctx.cluster_batch(narrow, per_sig=True).trimmed()
o = oracle.cluster_batch(wide, per_sig=True).trimmed()
for name, g in (("wide", w), ("narrow", n)):
    for f in ("bp1", "bp2", "support", "cipos", "cilen", "search_pos", "seq_pick", "dr", "dv"):
        if len(g[f]) != len(o[f]): print(name, f, "LEN", len(g[f]), len(o[f])); continue
        d = np.flatnonzero(g[f] != o[f])
        if len(d): print(name, f, "differs at", d[:10], "got", g[f][d[:10]], "want", o[f][d[:10]])
segs = wide.segments
for c in np.flatnonzero(n["bp1"] != o["bp1"])[:4]:
    k = o["call_seg"][c]; sg = segs[k]
    so = o["support_off"]; sup = o["support_sig"][so[c]:so[c+1]]
    print("call", c, "seg", k, "type", sg["svtype"], "bias", sg["max_cluster_bias"], "cluster", o["call_cluster"][c], "support", o["support"][c])
    cid = o["cluster_id"]; mem = np.flatnonzero(cid == o["call_cluster"][c])
    # cluster_id is in w space; map to global rows
    print(" members w:", mem[:70])
    woff = np.r_[0, np.cumsum(segs["sig_end"] - segs["sig_begin"])]
    rows = mem - woff[k] + sg["sig_begin"]
    print(" a:", wide.a[rows]); print(" b:", wide.b[rows]); print(" rid:", wide.read_id[rows]); print(" aux:", wide.aux[rows])
    print(" want bp1", o["bp1"][c], "narrow", n["bp1"][c], "wide", w["bp1"][c])
