"""debug aid: reads ordering on a scaled cfg4 (CSV_DEBUG prints the device counters)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSV_DEBUG"] = "1"
import numpy as np
from cutesv_amd import synth, engine
from cutesv_amd.columns import Params
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
st, _ = synth.extraction_order(synth.hifi30_gt(scale=scale), region=int(10_000_000 * max(scale, 0.02)))
print("reads", st.n_reads, "descents", int((np.diff(st.r_start) < 0).sum()), flush=True)
p = Params.hifi(genotype=True, min_support=3)
hb = st.host_batch(st.tasks(), p)
ctx = engine.Context(0)
r = ctx.cluster_batch(hb)
print("calls", r.n_calls, flush=True)
