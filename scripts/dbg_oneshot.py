"""debug aid: phases of the one-shot call on cfg3 (CSV_DEBUG_TIMING)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSV_DEBUG_TIMING"] = "1"
from cutesv_amd import synth, engine
from cutesv_amd.columns import Params
st = synth.ont30(); p = Params.ont()
hb = st.host_batch(st.tasks(), p)
ctx = engine.Context(0)
for i in range(5):
    t = time.perf_counter(); r = ctx.cluster_batch(hb, reuse=True); print("python side %.3f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
