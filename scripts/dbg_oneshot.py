"""debug aid: phases of the one-shot call (CSV_DEBUG_TIMING) and the device counters (CSV_DEBUG_COUNTERS)
    python scripts/dbg_oneshot.py [cfg3|cfg4|cfg5] [pinned]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSV_DEBUG_TIMING"] = "1"
os.environ["CSV_DEBUG_COUNTERS"] = "1"
import bench
from cutesv_amd import engine
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
st, p, name = bench.make_workload(wl, 1.0, 0)
ctx = engine.Context(0)
if "pinned" in sys.argv:
    st = st.pinned()
hb = st.host_batch(st.tasks(), p)
for i in range(4):
    t = time.perf_counter(); r = ctx.cluster_batch(hb, reuse=True); print("python side %.3f ms, reads mode %d" % ((time.perf_counter() - t) * 1e3, ctx.last_reads_mode()), flush=True)
