import sys, os, json, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, bench_stage
from cutesv_amd import resolve
store, params, _ = bench.make_workload("cfg4", 1.0, 0)
import tempfile
wd = tempfile.mkdtemp() + "/"
idx = store.write_reference_workdir(wd)
store.save(wd + "cutesv_amd.cols")
os.environ["CUTESV_AMD_BROKER"] = "1"; os.environ["CUTESV_AMD_TRA_GT"] = "off"
resolve.warm_up()
while bench_stage._broker_info() is None: time.sleep(0.02)
for T in (1, 8, 8, 8):
    i0 = bench_stage._broker_info()
    dt, res = bench_stage._stage(wd, idx, params, T)
    i1 = bench_stage._broker_info()
    print("T=%d wall %.1f ms | broker: busy %.1f ms (stage_in %.1f, engine %.1f, slice_out %.1f), %d calls in %d batches" % (T, dt * 1e3, (i1["busy_s"] - i0["busy_s"]) * 1e3,
          (i1.get("stage_in_s", 0) - i0.get("stage_in_s", 0)) * 1e3, (i1.get("engine_s", 0) - i0.get("engine_s", 0)) * 1e3, (i1.get("slice_out_s", 0) - i0.get("slice_out_s", 0)) * 1e3,
          i1["calls"] - i0["calls"], i1["batches"] - i0["batches"]))
bench_stage._broker_info(shutdown=True)
