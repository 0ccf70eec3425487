#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/c5; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; echo "bench rc=$?"; tail -5 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c5/bench_default.json').read().strip().splitlines()[-1])
print("value %.3g ms/step %.4f kernel_only %s parity %s" % (d["value"], d["ms_per_step"], d["kernel_only"], d["parity_vs_oracle"]))
print({k[2:]: v for k, v in d["kernel_us"].items() if k.startswith("k_") and v}, d["kernel_us"]["_event_floor"])
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "kernel_us", "traffic")})
print("speedups", d["speedups"])
for k, v in (d["other_workloads"] or {}).items():
    print(k, json.dumps(v)[:900])
PY
WL=cfg3 scripts/ab_quick.sh c2=CUTESV_AMD_LIB=$R/build/lib_c2.so new= 2>&1 | cut -c1-220
