#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/c6; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; tail -5 $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c6/bench_default.json').read().strip().splitlines()[-1])
print("value %.3g ms/step %.4f kernel_only %.4f parity %s" % (d["value"], d["ms_per_step"], d["kernel_only"]["ms_per_step"], d["parity_vs_oracle"]))
b = d["boundary"]
print("vcf", json.dumps(b["vcf_emit_native"])[:700])
print("task", json.dumps(b["per_task_drop_in"])[:400])
print("stage_wall_ms", b["stage_wall_ms"], "rows_ms", b["rows_ms"], "one_shot", b["one_shot_call_ms"])
for k, v in (d["other_workloads"] or {}).items():
    print(k, v.get("error") or (v["ms_per_step"], v["kernel_only_ms_per_step"], v["parity_vs_oracle"], json.dumps(v.get("vcf_emit_native"))[:500]))
PY
