#!/bin/bash
# GPU box: the default bench line, the neighbouring-step workloads and a 2-rank shard run -> gpurun_out/round/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/round; mkdir -p $O
run() { name=$1; shift; timeout -k 10 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; tail -c 500 $O/$name.err; }
run b3
run b_rebuild --workload rebuild --steps 5 --warmup 2
run b_extract --workload extract --steps 5 --warmup 2
run b_shard2 --gpus 2 --workload cfg4 --steps 5 --warmup 2 --scale 0.3
python - <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "round")
for n in ("b3", "b_rebuild", "b_extract", "b_shard2"):
    try:
        d = json.loads(open(os.path.join(O, n + ".json")).read().strip().splitlines()[-1])
    except Exception as e:
        print(n, "no line:", e); continue
    keep = {k: d.get(k) for k in ("metric", "value", "ms_per_step", "n_gpus", "scaling")}
    print(n, keep)
    print("   roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "kernel_us", "traffic", "algorithmic_bytes")} if d.get("roofline") else None)
    for k in ("chain", "split_reads", "cpu_baseline", "cpu_baseline_c", "cpu_baseline_c_mt", "shard_merge_equals_unsharded", "parity_vs_oracle", "ms_per_step_reads_order_kept"):
        if d.get(k) is not None:
            v = d[k]
            if isinstance(v, dict): v = {a: b for a, b in v.items() if a not in ("sample", "calibration_vs_reference")}
            print("   ", k, v)
    if d.get("boundary"):
        b = d["boundary"]; print("    boundary", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in b.items() if not isinstance(v, list)})
    if d.get("roofline_per_kernel"): print("    per kernel", d["roofline_per_kernel"])
PY
