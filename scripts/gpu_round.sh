#!/bin/bash
# One gpurun call of the build -> measure loop: GPU tests, then the bench lines.  Everything lands in gpurun_out/round/.
#   scripts/gpu_round.sh [tests|notests] [workloads...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/round; mkdir -p $O
MODE=${1:-tests}; shift
WLS=${@:-cfg3}
if [ "$MODE" = tests ]; then
  timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
  echo "pytest rc=$?"; tail -15 $O/pytest.log
fi
for wl in $WLS; do
  extra=""; [ $wl != cfg3 ] && extra="--no-cpu-baseline"
  timeout -k 10 600 python bench.py --full --workload $wl $extra > $O/bench_$wl.json 2> $O/bench_$wl.err
  echo "bench $wl rc=$?"; tail -c 1500 $O/bench_$wl.err
  python - $O/bench_$wl.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line:", e); sys.exit(0)
print("ms/step %.4f value %.3g" % (d["ms_per_step"], d["value"]))
print("kernel_us", {k[2:]: v for k, v in d["kernel_us"].items() if k.startswith("k_") and v})
print("cold", {k[2:]: v for k, v in d["kernel_us_cold"].items() if k.startswith("k_") and v})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "kernel_us")}, "cold", d["roofline"]["cold"])
b = d["boundary"]; print("boundary", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in b.items() if k not in ("vcf_emit_native",)})
print("cpu", d.get("cpu_baseline") and d["cpu_baseline"]["wall_s"], "parity", d["parity_vs_oracle"])
PY
done
