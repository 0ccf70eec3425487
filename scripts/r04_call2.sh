#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/c2; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 600 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c2/bench_cfg3.json').read().strip().splitlines()[-1])
print("ms/step %.4f parity %s" % (d["ms_per_step"], d["parity_vs_oracle"]), {k[2:]: v for k, v in d["kernel_us"].items() if v})
PY
export CSV_PMC_ISOLATE=1
scripts/profile_gpu.sh cfg3 c2 > $O/profile.log 2>&1
P=$R/gpurun_out/prof
python scripts/rocprof_summary.py $(ls $P/c2_cfg3_kt/*.db | head -1) > $O/kt_cfg3_iso.txt
python scripts/rocprof_traffic.py $(ls $P/c2_cfg3_fetch/*.db | head -1) $(ls $P/c2_cfg3_write/*.db | head -1) $O/traffic_cfg3_iso.json > $O/pmc_cfg3_iso.txt
rm -rf $P/c2_cfg3_kt $P/c2_cfg3_fetch $P/c2_cfg3_write
cat $O/pmc_cfg3_iso.txt; grep -E "k_l2_wbinv|refine_indel|k_emit|chain_count" $O/kt_cfg3_iso.txt | head
