#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/c4; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
WL=cfg3 scripts/ab_quick.sh r03=CUTESV_AMD_LIB=$R/build/lib_old.so c2=CUTESV_AMD_LIB=$R/build/lib_c2.so new= 2>&1 | tee $O/ab_cfg3.txt
WL=cfg5 scripts/ab_quick.sh r03=CUTESV_AMD_LIB=$R/build/lib_old.so c2=CUTESV_AMD_LIB=$R/build/lib_c2.so new= 2>&1 | tee $O/ab_cfg5.txt
timeout 600 python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/c4/bench_cfg3.json').read().strip().splitlines()[-1])
print("ms/step %.4f parity %s" % (d["ms_per_step"], d["parity_vs_oracle"]), {k[2:]: v for k, v in d["kernel_us"].items() if v})
PY
scripts/profile_gpu.sh cfg3 c4 > $O/profile.log 2>&1
P=$R/gpurun_out/prof
python scripts/rocprof_summary.py $(ls $P/c4_cfg3_kt/*.db | head -1) > $O/kt_cfg3.txt
python scripts/rocprof_traffic.py $(ls $P/c4_cfg3_fetch/*.db | head -1) $(ls $P/c4_cfg3_write/*.db | head -1) $O/traffic_cfg3.json > $O/pmc_cfg3.txt
rm -rf $P/c4_cfg3_kt $P/c4_cfg3_fetch $P/c4_cfg3_write
cat $O/pmc_cfg3.txt; head -16 $O/kt_cfg3.txt
