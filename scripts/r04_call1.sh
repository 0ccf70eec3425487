#!/bin/bash
# round 4, GPU call 1: tests on the re-laid refine -> emit hand-off, A/B against the r03 library, PMC calibration, cfg3 PMC
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/c1; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
scripts/micro/pmc_calib.sh > $O/calib.log 2>&1; tail -30 $O/calib.log
WL=cfg3 scripts/ab_quick.sh old=CUTESV_AMD_LIB=$R/build/lib_old.so new= w5=CUTESV_AMD_LIB=$R/build/lib_w5.so 2>&1 | tee $O/ab_cfg3.txt
WL=cfg5 scripts/ab_quick.sh old=CUTESV_AMD_LIB=$R/build/lib_old.so new= w5=CUTESV_AMD_LIB=$R/build/lib_w5.so 2>&1 | tee $O/ab_cfg5.txt
scripts/profile_gpu.sh cfg3 c1 > $O/profile.log 2>&1
P=$R/gpurun_out/prof
python scripts/rocprof_summary.py $(ls $P/c1_cfg3_kt/*.db | head -1) > $O/kt_cfg3.txt
python scripts/rocprof_traffic.py $(ls $P/c1_cfg3_fetch/*.db | head -1) $(ls $P/c1_cfg3_write/*.db | head -1) $O/traffic_cfg3.json > $O/pmc_cfg3.txt
rm -rf $P/c1_cfg3_kt $P/c1_cfg3_fetch $P/c1_cfg3_write
head -30 $O/kt_cfg3.txt; cat $O/pmc_cfg3.txt
