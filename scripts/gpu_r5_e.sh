#!/bin/bash
# genotype kernel off the scalar unit: parity first, then old vs new on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5e; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "genotyp or gt_ or reads or cover or golden or full" > $O/pytest_gt.log 2>&1
echo "pytest(gt) rc=$?"; tail -3 $O/pytest_gt.log
timeout 200 python scripts/stress_gpu.py 1500 5200000 > $O/stress.log 2>&1; tail -3 $O/stress.log
for wl in cfg5 cfg4; do
  for lib in build/lib_old.so cutesv_amd/libcutesv_hip.so build/lib_old.so cutesv_amd/libcutesv_hip.so; do
    CUTESV_AMD_LIB=$R/$lib timeout 300 python scripts/kernel_times.py $wl 60 2>&1 | tail -1
  done
done | tee $O/ab.txt
