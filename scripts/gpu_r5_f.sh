#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5f; mkdir -p $O
for wl in cfg5 cfg4; do
  for g in 1024 2048 3072 4096 6144 8192 16384 32768; do
    echo -n "grid $g: "; CSV_GT_GRID=$g timeout 300 python scripts/kernel_times.py $wl 40 2>&1 | tail -1 | sed 's/chain_count.*reads_maxlen=[0-9.]* //'
  done
done | tee $O/grid.txt
