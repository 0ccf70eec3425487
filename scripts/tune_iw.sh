#!/bin/bash
# tuning aid (GPU box): occupancy target x grid size of k_refine_indel_wave
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in 4 5 6 8; do
  make -s -C cutesv_amd/csrc clean >/dev/null
  make -s -C cutesv_amd/csrc FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wno-unused-value -DCSV_IW_WAVES=$w" 2>&1 | grep -E "error" 
  for g in 1024 2048 4096 8192; do
    CSV_IW_GRID=$g timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves=$w grid=$g', 'ms/step %.4f'%d['ms_per_step'], 'indel_wave_us', d['kernel_us']['k_refine_indel_wave'], 'parity', d['parity_vs_oracle'])"
  done
done
