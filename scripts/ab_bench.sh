#!/bin/bash
# A/B timing of two builds of libcutesv_hip.so on the same box:  scripts/ab_bench.sh libA.so libB.so [workloads...]
# (box-to-box noise is several microseconds per kernel; compare only numbers from one call)
A=$1; B=$2; shift 2
WL=${@:-cfg3 cfg2 cfg5}
for rep in 1 2; do
  for w in $WL; do
    for lib in $A $B; do
      CUTESV_AMD_LIB=$PWD/$lib timeout 400 python bench.py --workload $w --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_us']; print('$w', '$lib', 'ms/step %.4f' % d['ms_per_step'], ' '.join('%s=%.1f' % (n[2:], v) for n, v in k.items() if v > 0 and n.startswith('k_')))"
    done
  done
done
