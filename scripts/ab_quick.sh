#!/bin/bash
# quick A/B on one box: scripts/ab_quick.sh "<label>=<env assignments or lib path>" ...   (cfg3, 200 steps, kernel_us line)
#   e.g. scripts/ab_quick.sh base= w4=CUTESV_AMD_LIB=build/lib_w4.so g4096=CSV_IW_GRID=4096
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
WL=${WL:-cfg3}
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  line=$(env $envs timeout 300 python bench.py --workload $WL --steps 200 --warmup 10 --no-cpu-baseline --full 2>/dev/null | tail -1)
  echo "$line" | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_us']
print('%-10s ms/step %.4f kernels %.4f  ' % ('$label', d['ms_per_step'], (d.get('kernel_only') or {}).get('ms_per_step', 0)) + ' '.join('%s=%.1f' % (n[2:], v) for n, v in k.items() if n.startswith('k_') and v > 0) + '  one_shot %.3f stage %.2f rows %.2f' % (d['boundary']['one_shot_call_ms'], d['boundary']['stage_wall_ms'], d['boundary']['rows_ms']))"
done
done
