#!/bin/bash
# what the driver runs at round end, in its order: smoke(), the default bench line, the bench line with explicit flags
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/dress; mkdir -p $O
T0=$(date +%s.%N); python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
T1=$(date +%s.%N); python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.err
python - <<'P'
import json
l=open('gpurun_out/dress/bench_default.json').read().strip().split('\n')
print(len(l), 'line(s),', len(l[-1]), 'bytes')
d=json.loads(l[-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(d['config']); print(d['roofline']); print(d['cpu_baseline'])
P
T2=$(date +%s.%N); python bench.py --gpus 1 --steps 30 --warmup 5 > $O/bench_flags.json 2>/dev/null; python -c "import json;d=json.loads(open('$O/bench_flags.json').read().strip().split('\n')[-1]);print(d['value'],d['ms_per_step'],d['steps'],d['warmup'])"
T3=$(date +%s.%N); python -c "print(\"smoke %.1f s, default bench %.1f s, flagged bench %.1f s\" % ($T1-$T0, $T2-$T1, $T3-$T2))"
