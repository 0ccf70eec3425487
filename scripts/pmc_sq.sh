#!/bin/bash
# SQ counters of one bench run (GPU box).  usage: scripts/pmc_sq.sh <tag> "<counter list>" [workload] [kernel-grep]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/prof; mkdir -p $OUT; cd $R
export CSV_BENCH_EXIT_ALARM=15
WL=${3:-cfg3}
timeout -k 5 150 rocprofv3 --kernel-trace --pmc $2 -d $OUT/$1 -o pmc -- python bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline > $OUT/$1.log 2>&1
echo "$1 rc=$?"
python - <<PY
import sqlite3,glob
db=sqlite3.connect(glob.glob("$OUT/$1/*.db")[0]); cur=db.cursor()
rows=list(cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"))
ks=sorted(set(r[0] for r in rows))
for k in ks:
    if "${4:-}" not in k: continue
    print(k[:70])
    for r in rows:
        if r[0]==k: print("    %-28s %16.1f" % (r[1], r[3]))
PY
