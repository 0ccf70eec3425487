#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/c3; mkdir -p $O
WL=cfg3 scripts/ab_quick.sh dflt= dev1=HIP_FORCE_DEV_KERNARG=1 dev0=HIP_FORCE_DEV_KERNARG=0 2>&1 | tee $O/ab_kernarg.txt
cd /tmp && export TMPDIR=/tmp
export CSV_BENCH_EXIT_ALARM=15
for v in 0 1; do
  rm -rf /tmp/f$v
  ( cd $R && HIP_FORCE_DEV_KERNARG=$v timeout -k 5 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/f$v -o pmc -- python bench.py --workload cfg3 --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2>&1 )
  echo "HIP_FORCE_DEV_KERNARG=$v"
  python - /tmp/f$v <<'PY'
import sqlite3, sys, glob
db = (glob.glob(sys.argv[1] + "/*.db") + glob.glob(sys.argv[1] + "/*/*.db"))[0]
cur = sqlite3.connect(db).cursor()
for name, n, avg in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name='FETCH_SIZE' group by kernel_name order by avg(value) desc"):
    if 'csv::' in name: print("  %-60s %5d %10.1f KiB" % (name[:60], n, avg))
PY
done
