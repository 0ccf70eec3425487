#!/bin/bash
# arbitrary counter sets, one rocprofv3 --pmc pass each (kernel trace only):  scripts/pmc_sets.sh cfg3 out.txt "A B C" "D E" ...
WL=$1; OUT=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CSV_BENCH_EXIT_ALARM=15
ARGS="bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline"
DBS=""; i=0
for set in "$@"; do
  i=$((i+1)); rm -rf /tmp/pmcs_$i
  ( cd $R && timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmcs_$i -o pmc -- python $ARGS > $O/pmcs_$i.log 2>&1 )
  echo "pass $i ($set) rc=$?"
  DBS="$DBS $(ls /tmp/pmcs_$i/*.db /tmp/pmcs_$i/*/*.db 2>/dev/null | head -1)"
done
python $R/scripts/rocprof_counters.py $DBS > $O/$OUT
cat $O/$OUT
