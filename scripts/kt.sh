#!/bin/bash
# rocprofv3 kernel trace of one bench workload -> gpurun_out/round/kt_<wl>.txt   (scripts/kt.sh cfg3 [steps])
WL=${1:-cfg3}; STEPS=${2:-10}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$WL
export CSV_BENCH_EXIT_ALARM=15
( cd $R && timeout -k 5 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_$WL -o kt -- python bench.py --workload $WL --steps $STEPS --warmup 2 --no-cpu-baseline > $O/kt_$WL.log 2>&1 )
echo "kt $WL rc=$?"
python $R/scripts/rocprof_summary.py $(ls /tmp/kt_$WL/*.db /tmp/kt_$WL/*/*.db 2>/dev/null | head -1) > $O/kt_$WL.txt 2>> $O/kt_$WL.log
head -45 $O/kt_$WL.txt
python $R/scripts/rocprof_timeline.py $(ls /tmp/kt_$WL/*.db /tmp/kt_$WL/*/*.db 2>/dev/null | head -1) k_chain_count ${TL_STEP:-8} 2 > $O/tl_$WL.txt 2>&1
