#!/bin/bash
# dynamic instruction counts per kernel (separate --pmc passes, kernel trace only):  scripts/pmc_insts.sh cfg3 r02
WL=${1:-cfg3}; TAG=${2:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export CSV_BENCH_EXIT_ALARM=15
ARGS="bench.py --workload $WL --steps 20 --warmup 3 --no-cpu-baseline"
DBS=""
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  ( cd $R && timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i -o pmc -- python $ARGS > $O/pmc_$i.log 2>&1 )
  echo "pass $i rc=$?"
  DBS="$DBS $(ls /tmp/pmc_$i/*.db /tmp/pmc_$i/*/*.db 2>/dev/null | head -1)"
done
python $R/scripts/rocprof_counters.py $DBS > $O/${TAG}_${WL}_insts.txt
cat $O/${TAG}_${WL}_insts.txt
