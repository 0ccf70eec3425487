#!/bin/bash
# Instruction-level counters of one bench workload on the GPU box (through gpurun): two PMC passes of four SQ counters each.
#   scripts/profile_insts.sh cfg3 r04   ->  gpurun_out/final/<tag>_<workload>_insts.txt
set -u
WL=${1:-cfg3}; TAG=${2:-r04}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof; F=$R/gpurun_out/final
mkdir -p $OUT $F
cd $R
export CSV_BENCH_EXIT_ALARM=15
ARGS="bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline --no-others"
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/${TAG}_${WL}_i1 -o pmc -- python $ARGS > $OUT/${TAG}_${WL}_i1.log 2>&1
echo "pass 1 rc=$?"
timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/${TAG}_${WL}_i2 -o pmc -- python $ARGS > $OUT/${TAG}_${WL}_i2.log 2>&1
echo "pass 2 rc=$?"
cd $R/scripts && python rocprof_counters.py --json $F/insts_${WL}.json $(ls $OUT/${TAG}_${WL}_i1/*.db | head -1) $(ls $OUT/${TAG}_${WL}_i2/*.db | head -1) > $F/${TAG}_${WL}_insts.txt
rm -rf $OUT/${TAG}_${WL}_i1 $OUT/${TAG}_${WL}_i2
cp $F/insts_${WL}.json $R/profiles/insts_${WL}.json          # (the bench lines of this run pick it up)
head -16 $F/${TAG}_${WL}_insts.txt | cut -c1-250
