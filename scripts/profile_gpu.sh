#!/bin/bash
# Collect the rocprofv3 evidence for one bench workload on the GPU box (run through gpurun).
#   scripts/profile_gpu.sh cfg3 r01     -> gpurun_out/prof/<tag>_<workload>_{kt,fetch,write}/  (rocpd databases)
# Counters are collected in their own passes (--pmc with --kernel-trace only), as the pool requires.
set -u
WL=${1:-cfg3}; TAG=${2:-r01}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd $R
export CSV_BENCH_EXIT_ALARM=15
ARGS="bench.py --workload $WL --steps 50 --warmup 5 --no-cpu-baseline"
timeout -k 5 150 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_${WL}_kt -o kt -- python $ARGS > $OUT/${TAG}_${WL}_kt.log 2>&1
echo "kt rc=$?"
timeout -k 5 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/${TAG}_${WL}_fetch -o pmc -- python $ARGS > $OUT/${TAG}_${WL}_fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 5 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/${TAG}_${WL}_write -o pmc -- python $ARGS > $OUT/${TAG}_${WL}_write.log 2>&1
echo "write rc=$?"
ls -la $OUT/*/ | head -40
