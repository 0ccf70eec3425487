#!/bin/bash
# SQ_INSTS_VALU / SALU of one kernel for a list of library variants:  scripts/pmc_valu_ab.sh k_refine_indel_wave full= abl1=build/lib_abl1.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
KERN=$1; shift
cd /tmp && export TMPDIR=/tmp
export CSV_BENCH_EXIT_ALARM=15 CSV_BENCH_LENIENT=1
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  rm -rf /tmp/pv_$label
  ( cd $R && [ -n "$lib" ] && export CUTESV_AMD_LIB=$R/$lib; timeout -k 5 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/pv_$label -o pmc -- python scripts/stage_times.py ${WL:-cfg3} 10 > /tmp/pv_$label.log 2>&1 )
  python $R/scripts/rocprof_counters.py $(ls /tmp/pv_$label/*.db /tmp/pv_$label/*/*.db 2>/dev/null | head -1) | grep -E "^# kernel|$KERN" | sed "s/^/$label  /"
done
