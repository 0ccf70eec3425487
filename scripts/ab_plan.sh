#!/bin/bash
# where k_reads_plan spends its time: ablated libraries (scripts/ablate.sh 1048576 ... 16777216) on cfg4, reads_order slot
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for m in 1048576 2097152 4194304 8388608 16777216 ""; do
  if [ -n "$m" ]; then LABEL=abl$m CUTESV_AMD_LIB=$R/build/lib_abl$m.so REUSE_READS=0 timeout 200 python scripts/stage_times.py cfg4 50 2>&1 | tail -1 | grep -o "abl.*us/step\|reads_order=[0-9.]*" | tr '\n' ' '; echo
  else LABEL=full REUSE_READS=0 timeout 200 python scripts/stage_times.py cfg4 50 2>&1 | tail -1 | grep -o "full.*us/step\|reads_order=[0-9.]*" | tr '\n' ' '; echo; fi
done
