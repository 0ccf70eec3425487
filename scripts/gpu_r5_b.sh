#!/bin/bash
# round 5: timeline of the one-shot call (kernels + copies), no counters
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/os_tl
( cd $R && timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/os_tl -o tl -- python scripts/oneshot_ab.py cfg3 > $O/tl.log 2>&1 )
echo "rocprof rc=$?"; tail -3 $O/tl.log
DB=$(ls /tmp/os_tl/*.db /tmp/os_tl/*/*.db 2>/dev/null | head -1)
python $R/scripts/rocprof_oneshot_timeline.py $DB 2 > $O/oneshot_timeline.txt 2>&1
cat $O/oneshot_timeline.txt | tail -45
