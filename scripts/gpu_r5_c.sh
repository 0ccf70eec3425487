#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5c; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 600 python scripts/oneshot_ab.py cfg3 cfg5 > $O/oneshot_ab.txt 2>&1; echo "ab rc=$?"; cat $O/oneshot_ab.txt | tail -12
CSV_DEBUG_TIMING=1 timeout 300 python scripts/oneshot_ab.py cfg3 2>&1 | grep "one shot" | tail -3 > $O/timing.txt; cat $O/timing.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/os_tl
( cd $R && timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/os_tl -o tl -- python scripts/oneshot_ab.py cfg3 > $O/tl.log 2>&1 )
DB=$(ls /tmp/os_tl/*.db /tmp/os_tl/*/*.db 2>/dev/null | head -1)
python $R/scripts/rocprof_oneshot_timeline.py $DB 1 > $O/oneshot_timeline.txt 2>&1
tail -24 $O/oneshot_timeline.txt
