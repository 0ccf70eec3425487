#!/bin/bash
# round 5, first GPU call: the new tests, then the one-shot A/B
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "gate_first or slim or download_into or pinned or published or narrow" > $O/pytest_new.log 2>&1
echo "pytest(new) rc=$?"; tail -15 $O/pytest_new.log
timeout -k 10 600 python scripts/oneshot_ab.py cfg3 cfg4 cfg5 > $O/oneshot_ab.txt 2>&1; echo "ab rc=$?"; cat $O/oneshot_ab.txt | tail -20
CSV_DEBUG_TIMING=1 timeout 300 python scripts/oneshot_ab.py cfg3 2>&1 | grep "one shot" | tail -8 > $O/timing.txt; cat $O/timing.txt
