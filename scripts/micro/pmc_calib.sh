#!/bin/bash
# PMC calibration on the GPU box (through gpurun): scripts/micro/pmc_calib.sh  ->  gpurun_out/calib/pmc_calibration.json (+ .txt)
# FETCH_SIZE and WRITE_SIZE in their own rocprofv3 passes (--pmc with --kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/calib; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cal_fetch /tmp/cal_write
$R/build/pmc_calib > $O/known.json || exit 1
timeout -k 5 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/cal_fetch -o pmc -- $R/build/pmc_calib > /dev/null 2> $O/fetch.log; echo "fetch rc=$?"
timeout -k 5 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/cal_write -o pmc -- $R/build/pmc_calib > /dev/null 2> $O/write.log; echo "write rc=$?"
python $R/scripts/micro/pmc_calib_report.py $O/known.json $(ls /tmp/cal_fetch/*.db /tmp/cal_fetch/*/*.db 2>/dev/null | head -1) $(ls /tmp/cal_write/*.db /tmp/cal_write/*/*.db 2>/dev/null | head -1) $O/pmc_calibration.json | tee $O/pmc_calibration.txt
