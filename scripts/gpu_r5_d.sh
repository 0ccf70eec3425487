#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/r5d; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log
for wl in cfg3 cfg5 cfg2; do scripts/kt.sh $wl 20 2>&1 | grep -E "rc=|k_chain_count<true>|k_refine_indel_wave<true>|k_refine<64, 256|k_emit|k_genotype<1024" | head -12; cp $R/gpurun_out/round/kt_$wl.txt $O/; done
scripts/profile_insts.sh cfg3 r05 2>&1 | tail -12
timeout 300 python scripts/stress_gpu.py 3000 2100000 > $O/stress.log 2>&1; tail -3 $O/stress.log
