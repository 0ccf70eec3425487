#!/bin/bash
# Round-end evidence on the GPU box (through gpurun): tests, profiles (kernel trace + calibrated PMC traffic), every bench line,
# the N-ranks-on-one-device rehearsal and a stress campaign.  scripts/round_end.sh r04   -> gpurun_out/final/
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
F=$R/gpurun_out/final; mkdir -p $F
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $F/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $F/${TAG}_pytest_gpu.log
for wl in cfg3 cfg5; do scripts/profile_insts.sh $wl $TAG > $F/${TAG}_${wl}_insts.log 2>&1; done
scripts/refresh_profiles.sh $TAG "cfg3 cfg4 cfg5" > $F/${TAG}_refresh.log 2>&1
# the drop-in under the reference's forked pool (bench_stage.py, a process of its own) and where its wall time goes
for wl in cfg3 cfg4; do timeout 600 python bench_stage.py --workload $wl > $F/${TAG}_stage_${wl}.json 2> $F/${TAG}_stage_${wl}.log; done
timeout 300 python scripts/stage_timeline.py --workload cfg4 --workers 1,8,32 2>/dev/null | grep -v "^\[" > $F/${TAG}_stage_timeline_cfg4_pickles.txt
timeout 300 python scripts/stage_timeline.py --workload cfg4 --workers 1,8,32 --cols 2>/dev/null | grep -v "^\[" > $F/${TAG}_stage_timeline_cfg4_cols.txt
for wl in cfg3 cfg4; do timeout 300 python scripts/stage_timeline.py --workload $wl --workers 8 --cols --all 2>/dev/null | grep -v "^\[" | cut -c1-1200 > $F/${TAG}_stage_timeline_${wl}_cols_all.txt; done
( echo "# the first 32-worker stage of a broker (after one 8-worker stage): descriptor table grown on demand, then grown once at start"; for v in 0 1; do echo "## CUTESV_AMD_BROKER_FD_TABLE=$v"; CUTESV_AMD_BROKER_FD_TABLE=$v timeout 300 python scripts/stage_timeline.py --workload cfg3 --workers 8,32 --all 2>/dev/null | grep -v "^\[" | awk '/== T=32/{f=1} f' | cut -c1-400 | head -42; done ) > $F/${TAG}_stage_timeline_cfg3_first_t32.txt
timeout 300 python bench.py --workload rebuild > $F/${TAG}_bench_rebuild.json 2>/dev/null
timeout 300 python bench.py --workload extract > $F/${TAG}_bench_extract.json 2>/dev/null
timeout 600 python bench.py --full --gpus 8 --steps 10 --warmup 2 > $F/${TAG}_bench_cfg3_replica_plus_cfg4_sharded_8ranks_one_device.json 2>/dev/null; echo "8 ranks rc=$?"
timeout 600 python bench.py --full --workload cfg5 --mode shard --gpus 8 --steps 5 --warmup 2 > $F/${TAG}_bench_cfg5_sharded_8ranks_one_device.json 2>/dev/null; echo "cfg5 8 ranks rc=$?"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --full --gpus 2 --steps 5 --warmup 2 2>/dev/null | tail -1 > $F/${TAG}_bench_torchrun_2ranks_one_device.json; echo "torchrun rc=$?"
timeout 700 python scripts/stress_gpu.py ${STRESS_N:-30000} 1200000 > $F/${TAG}_stress.log 2>&1; tail -2 $F/${TAG}_stress.log
timeout 300 python scripts/stress_gpu.py ${STRESS_BIG:-3000} 1300000 big > $F/${TAG}_stress_big.log 2>&1; tail -2 $F/${TAG}_stress_big.log
timeout 600 python scripts/stress_pool.py ${STRESS_POOL:-400} 70000 > $F/${TAG}_stress_pool.log 2>&1; tail -2 $F/${TAG}_stress_pool.log
ls $F | wc -l
# the host -> host call: bulk vs gate-first, full vs slim results; its kernels + copies on a timeline (no counters)
timeout 300 python scripts/oneshot_ab.py cfg3 cfg4 cfg5 cfg2 > $F/${TAG}_oneshot_ab.txt 2>&1; tail -4 $F/${TAG}_oneshot_ab.txt
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/os_tl && cd $R && timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/os_tl -o tl -- python scripts/oneshot_ab.py cfg3 > $F/${TAG}_oneshot_tl.log 2>&1 )
python scripts/rocprof_oneshot_timeline.py $(ls /tmp/os_tl/*.db /tmp/os_tl/*/*.db 2>/dev/null | head -1) 2 > $F/${TAG}_oneshot_timeline.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/os_tl4 && cd $R && timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/os_tl4 -o tl -- python scripts/oneshot_ab.py cfg4 only=gate-first/full > /dev/null 2>&1 )
python scripts/rocprof_oneshot_timeline.py $(ls /tmp/os_tl4/*.db /tmp/os_tl4/*/*.db 2>/dev/null | head -1) 1 > $F/${TAG}_oneshot_timeline_cfg4.txt 2>&1
python scripts/resource_usage.py > $F/${TAG}_resource_usage.txt 2>/dev/null
# where k_genotype's wavefronts spend their cycles (measurement build: make -C cutesv_amd/csrc gt-prof)
if [ -f $R/build/lib_prof.so ]; then for wl in cfg5 cfg4; do echo "# $wl"; CUTESV_AMD_LIB=$R/build/lib_prof.so timeout 300 python scripts/gt_prof.py $wl 2>&1 | grep gt_prof | tail -1; done > $F/${TAG}_gt_prof.txt; fi
# the pipelined resident loop: host time per call, and both queues + the copy engine on a timeline, block delivery vs k_publish in place
timeout 200 python scripts/pipe_ab.py cfg3 300 > $F/${TAG}_pipe_ab.txt 2>&1; CSV_PUB_INPLACE=1 timeout 200 python scripts/pipe_ab.py cfg3 300 2>&1 | grep pipelined | sed 's/^/in place: /' >> $F/${TAG}_pipe_ab.txt
for form in block inplace; do
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pp_$form && cd $R && if [ $form = inplace ]; then export CSV_PUB_INPLACE=1; fi; timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/pp_$form -o pp -- python scripts/pipe_ab.py cfg3 40 > /dev/null 2>&1 )
  python scripts/pipe_timeline.py $(ls /tmp/pp_$form/*.db /tmp/pp_$form/*/*.db 2>/dev/null | head -1) 3 > $F/${TAG}_pipe_timeline_$form.txt 2>&1
done
