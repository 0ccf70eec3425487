#!/bin/bash
# Round-end evidence refresh on the GPU box (run through gpurun):  scripts/refresh_profiles.sh r01 "cfg3 cfg4 cfg5"
# rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE passes per workload -> text summaries and traffic JSON,
# then the bench lines (which pick the fresh traffic up).  Everything lands in gpurun_out/final/ (copy to profiles/).
TAG=${1:-r03}; WLS=${2:-cfg3 cfg4 cfg5}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
F=$R/gpurun_out/final; mkdir -p $F
for wl in $WLS; do
  scripts/profile_gpu.sh $wl $TAG > $F/${TAG}_${wl}_profile.log 2>&1
  P=$R/gpurun_out/prof
  python scripts/rocprof_summary.py $(ls $P/${TAG}_${wl}_kt/*.db | head -1) > $F/${TAG}_${wl}_kernel_trace.txt 2>> $F/${TAG}_${wl}_profile.log
  python scripts/rocprof_traffic.py $(ls $P/${TAG}_${wl}_fetch/*.db | head -1) $(ls $P/${TAG}_${wl}_write/*.db | head -1) $F/traffic_${wl}.json > $F/${TAG}_${wl}_pmc.txt 2>> $F/${TAG}_${wl}_profile.log
  cp $F/traffic_${wl}.json profiles/traffic_${wl}.json
  rm -rf $P/${TAG}_${wl}_kt $P/${TAG}_${wl}_fetch $P/${TAG}_${wl}_write
done
for wl in $WLS cfg2; do
  if [ $wl = cfg3 ]; then timeout 900 python bench.py --full --workload $wl > $F/${TAG}_bench_${wl}.json 2> $F/${TAG}_bench_${wl}.err
  else timeout 900 python bench.py --full --workload $wl --steps 50 --warmup 5 > $F/${TAG}_bench_${wl}.json 2> $F/${TAG}_bench_${wl}.err; fi
  tail -c 300 $F/${TAG}_bench_${wl}.json
done
ls -la $F
