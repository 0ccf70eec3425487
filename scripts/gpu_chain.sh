#!/bin/bash
# the device-side chains (extraction -> rebuild -> cluster): their tests, the pieces of the rebuild hand-off, both bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/chain; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "rebuild or unsorted or pool or extract or cigar or chain or tie" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout 300 python scripts/chain_prof.py 2>&1 | tail -6 | tee $O/chain_prof.txt
timeout 300 python bench.py --workload rebuild > $O/r05_bench_rebuild.json 2>/dev/null; python -c "import json;d=json.load(open('$O/r05_bench_rebuild.json'));print(d['ms_per_step'], d['chain'])"
timeout 300 python bench.py --workload extract > $O/r05_bench_extract.json 2>/dev/null; python -c "import json;d=json.load(open('$O/r05_bench_extract.json'));print(d['ms_per_step'], d['chain_extract_rebuild'])"
