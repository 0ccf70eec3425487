#!/bin/bash
# quick A/B of libraries on the resident step: scripts/ab_stage.sh "<label>=<lib path or empty>" ...   (WL=cfg3)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
WL=${WL:-cfg3}
for rep in 1 2; do
for spec in "$@"; do
  label=${spec%%=*}; lib=${spec#*=}
  if [ -n "$lib" ]; then LABEL=$label CUTESV_AMD_LIB=$lib timeout 300 python scripts/stage_times.py $WL 2>&1 | tail -1
  else LABEL=$label timeout 300 python scripts/stage_times.py $WL 2>&1 | tail -1; fi
done
done
