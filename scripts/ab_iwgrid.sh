#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for g in 1536 2048 2560 3072 4096; do
  LABEL=g$g CSV_IW_GRID=$g timeout 300 python scripts/stage_times.py cfg3 2>&1 | tail -1 | cut -c1-130
done; done
