#!/bin/bash
# Build ablation variants of libcutesv_hip.so (CSV_ABLATE bit mask, see kernels.hip.h) into build/:  scripts/ablate.sh 1 2 4 8 16 32 64
cd "$(dirname "$0")/../cutesv_amd/csrc"
mkdir -p ../../build
for m in "$@"; do
  make -s OUT=../../build/lib_abl$m.so EXTRA=-DCSV_ABLATE=$m ../../build/lib_abl$m.so 2>&1 | grep -E "error" &
done
wait
ls ../../build/
