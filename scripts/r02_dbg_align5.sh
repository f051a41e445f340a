#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r02l
mkdir -p $OUT
for i in 1 2 3 4 5 6 7 8; do
  UNIVL_GUARD=0 timeout 100 python scripts/dbg_guard.py align_full bf16 > $OUT/run$i.log 2>&1
  echo "run $i: $(grep -h 'significant' $OUT/run$i.log | sed 's/.*median//' | cut -c1-90)"
  grep -h "per-group" $OUT/run$i.log | cut -c1-1200
done
