#!/bin/bash
# Round 4, session x: stress form of the fold race screens (concurrent HBM traffic on a second stream), five times.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04x
mkdir -p $OUT
for r in 1 2 3 4 5; do
  timeout 200 python3 -m pytest tests/test_kernels_gpu.py -x -q -s -m gpu -p no:cacheprovider -k "concurrent_hbm" 2>&1 | grep -E "fold stress|passed|failed|Error|assert" | tee -a $OUT/stress.log
done
