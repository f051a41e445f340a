#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export UNIVL_LIB=$PWD/univl_amd/lib/libunivl_hip_trace.so
for cfg in "4 4" "8 2" "8 3" "8 4" "8 6"; do
  set -- $cfg
  echo "== waves per workgroup $1, rows per wave $2"
  UNIVL_LN_NW=$1 UNIVL_LN_RPW=$2 timeout 200 python scripts/mb_ln_bwd_parts.py 2>&1 | grep "rows  6144\|rows  3072"
done | tee gpurun_out/r05m_mb_ln_nw.txt
unset UNIVL_LIB
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "layernorm" 2>&1 | tail -3
