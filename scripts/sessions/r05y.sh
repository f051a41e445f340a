#!/bin/bash
# Round 5, session y: LayerNorm backward (8 waves) with the next row's loads issued before this row's reductions (register prefetch)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r05y
export UNIVL_LIB=$PWD/univl_amd/lib/libunivl_hip_trace.so
for cfg in "8 2" "8 3" "8 4" "8 6"; do
  set -- $cfg
  echo "== waves per workgroup $1, rows per wave $2 (prefetch)"
  UNIVL_LN_NW=$1 UNIVL_LN_RPW=$2 timeout 200 python3 scripts/mb_ln_bwd_parts.py 2>&1 | grep "rows  6144\|rows  3072"
done | tee gpurun_out/r05y/mb_ln_prefetch.txt
unset UNIVL_LIB
timeout 300 python3 -m pytest tests/test_kernels_gpu.py -m gpu -q --no-header -p no:cacheprovider -x -k "layernorm" 2>&1 | tail -3 | tee gpurun_out/r05y/pytest_ln.txt
b() { local tag=$1; shift
  timeout 150 python3 bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-others --no-extras --no-preheat "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a gpurun_out/r05y/steps.txt; }
b b128_1 --batch 128; b b64_1 --batch 64; b b128_2 --batch 128; b b64_2 --batch 64
