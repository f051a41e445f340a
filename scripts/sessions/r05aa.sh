#!/bin/bash
# Round 5, session aa: rider chunks with 16 loads per thread in flight (variant library -DUNIVL_RIDER_UNROLL4) against the product library, alternating
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r05aa
mkdir -p $OUT
b() { local tag=$1 lib=$2; shift 2
  UNIVL_LIB=$lib timeout 150 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-others --no-extras "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | sed "s/^/$tag: /" | tee -a $OUT/ab_rider_unroll4.txt; }
P=$PWD/univl_amd/lib/libunivl_hip.so; V=$PWD/univl_amd/lib/libunivl_hip_u4.so
for rep in 1 2 3; do b "u2_b4_$rep" $P; b "u4_b4_$rep" $V; done
for rep in 1 2; do b "u2_b16_$rep" $P --batch 16; b "u4_b16_$rep" $V --batch 16; done
b "u2_b128" $P --batch 128; b "u4_b128" $V --batch 128
UNIVL_LIB=$V timeout 300 python3 -m pytest tests/test_model_gpu.py -q -x -k "riding" -p no:cacheprovider 2>&1 | grep -v "Extension modules" | tail -3 | tee $OUT/pytest_riding_u4.txt
