#!/bin/bash
# Round 3, session h: how much of a GEMM is its epilogue's stores?  (UNIVL_GEMM_PROBE=1: computed, not stored -- garbage results, timing only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03h
mkdir -p $OUT
ab() { local name=$1; shift
  env "$@" timeout 60 python bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt; }
STEPS=40
EXTRA="--batch 128 --no-pipeline" ab b128_base UNIVL_X=0
EXTRA="--batch 128 --no-pipeline" ab b128_nostore UNIVL_GEMM_PROBE=1
EXTRA="--batch 16 --no-pipeline" ab b16_base UNIVL_X=0
EXTRA="--batch 16 --no-pipeline" ab b16_nostore UNIVL_GEMM_PROBE=1
STEPS=100
EXTRA="--no-pipeline" ab b4_base UNIVL_X=0
EXTRA="--no-pipeline" ab b4_nostore UNIVL_GEMM_PROBE=1
