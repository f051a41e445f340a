#!/bin/bash
# Round 3, session i: 64-deep K steps (four workgroups per CU) for the deep-contraction 64 x 64 products at 128 pairs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03i
mkdir -p $OUT
(UNIVL_GEMM_NC64_MIN=1024 timeout 150 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "golden and joint_b128" > $OUT/pytest_nc64.log 2>&1; echo "rc=$?" >> $OUT/pytest_nc64.log) &
P1=$!
(UNIVL_GEMM_NC64_MIN=256 timeout 150 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "gemm" > $OUT/pytest_gemm_nc64.log 2>&1; echo "rc=$?" >> $OUT/pytest_gemm_nc64.log) &
P2=$!
wait $P1 $P2
tail -3 $OUT/pytest_nc64.log; tail -3 $OUT/pytest_gemm_nc64.log
ab() { local name=$1; shift
  env "$@" timeout 60 python bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt; }
STEPS=60
EXTRA="--batch 128" ab b128_base UNIVL_X=0
EXTRA="--batch 128" ab b128_nc64 UNIVL_GEMM_NC64_MIN=1024
EXTRA="--batch 128" ab b128_base2 UNIVL_X=0
EXTRA="--batch 128" ab b128_nc64_2 UNIVL_GEMM_NC64_MIN=1024
EXTRA="--batch 16" ab b16_base UNIVL_X=0
EXTRA="--batch 16" ab b16_nc64 UNIVL_GEMM_NC64_MIN=512
