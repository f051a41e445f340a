#!/bin/bash
# Round 3, session r: position-table gradients by gather from 32 rows per position on -- kernel test, goldens at 128 rows, A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03r
mkdir -p $OUT
(timeout 150 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "golden and joint_b128" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout 150 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "rows_gather or embed or layernorm" > $OUT/pytest_k.log 2>&1; echo "rc=$?" >> $OUT/pytest_k.log) &
P2=$!
wait $P1 $P2
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_model.log | tail -5; grep -E "passed|failed|^FAILED|rc=" $OUT/pytest_k.log | tail -5
ab() { local name=$1; shift
  env "$@" timeout 60 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt; }
EXTRA="--batch 128" ab b128_gather UNIVL_X=0
EXTRA="--batch 128" ab b128_atomics UNIVL_DPOS_GATHER_MIN=0
EXTRA="--batch 128" ab b128_gather2 UNIVL_X=0
EXTRA="--batch 128" ab b128_atomics2 UNIVL_DPOS_GATHER_MIN=0
EXTRA="--batch 32" ab b32_gather UNIVL_X=0
EXTRA="--batch 32" ab b32_atomics UNIVL_DPOS_GATHER_MIN=0
