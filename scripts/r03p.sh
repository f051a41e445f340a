#!/bin/bash
# Round 3, session p: fused similarity head (pool pair launches, sim backward folded in) -- kernel test, goldens, A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03p
mkdir -p $OUT
(timeout 280 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "(golden and joint) or deterministic or atomic or accumulation or riding or pretrain_small or graphed" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout 280 python -m pytest tests/test_kernels_gpu.py tests/test_eval_gpu.py -m gpu -q -p no:cacheprovider -k "pool or eval or retrieval or metrics" > $OUT/pytest_k.log 2>&1; echo "rc=$?" >> $OUT/pytest_k.log) &
P2=$!
wait $P1 $P2
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_model.log | tail -8; grep -E "passed|failed|^FAILED|rc=" $OUT/pytest_k.log | tail -5
ab() { local name=$1; shift
  env "$@" timeout 90 python bench.py --steps 200 --warmup 15 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt; }
EXTRA="" ab b4_fused UNIVL_X=0
EXTRA="" ab b4_separate UNIVL_FUSED_SIM=0
EXTRA="" ab b4_fused2 UNIVL_X=0
EXTRA="" ab b4_separate2 UNIVL_FUSED_SIM=0
EXTRA="--batch 16" ab b16_fused UNIVL_X=0
EXTRA="--batch 16" ab b16_separate UNIVL_FUSED_SIM=0
EXTRA="--batch 128 --steps 60" ab b128_fused UNIVL_X=0
EXTRA="--batch 128 --steps 60" ab b128_separate UNIVL_FUSED_SIM=0
