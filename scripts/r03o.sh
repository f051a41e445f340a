#!/bin/bash
# Round 3, session o: two-pass vectorised CE kernel -- kernel test, caption / pretrain goldens, A/B.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03o
mkdir -p $OUT
(timeout 250 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "(golden or deterministic or atomic) and (caption or pretrain)" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout 250 python -m pytest tests/test_kernels_gpu.py tests/test_decode_gpu.py -m gpu -q -p no:cacheprovider -k "ce_loss or mfm or decode or beam" > $OUT/pytest_k.log 2>&1; echo "rc=$?" >> $OUT/pytest_k.log) &
P2=$!
wait $P1 $P2
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_model.log | tail -8; grep -E "passed|failed|^FAILED|rc=" $OUT/pytest_k.log | tail -5
ab() { local name=$1; shift
  env "$@" timeout 90 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt; }
EXTRA="--kind caption" ab caption UNIVL_X=0
EXTRA="--kind pretrain --batch 6" ab pretrain UNIVL_X=0
EXTRA="--kind caption" ab caption2 UNIVL_X=0
