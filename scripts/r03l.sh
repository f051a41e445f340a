#!/bin/bash
# Round 3, session l: decoder weight gradients riding with their dgrad products -- parity, then A/B at cfg4 / cfg5.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03l
mkdir -p $OUT
(timeout 280 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "caption or pretrain" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log) &
P1=$!
(timeout 280 python -m pytest tests/test_decode_gpu.py tests/test_eval_gpu.py -m gpu -q -p no:cacheprovider > $OUT/pytest_rest.log 2>&1; echo "rc=$?" >> $OUT/pytest_rest.log) &
P2=$!
wait $P1 $P2
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_model.log | tail -8; grep -E "passed|failed|rc=" $OUT/pytest_rest.log | tail -3
ab() { local name=$1; shift
  env "$@" timeout 90 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt; }
EXTRA="--kind caption" ab caption_pair UNIVL_X=0
EXTRA="--kind caption" ab caption_seq UNIVL_DECODER_PAIR=0
EXTRA="--kind pretrain --batch 6" ab pretrain_pair UNIVL_X=0
EXTRA="--kind pretrain --batch 6" ab pretrain_seq UNIVL_DECODER_PAIR=0
EXTRA="--kind caption" ab caption_pair2 UNIVL_X=0
EXTRA="--kind caption" ab caption_seq2 UNIVL_DECODER_PAIR=0
