#!/bin/bash
# Round 3, session m: vocabulary-head dgrad split over the vocabulary -- parity, A/B, and a kernel-stat profile of the caption step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r03m
mkdir -p $OUT
P=$PWD
(timeout 200 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "(golden or atomic or riding or deterministic) and (caption or pretrain)" > $OUT/pytest_model.log 2>&1; echo "rc=$?" >> $OUT/pytest_model.log) &
P1=$!
wait $P1
grep -E "passed|failed|^FAILED|^ERROR|rc=" $OUT/pytest_model.log | tail -8
ab() { local name=$1; shift
  env "$@" timeout 90 python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-extras $EXTRA > $OUT/ab_$name.json 2> $OUT/ab_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab_$name.json) $(grep -o '"last_loss": [0-9.]*' $OUT/ab_$name.json)" | tee -a $OUT/ab_summary.txt; }
EXTRA="--kind caption" ab caption_split UNIVL_X=0
EXTRA="--kind caption" ab caption_unsplit UNIVL_VOCAB_DGRAD_SPLIT=0
EXTRA="--kind pretrain --batch 6" ab pretrain_split UNIVL_X=0
EXTRA="--kind pretrain --batch 6" ab pretrain_unsplit UNIVL_VOCAB_DGRAD_SPLIT=0
EXTRA="--kind caption" ab caption_split2 UNIVL_X=0
EXTRA="--kind caption" ab caption_unsplit2 UNIVL_VOCAB_DGRAD_SPLIT=0
(cd /tmp && timeout 90 rocprofv3 --kernel-trace --stats -d $P/$OUT/prof -o cap --output-format csv -- python $P/bench.py --kind caption --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $P/$OUT/prof_bench.json 2> $P/$OUT/prof_bench.err)
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/caption_graph_kernel_stats.csv \; ; rm -rf $OUT/prof; head -22 $OUT/caption_graph_kernel_stats.csv | cut -c1-230
