#!/bin/bash
# Round 4, session i: 128 x 64 weight-gradient body in the pair launch below 384 tokens: pair tests, A/B against round 2's square form at
# 4 / 8 pairs (UNIVL_PAIR_FORM=square), phase trace at 192 rows.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04i
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm" > $OUT/pytest_gemm.log 2>&1; tail -3 $OUT/pytest_gemm.log; stamp "kernel tests"
timeout 600 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "golden and (joint_full or caption_small or pretrain_small or joint_small)" > $OUT/pytest_golden.log 2>&1; tail -3 $OUT/pytest_golden.log; stamp "golden tests"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 120 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; }
for r in 1 2 3; do
  line b4_new_$r "X=0" --steps 150 --warmup 10
  line b4_square_$r "UNIVL_PAIR_FORM=square" --steps 150 --warmup 10
done
for r in 1 2; do
  line b6_new_$r "X=0" --batch 6 --steps 100 --warmup 10
  line b6_square_$r "UNIVL_PAIR_FORM=square" --batch 6 --steps 100 --warmup 10
done
stamp "ab"
timeout 200 python3 scripts/mb_trace_gemm.py --rows 192 > $OUT/trace_gemm.txt 2>&1
grep -E "^pair|^----" $OUT/trace_gemm.txt | cut -c1-200
stamp "done"
