#!/bin/bash
# Round 4, session d: colsum roles (pair + group launches), rectangular pair form from 384 tokens, mid-size split-K: kernel tests, golden
# tests at every plan size, bench lines, phase trace of the pair launches.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04d
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm" > $OUT/pytest_gemm.log 2>&1; tail -5 $OUT/pytest_gemm.log; stamp "gemm tests"
timeout 900 python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -p no:cacheprovider -k "golden and (joint_full or joint_b16 or joint_b128 or align_full or caption_full or pretrain_full)" > $OUT/pytest_golden.log 2>&1; tail -5 $OUT/pytest_golden.log; stamp "golden tests"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 120 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; }
line b4 "X=0" --steps 100 --warmup 10
line b8 "X=0" --batch 8 --steps 100 --warmup 10
line b8_nomid "UNIVL_SPLITK_MID=0" --batch 8 --steps 100 --warmup 10
line b16 "X=0" --batch 16 --steps 100 --warmup 10
line b16_nomid "UNIVL_SPLITK_MID=0" --batch 16 --steps 100 --warmup 10
line b32 "X=0" --batch 32 --steps 60 --warmup 10
line b32_nomid "UNIVL_SPLITK_MID=0" --batch 32 --steps 60 --warmup 10
line b64 "X=0" --batch 64 --steps 40 --warmup 8
line b128 "X=0" --batch 128 --steps 30 --warmup 5
for k in align caption pretrain; do line kind_$k "X=0" --kind $k --steps 60 --warmup 10; done
line kind_pretrain6 "X=0" --kind pretrain --batch 6 --steps 60 --warmup 10
stamp "bench lines"
timeout 300 python3 scripts/mb_trace_gemm.py --rows 192,768 > $OUT/trace_gemm.txt 2>&1
grep -E "^pair|^----" $OUT/trace_gemm.txt | grep -E "all|----" | cut -c1-200
stamp "done"
