#!/bin/bash
# Round 4, session g: after the rect-form rule (shallow dgrad slices only), the LN rows-per-wave table and the epilogue prefetch:
# bench lines of every configuration, the phase trace at 192 / 768 rows, the GEMM kernel tests.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/r04g
mkdir -p $OUT
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $OUT/timeline.txt; }
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -p no:cacheprovider -k "gemm or layernorm" > $OUT/pytest_gemm.log 2>&1; tail -3 $OUT/pytest_gemm.log; stamp "kernel tests"
line() { local name=$1 envs=$2; shift 2
  env $envs timeout 120 python3 bench.py --no-cpu-baseline --no-others --no-extras "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "$name: $(grep -o '"ms_per_step": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"value": [0-9.]*' $OUT/bench_$name.json | head -1) $(grep -o '"last_loss": [0-9.]*' $OUT/bench_$name.json)" | tee -a $OUT/summary.txt; }
for r in 1 2; do
  line b4_$r "X=0" --steps 150 --warmup 10
  line b16_$r "X=0" --batch 16 --steps 100 --warmup 10
  for k in align caption; do line kind_${k}_$r "X=0" --kind $k --steps 60 --warmup 10; done
  line kind_pretrain6_$r "X=0" --kind pretrain --batch 6 --steps 60 --warmup 10
  for k in align caption; do line kind_${k}_square_$r "UNIVL_PAIR_FORM=square" --kind $k --steps 60 --warmup 10; done
  line kind_pretrain6_square_$r "UNIVL_PAIR_FORM=square" --kind pretrain --batch 6 --steps 60 --warmup 10
done
line b32 "X=0" --batch 32 --steps 60 --warmup 10
line b64 "X=0" --batch 64 --steps 40 --warmup 8
line b128 "X=0" --batch 128 --steps 30 --warmup 5
stamp "bench lines"
timeout 300 python3 scripts/mb_trace_gemm.py --rows 192,768 > $OUT/trace_gemm.txt 2>&1
grep -E "^fwd|^dgrad|^pair.*all|^----" $OUT/trace_gemm.txt | cut -c1-200
stamp "done"
